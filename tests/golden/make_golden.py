#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by importing the reference.

Runs ONLY in the build container (needs /root/reference). Nothing under tests/,
bench.py or smoke() reads /root/reference at run time: they read the .npz/.json
files this script writes. Fixtures are data (inputs + the reference's outputs);
no reference source text is stored.

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden.py

What is captured:
  fk.npz               q -> pose6 for 4096 joint vectors (stage-11 box + full limits)
  pose_error.npz       pose_error_components incl. wrap edge cases
  joint_utils.npz      margin / q_norm / dq_norm / clip
  configs/*.json       resolved (deep-merged) config dicts the reference trainers see
  trace_*.npz          open-loop step traces with scripted actions + auto-reset
  resets_*.npz         first 64 resets (seeded stream) + PCG64 state before/after
  eval_suites.npz      build_curriculum_local_eval_suite(seed+1009k, k, 8), k=0..11
  curriculum_tracker.json   PointCurriculumTracker promotion traces
  gated_score.json     gated_score on synthetic stage tables
  handoff_state_buffer.json  SYNTHETIC finisher handoff buffer (the reference's is git-ignored)
"""
from __future__ import annotations

import json
import os
import sys
from pathlib import Path

import numpy as np

REF = Path("/root/reference/hrl_ws/src/hrl_trainer")
sys.dont_write_bytecode = True
sys.path.insert(0, str(REF))

from hrl_trainer.kinematic_phase1.envs.arm_kinematic_env import ArmKinematicEnv  # noqa: E402
from hrl_trainer.kinematic_phase1.envs.curriculum import (  # noqa: E402
    PointCurriculumConfig,
    PointCurriculumTracker,
)
from hrl_trainer.kinematic_phase1.eval.fixed_eval_suite import build_curriculum_local_eval_suite  # noqa: E402
from hrl_trainer.kinematic_phase1.kinematics import joint_limits as jl  # noqa: E402
from hrl_trainer.kinematic_phase1.kinematics.fk_interface import compute_ee_pose6  # noqa: E402
from hrl_trainer.kinematic_phase1.kinematics.pose_utils import pose_error_components  # noqa: E402
from hrl_trainer.kinematic_phase1.training.policy_config import (  # noqa: E402
    approach_default_config_path,
    config_dir,
    deep_merge,
    dock_default_config_path,
    load_yaml_file,
    ppo_default_config_path,
    to_env_config,
)
from hrl_trainer.kinematic_phase1.train_workspace_expansion import _load_config as _load_ws_config  # noqa: E402
from hrl_trainer.kinematic_phase1.workspace.workspace_curriculum import gate_config_from_dict, gated_score  # noqa: E402

OUT = Path(__file__).resolve().parent
OBS_KEYS = [
    "q", "dq", "prev_action", "goal_pos_err", "goal_ori_err", "wp_pos_err", "wp_ori_err",
    "next_wp_pos_err", "next_wp_ori_err", "task_type", "mode_flag", "progress", "joint_limit_margin",
]


def flat_obs(obs: dict) -> np.ndarray:
    return np.concatenate([np.asarray(obs[k], dtype=np.float32).ravel() for k in OBS_KEYS])


def rng_state_words(rng: np.random.Generator) -> np.ndarray:
    st = rng.bit_generator.state
    s, inc = int(st["state"]["state"]), int(st["state"]["inc"])
    m = (1 << 64) - 1
    return np.array([s >> 64, s & m, inc >> 64, inc & m, int(st["has_uint32"]), int(st["uinteger"])], dtype=np.uint64)


# --------------------------------------------------------------------------- configs
def resolved_configs() -> dict[str, dict]:
    cdir = config_dir()
    cfgs: dict[str, dict] = {}
    # config 1: approach_default + ppo_default (what train_approach_policy sees with no overlay)
    cfgs["approach_default"] = deep_merge(load_yaml_file(approach_default_config_path()), load_yaml_file(ppo_default_config_path()))
    for name in (
        "workspace_expansion_bigtrain",
        "workspace_expansion_1h_extend",
        "workspace_expansion_dynamic_scale_big",
        "workspace_expansion_late_stage_ft",
        "workspace_full_coverage_randomstart_overnight",
    ):
        cfgs[name] = _load_ws_config(str(cdir / f"{name}.yaml"))
    # finisher: the evaluator loads it raw (eval_workspace_expansion.py:103); the trainer merges defaults.
    raw = load_yaml_file(cdir / "dock_workspace_handoff_noop_ft_12env.yaml")
    cfgs["dock_workspace_handoff_noop_ft_12env_raw"] = raw
    cfgs["dock_workspace_handoff_noop_ft_12env"] = deep_merge(
        deep_merge(load_yaml_file(dock_default_config_path()), load_yaml_file(ppo_default_config_path())), raw
    )
    cfgs["approach_finisher_ready_v2_settle"] = deep_merge(
        deep_merge(load_yaml_file(approach_default_config_path()), load_yaml_file(ppo_default_config_path())),
        load_yaml_file(cdir / "approach_finisher_ready_v2_settle.yaml"),
    )
    return cfgs


# --------------------------------------------------------------------------- basic kinematics
def gen_fk(rng: np.random.Generator) -> None:
    specs = jl.default_joint_specs()
    lo, hi = jl.lower_bounds(specs), jl.upper_bounds(specs)
    n = 4096
    box11 = np.array([0.132, 0.396, 0.484, 0.352, 0.22, 0.176, 0.154])
    q = np.empty((n, 7))
    q[: n // 2] = rng.uniform(-box11, box11, size=(n // 2, 7))
    q[n // 2 :] = rng.uniform(lo, hi, size=(n - n // 2, 7))
    q[0] = 0.0
    q[1] = [0.12, -0.35, 0.48, -0.62, 0.27, -0.14, 0.51]
    q[2] = lo
    q[3] = hi
    pose = np.stack([compute_ee_pose6(row) for row in q])
    np.savez_compressed(OUT / "fk.npz", q=q, pose6=pose)


def gen_pose_error(rng: np.random.Generator) -> None:
    n = 512
    curr = rng.uniform(-1.0, 1.0, size=(n, 6))
    goal = rng.uniform(-1.0, 1.0, size=(n, 6))
    curr[:, 3:] = rng.uniform(-np.pi, np.pi, size=(n, 3))
    goal[:, 3:] = rng.uniform(-np.pi, np.pi, size=(n, 3))
    edge = [np.pi, -np.pi, 3 * np.pi, -3 * np.pi, 0.0, 2 * np.pi, np.pi - 1e-12, -np.pi + 1e-12]
    for i, e in enumerate(edge):
        curr[i, 3:] = 0.0
        goal[i, 3:] = e
    pe = np.empty((n, 3))
    oe = np.empty((n, 3))
    for i in range(n):
        p, o = pose_error_components(curr[i], goal[i])
        pe[i], oe[i] = p, o
    np.savez_compressed(OUT / "pose_error.npz", curr=curr, goal=goal, pos_err=pe, ori_err=oe)


def gen_joint_utils(rng: np.random.Generator) -> None:
    specs = jl.default_joint_specs()
    lo, hi = jl.lower_bounds(specs), jl.upper_bounds(specs)
    n = 256
    q = rng.uniform(lo * 1.2, hi * 1.2, size=(n, 7))
    dq = rng.uniform(-0.6, 0.6, size=(n, 7))
    np.savez_compressed(
        OUT / "joint_utils.npz",
        q=q,
        dq=dq,
        lower=lo,
        upper=hi,
        delta_limits=jl.delta_limits(specs),
        clipped=np.stack([jl.clip_joint_configuration(r, specs) for r in q]),
        margin=np.stack([jl.joint_limit_margin(jl.clip_joint_configuration(r, specs), specs) for r in q]),
        q_norm=np.stack([jl.normalize_joint_positions(r, specs) for r in q]),
        dq_norm=np.stack([jl.normalize_joint_deltas(r, specs) for r in dq]),
    )


# --------------------------------------------------------------------------- traces
def scripted_action(kind: str, t: int, env: ArmKinematicEnv, arng: np.random.Generator) -> np.ndarray:
    """Action scripts that reach the zones random actions never reach."""
    cfg = env.config
    dl = jl.delta_limits(cfg.joint_specs)
    scale = cfg.action_delta_scale
    if env._policy_mode_name == "dock" and cfg.dock_action_delta_scale > 0.0:
        scale = cfg.dock_action_delta_scale
    servo = (env._goal_q - env._q) / (dl * max(scale, 1e-9))
    if kind == "servo":
        return 0.7 * servo + arng.uniform(-0.004, 0.004, size=7)
    if kind == "servo_exact":
        return servo
    if kind == "random":
        return arng.uniform(-1.5, 1.5, size=7)
    if kind == "kick":
        if 30 <= t % 48 < 33:
            return arng.uniform(-1.0, 1.0, size=7)
        return 0.8 * servo + arng.uniform(-0.02, 0.02, size=7)
    if kind == "zero":
        return np.zeros(7)
    if kind == "saturate":
        sign = 1.0 if (t // 70) % 2 == 0 else -1.0
        return np.full(7, 1.3 * sign)
    if kind == "settle":  # reach the goal then issue tiny corrections (finisher regime)
        return np.clip(servo, -1.0, 1.0) * 0.9 + arng.uniform(-0.01, 0.01, size=7)
    raise ValueError(kind)


def run_trace(name: str, cfg_dict: dict, *, seed: int, stage: int, steps: int, kinds: list[str], mode: str | None = None) -> None:
    env_cfg = to_env_config(cfg_dict)
    env = ArmKinematicEnv(config=env_cfg)
    env.set_curriculum_stage(stage)
    if mode is not None:
        env.set_policy_mode(mode)
    arng = np.random.default_rng(10_000 + seed)

    rec: dict[str, list] = {k: [] for k in (
        "action", "obs", "reward", "terminated", "truncated", "success", "pos_err", "ori_err", "dwell", "entry",
        "drift", "pre_near_hit", "near_hit", "min_pos_error", "q", "dq", "ee_pose6", "components", "exec_dq_l2",
        "action_l2", "dq_change_l2", "dock_action_limit", "dock_dq_change_limit_scale", "margin_min", "episode_step",
    )}
    rst: dict[str, list] = {k: [] for k in (
        "at_step", "initial_q", "initial_dq", "initial_prev_action", "goal_q", "goal_pose6", "ee_pose6", "obs",
        "entry_metrics", "rng_before", "rng_after",
    )}
    comp_keys: list[str] | None = None

    def do_reset(at: int, first: bool) -> None:
        before = None
        if first:
            # default_rng(seed) happens inside reset(seed=...); record the fresh-stream state
            before = rng_state_words(np.random.default_rng(seed))
            obs, info = env.reset(seed=seed)
        else:
            before = rng_state_words(env._rng)
            obs, info = env.reset()
        rst["at_step"].append(at)
        rst["initial_q"].append(env._q.copy())
        rst["initial_dq"].append(env._dq.copy())
        rst["initial_prev_action"].append(env._prev_action.copy())
        rst["goal_q"].append(env._goal_q.copy())
        rst["goal_pose6"].append(env._goal_pose6.copy())
        rst["ee_pose6"].append(env._ee_pose6.copy())
        rst["obs"].append(flat_obs(obs))
        rst["entry_metrics"].append([
            info["entry_position_error_norm"], info["entry_orientation_error_norm"], info["entry_action_l2"], info["entry_dq_norm"],
        ])
        rst["rng_before"].append(before)
        rst["rng_after"].append(rng_state_words(env._rng))

    do_reset(0, True)
    episode = 0
    t_in_ep = 0
    for t in range(steps):
        kind = kinds[episode % len(kinds)]
        a = scripted_action(kind, t_in_ep, env, arng)
        obs, reward, terminated, truncated, info = env.step(a)
        t_in_ep += 1
        comps = info["reward_components"]
        if comp_keys is None:
            comp_keys = list(comps.keys())
        assert list(comps.keys()) == comp_keys
        rec["action"].append(a)
        rec["obs"].append(flat_obs(obs))
        rec["reward"].append(reward)
        rec["terminated"].append(terminated)
        rec["truncated"].append(truncated)
        rec["success"].append(info["success"])
        rec["pos_err"].append(info["position_error_norm"])
        rec["ori_err"].append(info["orientation_error_norm"])
        rec["dwell"].append(info["dwell_count"])
        rec["entry"].append(info["near_goal_entry_count"])
        rec["drift"].append(info["near_goal_drift_count"])
        rec["pre_near_hit"].append(info["pre_near_goal_hit"])
        rec["near_hit"].append(info["near_goal_hit"])
        rec["min_pos_error"].append(info["min_position_error"])
        rec["q"].append(info["q"])
        rec["dq"].append(info["dq"])
        rec["ee_pose6"].append(info["ee_pose6"])
        rec["components"].append([comps[k] for k in comp_keys])
        rec["exec_dq_l2"].append(info["executed_delta_q_l2"])
        rec["action_l2"].append(info["action_l2"])
        rec["dq_change_l2"].append(info["delta_q_change_l2"])
        rec["dock_action_limit"].append(info["dock_action_limit"])
        rec["dock_dq_change_limit_scale"].append(info["dock_delta_q_change_limit_scale"])
        rec["margin_min"].append(info["joint_limit_margin_min"])
        rec["episode_step"].append(info["step_count"])
        if terminated or truncated:
            episode += 1
            t_in_ep = 0
            do_reset(t + 1, False)

    arrays = {
        "action": np.asarray(rec["action"], dtype=np.float64),
        "obs": np.asarray(rec["obs"], dtype=np.float32),
        "reward": np.asarray(rec["reward"], dtype=np.float64),
        "terminated": np.asarray(rec["terminated"], dtype=np.uint8),
        "truncated": np.asarray(rec["truncated"], dtype=np.uint8),
        "success": np.asarray(rec["success"], dtype=np.uint8),
        "pos_err": np.asarray(rec["pos_err"]),
        "ori_err": np.asarray(rec["ori_err"]),
        "dwell": np.asarray(rec["dwell"], dtype=np.int32),
        "entry": np.asarray(rec["entry"], dtype=np.int32),
        "drift": np.asarray(rec["drift"], dtype=np.int32),
        "pre_near_hit": np.asarray(rec["pre_near_hit"], dtype=np.uint8),
        "near_hit": np.asarray(rec["near_hit"], dtype=np.uint8),
        "min_pos_error": np.asarray(rec["min_pos_error"]),
        "q": np.asarray(rec["q"]),
        "dq": np.asarray(rec["dq"]),
        "ee_pose6": np.asarray(rec["ee_pose6"]),
        "components": np.asarray(rec["components"], dtype=np.float64),
        "component_keys": np.asarray(comp_keys),
        "exec_dq_l2": np.asarray(rec["exec_dq_l2"]),
        "action_l2": np.asarray(rec["action_l2"]),
        "dq_change_l2": np.asarray(rec["dq_change_l2"]),
        "dock_action_limit": np.asarray(rec["dock_action_limit"]),
        "dock_dq_change_limit_scale": np.asarray(rec["dock_dq_change_limit_scale"]),
        "margin_min": np.asarray(rec["margin_min"]),
        "episode_step": np.asarray(rec["episode_step"], dtype=np.int32),
        "obs_keys": np.asarray(OBS_KEYS),
        "meta": np.asarray(json.dumps({"seed": seed, "stage": stage, "steps": steps, "kinds": kinds, "mode": mode or env_cfg.mode_name})),
    }
    for k, v in rst.items():
        dt = np.uint64 if k.startswith("rng") else (np.int64 if k == "at_step" else (np.float32 if k == "obs" else np.float64))
        arrays["reset_" + k] = np.asarray(v, dtype=dt)
    np.savez_compressed(OUT / f"trace_{name}.npz", **arrays)
    print(f"trace_{name}: steps={steps} resets={len(rst['at_step'])} success_steps={int(arrays['success'].sum())} "
          f"max_dwell={int(arrays['dwell'].max())} max_entry={int(arrays['entry'].max())} max_drift={int(arrays['drift'].max())}")


def gen_resets(name: str, cfg_dict: dict, *, seed: int, stage: int, n: int = 64, mode: str | None = None) -> None:
    env = ArmKinematicEnv(config=to_env_config(cfg_dict))
    env.set_curriculum_stage(stage)
    if mode is not None:
        env.set_policy_mode(mode)
    out: dict[str, list] = {k: [] for k in ("initial_q", "initial_dq", "initial_prev_action", "goal_q", "goal_pose6", "obs", "rng_before", "rng_after")}
    for i in range(n):
        if i == 0:
            before = rng_state_words(np.random.default_rng(seed))
            obs, _ = env.reset(seed=seed)
        else:
            before = rng_state_words(env._rng)
            obs, _ = env.reset()
        out["initial_q"].append(env._q.copy())
        out["initial_dq"].append(env._dq.copy())
        out["initial_prev_action"].append(env._prev_action.copy())
        out["goal_q"].append(env._goal_q.copy())
        out["goal_pose6"].append(env._goal_pose6.copy())
        out["obs"].append(flat_obs(obs))
        out["rng_before"].append(before)
        out["rng_after"].append(rng_state_words(env._rng))
    arrays = {k: np.asarray(v, dtype=(np.uint64 if k.startswith("rng") else (np.float32 if k == "obs" else np.float64))) for k, v in out.items()}
    arrays["meta"] = np.asarray(json.dumps({"seed": seed, "stage": stage, "mode": mode}))
    np.savez_compressed(OUT / f"resets_{name}.npz", **arrays)


def make_handoff_buffer(cfg_dict: dict) -> Path:
    """Synthetic finisher handoff buffer: near-goal states in the stage 0-5 shells."""
    env_cfg = to_env_config(cfg_dict)
    rng = np.random.default_rng(4242)
    states = []
    for i in range(96):
        stage = env_cfg.curriculum_config.stages[i % 6]
        gq = np.clip(np.asarray(stage.goal_q) + rng.uniform(-np.asarray(stage.goal_noise), np.asarray(stage.goal_noise)),
                     jl.lower_bounds(env_cfg.joint_specs), jl.upper_bounds(env_cfg.joint_specs))
        iq = gq + rng.uniform(-1.0, 1.0, size=7) * np.array([0.0006, 0.0012, 0.0018, 0.0012, 0.0009, 0.0009, 0.0006]) * (1 + 3 * (i % 4 == 0))
        gp = compute_ee_pose6(gq)
        p, o = pose_error_components(compute_ee_pose6(iq), gp)
        pa = rng.uniform(-0.04, 0.04, size=7)
        states.append({
            "initial_q": iq.tolist(),
            "goal_q": gq.tolist(),
            "goal_pose6": gp.tolist(),
            "initial_dq": rng.uniform(-4e-4, 4e-4, size=7).tolist(),
            "initial_prev_action": pa.tolist(),
            "position_error_norm": float(np.linalg.norm(p)),
            "orientation_error_norm": float(np.linalg.norm(o)),
            "action_l2": float(np.linalg.norm(pa)),
        })
    path = OUT / "handoff_state_buffer.json"
    path.write_text(json.dumps({"states": states}, indent=0))
    return path


def gen_eval_suites(cfg_dict: dict) -> None:
    env_cfg = to_env_config(cfg_dict)
    out = {}
    for k in range(len(env_cfg.curriculum_config.stages)):
        suite = build_curriculum_local_eval_suite(env_cfg, seed=700001 + 1009 * k, stage_index=k, n_episodes=8)
        out[f"s{k}_initial_q"] = np.asarray([e.initial_q for e in suite])
        out[f"s{k}_goal_q"] = np.asarray([e.goal_q for e in suite])
        out[f"s{k}_goal_pose6"] = np.asarray([e.goal_pose6 for e in suite])
    np.savez_compressed(OUT / "eval_suites.npz", **out)


def gen_curriculum_tracker() -> None:
    rng = np.random.default_rng(99)
    cases = []
    for (thr, window, min_eps, p) in [(0.8, 20, 30, 0.9), (0.72, 48, 96, 0.8), (0.5, 4, 2, 0.6), (0.72, 48, 96, 0.7)]:
        cfg = PointCurriculumConfig(success_rate_threshold=thr, window_episodes=window, min_episodes_per_stage=min_eps)
        tr = PointCurriculumTracker(cfg)
        seq = (rng.random(600) < p).astype(int).tolist()
        promoted_at, stages = [], []
        for i, s in enumerate(seq):
            if tr.record_episode(success=bool(s)):
                promoted_at.append(i)
            stages.append(tr.stage_index)
        cases.append({"threshold": thr, "window": window, "min_episodes": min_eps, "n_stages": len(cfg.stages),
                      "successes": seq, "promoted_at": promoted_at, "stage_after": stages,
                      "trigger_rates": [h["trigger_success_rate"] for h in tr.history]})
    (OUT / "curriculum_tracker.json").write_text(json.dumps({"cases": cases}))


def gen_gated_score(cfgs: dict[str, dict]) -> None:
    rng = np.random.default_rng(5)
    cases = []
    for name in ("workspace_expansion_bigtrain", "workspace_expansion_1h_extend", "workspace_full_coverage_randomstart_overnight"):
        gate = cfgs[name]["workspace_expansion"]["gate"]
        for trial in range(4):
            table = {}
            for k in range(12):
                base = max(0.2, 1.0 - 0.06 * k - 0.05 * trial)
                table[k] = {
                    "success_rate": float(np.clip(base + rng.uniform(-0.05, 0.05), 0, 1)),
                    "finisher_ready_hit_rate": float(np.clip(base + rng.uniform(-0.05, 0.1), 0, 1)),
                    "dwell_success_rate": float(np.clip(base + rng.uniform(-0.1, 0.05), 0, 1)),
                    "mean_final_position_error": float(0.002 + 0.002 * k * rng.uniform(0.5, 1.5)),
                    "mean_final_orientation_error": float(0.02 + 0.012 * k * rng.uniform(0.5, 1.5)),
                    "mean_final_action_magnitude": float(rng.uniform(0.01, 0.2)),
                    "mean_final_dq_norm": float(rng.uniform(0.0, 0.004)),
                    "regression_rate": float(rng.uniform(0, 0.3)),
                }
            score_idx = int(gate.get("score_stage_index", 11))
            sel = gated_score(table, score_idx, gate_config_from_dict(gate))
            cases.append({"gate": gate, "score_stage_index": score_idx, "stage_metrics": {str(k): v for k, v in table.items()}, "selection": sel})
    (OUT / "gated_score.json").write_text(json.dumps({"cases": cases}))


def main() -> None:
    os.chdir(OUT)
    rng = np.random.default_rng(20260422)
    gen_fk(rng)
    gen_pose_error(rng)
    gen_joint_utils(rng)

    cfgs = resolved_configs()
    (OUT / "configs").mkdir(exist_ok=True)

    hb = make_handoff_buffer(cfgs["workspace_expansion_bigtrain"])
    for key in ("dock_workspace_handoff_noop_ft_12env_raw", "dock_workspace_handoff_noop_ft_12env"):
        # the reference's buffer is git-ignored; point the finisher config at the synthetic one (relative name,
        # resolved against tests/golden by the test suite)
        cfgs[key]["env"]["dock_reset"]["handoff_state_buffer_path"] = hb.name
    for name, cfg in cfgs.items():
        (OUT / "configs" / f"{name}.json").write_text(json.dumps(cfg, indent=1, sort_keys=True))

    approach_kinds = ["servo", "random", "kick", "saturate", "settle", "zero"]
    run_trace("approach_default_s0_seed7", cfgs["approach_default"], seed=7, stage=0, steps=160, kinds=approach_kinds)
    run_trace("approach_default_s5_seed0", cfgs["approach_default"], seed=0, stage=5, steps=120, kinds=["servo", "kick", "random"])
    run_trace("bigtrain_s5_seed806", cfgs["workspace_expansion_bigtrain"], seed=806, stage=5, steps=96 * 6, kinds=approach_kinds)
    run_trace("bigtrain_s0_seed123", cfgs["workspace_expansion_bigtrain"], seed=123, stage=0, steps=96 * 2, kinds=["settle", "kick"])
    run_trace("extend_s11_seed0", cfgs["workspace_expansion_1h_extend"], seed=0, stage=11, steps=96 * 5, kinds=["kick", "servo", "random", "settle", "saturate"])
    run_trace("extend_s8_seed7", cfgs["workspace_expansion_1h_extend"], seed=7, stage=8, steps=96 * 3, kinds=["servo", "settle", "kick"])
    run_trace("dynscale_s9_seed123", cfgs["workspace_expansion_dynamic_scale_big"], seed=123, stage=9, steps=128 * 3, kinds=["servo", "kick", "random"])
    run_trace("randomstart_s10_seed931", cfgs["workspace_full_coverage_randomstart_overnight"], seed=931, stage=10, steps=160 * 5, kinds=["servo", "random", "settle", "kick", "zero"])
    run_trace("settle_v2_s5_seed7", cfgs["approach_finisher_ready_v2_settle"], seed=7, stage=5, steps=300, kinds=["settle", "kick", "servo"])
    dock_kinds = ["settle", "zero", "kick", "random", "servo_exact"]
    run_trace("dock_noop_seed7", cfgs["dock_workspace_handoff_noop_ft_12env_raw"], seed=7, stage=0, steps=36 * 12, kinds=dock_kinds)
    run_trace("dock_noop_merged_seed0", cfgs["dock_workspace_handoff_noop_ft_12env"], seed=0, stage=0, steps=36 * 8, kinds=dock_kinds)
    dock_default = deep_merge(load_yaml_file(dock_default_config_path()), load_yaml_file(ppo_default_config_path()))
    cfgs["dock_default"] = dock_default
    (OUT / "configs" / "dock_default.json").write_text(json.dumps(dock_default, indent=1, sort_keys=True))
    run_trace("dock_default_seed123", dock_default, seed=123, stage=0, steps=20 * 10, kinds=dock_kinds)

    for nm, key, seed, stage in [
        ("approach_default_s0", "approach_default", 7, 0),
        ("approach_default_s3", "approach_default", 123, 3),
        ("bigtrain_s5", "workspace_expansion_bigtrain", 806, 5),
        ("bigtrain_s9", "workspace_expansion_bigtrain", 0, 9),
        ("extend_s11", "workspace_expansion_1h_extend", 0, 11),
        ("extend_s1", "workspace_expansion_1h_extend", 7, 1),
        ("randomstart_s10", "workspace_full_coverage_randomstart_overnight", 931, 10),
        ("randomstart_s4", "workspace_full_coverage_randomstart_overnight", 123, 4),
        ("dock_noop", "dock_workspace_handoff_noop_ft_12env_raw", 7, 0),
        ("dock_default", "dock_default", 0, 0),
    ]:
        gen_resets(nm, cfgs[key], seed=seed, stage=stage, n=96 if "randomstart" in nm or "dock" in nm else 64)

    gen_eval_suites(cfgs["workspace_expansion_1h_extend"])
    gen_curriculum_tracker()
    gen_gated_score(cfgs)
    print("done")


if __name__ == "__main__":
    main()
