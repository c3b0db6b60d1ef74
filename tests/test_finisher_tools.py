"""Finisher tooling: the dock reverse curriculum -- stage payloads / resolution on the host (CPU), the device tracker vs the
reference callback's golden trace (GPU); handoff buffer builder vs the serial oracle loop, and the buffer it writes feeds the dock reset
sampler (GPU)."""
from __future__ import annotations

import json

import numpy as np
import pytest

from conftest import GOLDEN, load_golden_config
from rl_brain_trainer_amd import finisher_tools as ft


def _overlay(values: dict, stage: dict) -> dict:
    """apply_dock_training_stage: a stage overrides only the keys it names (arm_kinematic_env.py:459-487)"""
    out = dict(values)
    out.update({k: v for k, v in stage.items() if k in out})
    return out


def test_dock_reverse_curriculum_stage_payloads_and_resolution():
    """Host half of the tracker: the payload of every stage is the reference callback's (golden ``calls``), and the per-stage table handed to
    the device holds what the env contains after stages 0..k were applied in order; stages that re-filter the handoff buffer are refused."""
    g = json.loads((GOLDEN / "dock_reverse_curriculum.json").read_text())
    cb = ft.DockReverseCurriculum(stages=g["stages"], window_episodes=g["window_episodes"])
    for trace in g["traces"]:
        for k, (name, payload) in enumerate(trace["calls"]):        # stage 0 at training start, then one call per promotion
            assert name == "apply_dock_training_stage" and json.loads(json.dumps(cb.stage_payload(k))) == payload
    cfg = load_golden_config("dock_workspace_handoff_noop_ft_12env_raw")
    c = cfg.c
    keys_env = ("action_delta_scale", "dock_residual_action_limit", "dock_delta_q_change_limit_scale")
    keys_dr = ("close_bucket_probability", "close_bucket_min_pos_error_m", "close_bucket_max_pos_error_m", "close_bucket_max_ori_error_rad",
               "handoff_state_probability", "close_init_q_noise", "init_q_noise")
    live = {k: float(getattr(c.env, k)) for k in keys_env}
    live.update({k: (list(getattr(c.dock_reset, k)[:]) if k.endswith("noise") else float(getattr(c.dock_reset, k))) for k in keys_dr})
    table = cb.resolved_stages(cfg)
    for k, stage in enumerate(g["stages"]):
        live = _overlay(live, stage)
        for key, want in live.items():
            got = getattr(table[k], key)
            assert (list(got[:]) == [float(v) for v in want]) if key.endswith("noise") else (float(got) == float(want)), (k, key)
        assert table[k].min_episodes == max(int(stage.get("min_episodes", g["window_episodes"])), 1)
        assert table[k].window_episodes == max(int(stage.get("window_episodes", g["window_episodes"])), 1)
        assert table[k].success_rate_threshold == float(stage.get("success_rate_threshold", 1.0))
    with pytest.raises(ValueError):
        ft.DockReverseCurriculum(stages=[], window_episodes=4)
    # the "wide" stage tightens handoff_state_max_action_l2: every stage gets its own filtered slice of the (concatenated) buffer
    cb2 = ft.DockReverseCurriculum(stages=g["stages"], window_episodes=g["window_episodes"], handoff_base_dirs=(GOLDEN,))
    buffers = cb2.stage_buffers(cfg)
    assert buffers is not None and len(buffers) == len(g["stages"])
    raw = json.loads((GOLDEN / "handoff_state_buffer.json").read_text())
    raw = raw["states"] if isinstance(raw, dict) else raw
    dr = cfg.source["env"]["dock_reset"]
    lim = {"p": dr.get("handoff_state_max_position_error_m", 1.0), "o": dr.get("handoff_state_max_orientation_error_rad", 10.0), "a": dr.get("handoff_state_max_action_l2", 10.0)}
    for stage, states in zip(g["stages"], buffers):
        lim["a"] = stage.get("handoff_state_max_action_l2", lim["a"])
        want = [it for it in raw if it.get("position_error_norm", 0.0) <= lim["p"] and it.get("orientation_error_norm", 0.0) <= lim["o"] and it.get("action_l2", 0.0) <= lim["a"]]
        assert [s_["initial_q"] for s_ in states] == [[float(v) for v in it["initial_q"]] for it in want]
    table2 = cb2.resolved_stages(cfg, buffers)
    assert [(t.handoff_offset, t.handoff_count) for t in table2] == [(sum(len(b) for b in buffers[:k]), len(buffers[k])) for k in range(len(buffers))]


@pytest.mark.gpu
def test_dock_reverse_curriculum_device_matches_reference_callback():
    """The reference callback's recorded (dones, success) stream fed to the DEVICE tracker: stage index / episode count after every step, the
    promotion history (trigger rate, episode count, clock), the final window rate -- and the device config the dock kernels read carries the
    promoted stage's values afterwards."""
    import torch

    from rl_brain_trainer_amd.vec_env import ArmKinematicVecEnv

    g = json.loads((GOLDEN / "dock_reverse_curriculum.json").read_text())
    for trace in g["traces"]:
        n = len(trace["steps"][0]["dones"])
        cfg = load_golden_config("dock_workspace_handoff_noop_ft_12env_raw")
        env = ArmKinematicVecEnv(cfg, n, seed=3)
        cb = ft.DockReverseCurriculum(stages=g["stages"], window_episodes=g["window_episodes"], handoff_base_dirs=(GOLDEN,))
        cb.attach(env)
        env.reset()
        for step in trace["steps"]:
            bits = [(2 if d else 0) | (4 if s else 0) for d, s in zip(step["dones"], step["success"])]   # success without done must be ignored
            cb.observe(torch.tensor(bits, dtype=torch.uint8, device="cuda"), n)
            st = cb.read()
            assert (int(st.stage_index), int(st.stage_episode_count)) == (step["stage"], step["count"])
        assert json.loads(json.dumps(cb.summary())) == trace["summary"]
        final = cb.resolved_stages(load_golden_config("dock_workspace_handoff_noop_ft_12env_raw"))[trace["summary"]["stage_index"]]
        assert env.config.c.dock_reset.close_bucket_probability == final.close_bucket_probability
        assert env.config.c.env.dock_residual_action_limit == final.dock_residual_action_limit
        env.reset()                                                  # resets keep working on the promoted stage's slice of the handoff buffer
        assert torch.isfinite(env.info()["position_error_norm"]).all()
        cb.close()
        env.close()


@pytest.mark.gpu
def test_handoff_buffer_builder_matches_serial_oracle_and_feeds_the_dock_sampler(tmp_path):
    import torch

    from rl_brain_trainer_amd import config as kcfg
    from rl_brain_trainer_amd import evaluate as ev
    from rl_brain_trainer_amd.vec_env import ArmKinematicVecEnv
    from test_eval_checkpoint_gpu import _serial_oracle_episode

    cfg = load_golden_config("workspace_expansion_bigtrain")
    gain, episodes, stage = 0.7, 24, 5
    holder = {}
    orig = ev.run_episodes

    def run(env, policy, opts, **kw):
        holder["env"] = env
        return orig(env, policy, opts, **kw)

    def policy(obs):
        env = holder["env"]
        dl = torch.tensor(env.config.c.joints.delta_limit[:], device="cuda", dtype=torch.float64)
        info = env.info()
        a = gain * (info["goal_q"].double().t() - info["q"].double().t()) / (dl * env.config.c.env.action_delta_scale)
        return a.clamp(-1, 1).to(env.dtype)

    ev.run_episodes = run
    try:
        out = {m: ft.build_finisher_handoff_state_buffer(approach_policy=policy, approach_cfg=cfg, artifact_root=tmp_path / m, episodes=episodes, seed=700001 + 1009 * stage,
                                                         stage_index=stage, handoff_mode=m) for m in ("final_settled", "first_confirmed", "final_always")}
    finally:
        ev.run_episodes = orig
    assert out["final_always"]["stored_handoff_count"] == episodes
    assert out["first_confirmed"]["stored_handoff_count"] > 0
    suite = ev.build_curriculum_local_eval_suite(cfg, seed=700001 + 1009 * stage, stage_index=stage, n_episodes=episodes)
    stored = {s["episode_id"]: s for s in out["first_confirmed"]["states"]}
    for e in range(episodes):
        o = {k: suite[k][e] for k in ("initial_q", "goal_q", "goal_pose6")}
        ref = _serial_oracle_episode(cfg, {**o, "policy_mode": "approach"}, gain, cfg.c.reward, 2)
        assert (ref["hand"] is not None) == (e in stored), e
        if ref["hand"] is not None:
            assert stored[e]["step_index"] == ref["hand"]["step"]
            assert np.max(np.abs(np.array(stored[e]["initial_q"]) - ref["hand"]["state"]["q"])) <= 2e-5
            assert abs(stored[e]["position_error_norm"] - ref["hand"]["pos"]) <= 2e-5
    # the written file is a valid dock_reset.handoff_state_buffer_path
    dock = json.loads((GOLDEN / "configs" / "dock_workspace_handoff_noop_ft_12env_raw.json").read_text())
    dock["env"]["dock_reset"]["handoff_state_buffer_path"] = str(tmp_path / "first_confirmed" / "finisher_handoff_state_buffer.json")
    dock["env"]["dock_reset"]["handoff_state_probability"] = 1.0
    dcfg = kcfg.to_env_config(dock)
    assert len(dcfg.handoff_states) > 0
    env = ArmKinematicVecEnv(dcfg, 64, seed=3)
    env.reset()
    goal = env.get_state()["goal_q"]
    stored_goals = np.array([s["goal_q"] for s in out["first_confirmed"]["states"]])
    assert all(np.min(np.abs(stored_goals - gq).max(axis=1)) <= 1e-6 for gq in goal)
    env.close()
