#!/usr/bin/env python3
"""Approach training as a CHAIN of fine-tuning phases on one MI355X, each resumed from the previous one (weights + Adam state), with a
deterministic evaluation after every phase -- the shape of the reference's own schedule (approach_default -> finisher-ready fine-tunes ->
workspace_expansion_bigtrain -> ... each `init_approach_checkpoint` = the previous run's zip, each with a lower learning rate).

    python tools/train_chain.py <chain.json | inline JSON list> <out.json> [--save /path/prefix]

A phase is {"config": builtin yaml stem, "steps": env steps, "lr": ..., "epochs": ..., "batch": rows per minibatch, "clip": ..., "ent": ...,
"start_stage": k, "max_stage": k, "n_steps": rollout length, "gamma": ..., "label": "..."}; omitted keys keep the previous phase's value (first phase:
the YAML's algorithms.ppo).  After each phase: 200 held-out episodes per stage 0..5 with the deterministic policy (eval_workspace_expansion.py:86-211
protocol, approach only), success = the config's own termination criterion, plus the finisher-ready rate the handoff needs.
"""
from __future__ import annotations

import json
import os
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from rl_brain_trainer_amd import checkpoint  # noqa: E402
from rl_brain_trainer_amd import config as kcfg  # noqa: E402
from rl_brain_trainer_amd import evaluate as ev  # noqa: E402
from rl_brain_trainer_amd.curriculum import PointCurriculum  # noqa: E402
from rl_brain_trainer_amd.ppo import PPO, PPOConfig  # noqa: E402
from rl_brain_trainer_amd.vec_env import ArmKinematicVecEnv  # noqa: E402

N_ENVS = int(os.environ.get("KP1_ENVS", "4096"))
EVAL_STAGES = list(range(6))


def load_config(stem: str) -> dict:
    if stem == "approach_default":
        cfg = kcfg.deep_merge(kcfg.load_yaml_file(kcfg.builtin_config_dir() / "approach_default.yaml"), kcfg.load_yaml_file(kcfg.builtin_config_dir() / "ppo_default.yaml"))
    else:
        cfg = kcfg.load_workspace_expansion_config(kcfg.builtin_config_dir() / f"{stem}.yaml")
    cfg.setdefault("workspace_expansion", {}).setdefault("gate", {})
    return cfg


def evaluate(ppo: PPO, env_cfg, cfg_dict, episodes: int = 200) -> dict:
    res = ev.evaluate_workspace_expansion(approach_policy=ppo.predict, finisher_policy=None, approach_cfg=env_cfg, finisher_cfg=None, episodes=episodes,
                                          seed=700001, stage_indices=[s for s in EVAL_STAGES if s < env_cfg.n_stages], gate_config=cfg_dict["workspace_expansion"]["gate"],
                                          obs_stride=ppo.obs_w)
    keep = ("success_rate", "finisher_ready_hit_rate", "dwell_success_rate", "mean_final_position_error", "mean_final_orientation_error")
    return {k: {m: round(float(v[m]), 5) for m in keep if m in v} for k, v in res["stage_metrics"].items()}


def main() -> None:
    spec = sys.argv[1]
    phases = json.loads(Path(spec).read_text()) if os.path.exists(spec) else json.loads(spec)
    out_path = Path(sys.argv[2])
    save_prefix = sys.argv[sys.argv.index("--save") + 1] if "--save" in sys.argv else None
    init = sys.argv[sys.argv.index("--init") + 1] if "--init" in sys.argv else None
    tmp = Path(tempfile.mkdtemp())
    state: dict = {}
    ppo = env = curriculum = None
    cur_config = None
    report = []
    t_all = time.time()
    for k, ph in enumerate(phases):
        state.update(ph)
        stem = state["config"]
        if stem != cur_config or ph.get("rebuild"):
            prev_zip = None
            if ppo is not None:
                prev_zip = checkpoint.save(tmp / f"phase{k - 1}", ppo, env_cfg)
                env.close()
            cfg_dict = load_config(stem)
            if "max_stage" in state:
                cfg_dict["env"]["curriculum"]["stages"] = cfg_dict["env"]["curriculum"]["stages"][: int(state["max_stage"]) + 1]
            for dotted, value in (state.get("override") or {}).items():      # a bridge phase: the named config with a few reward keys changed
                node = cfg_dict
                *parents, leaf = dotted.split(".")
                for p in parents:
                    node = node[p]
                if leaf not in node:
                    raise KeyError(f"override {dotted}: no such key in {stem}")
                node[leaf] = value
            env_cfg = kcfg.to_env_config(cfg_dict)
            algo = kcfg.to_algorithm_kwargs(cfg_dict)
            cur = cfg_dict["env"]["curriculum"]
            seed = int(state.get("seed", algo.get("seed", 0)))
            env = ArmKinematicVecEnv(env_cfg, N_ENVS, seed=seed)
            curriculum = PointCurriculum(success_rate_threshold=float(cur.get("success_rate_threshold", 0.8)), window_episodes=int(cur.get("window_episodes", 20)),
                                         min_episodes_per_stage=int(cur.get("min_episodes_per_stage", 30)), max_stage_index=env_cfg.n_stages - 1,
                                         initial_stage_index=int(state.get("start_stage", 0)), device=0)
            pcfg = PPOConfig(learning_rate=float(state.get("lr", algo["learning_rate"])), n_steps=int(state.get("n_steps", 128)), batch_size=int(state.get("batch", 8192)),
                             n_epochs=int(state.get("epochs", algo["n_epochs"])), gamma=float(state.get("gamma", algo["gamma"])), gae_lambda=float(algo["gae_lambda"]),
                             clip_range=float(state.get("clip", algo["clip_range"])), ent_coef=float(state.get("ent", algo.get("ent_coef", 0.0))),
                             seed=seed, hidden=int(state.get("hidden", 256)))
            ppo = PPO(env, pcfg, curriculum=curriculum, backend="hip")
            src = prev_zip or init
            if src:
                print("resumed:", ppo.load_checkpoint(str(src)), flush=True)
            cur_config = stem
        if "set_log_std" in ph:
            # exploration reset between fine-tuning phases (SB3: model.policy.log_std.data.fill_(v)): a collapsed state-independent std cannot
            # discover a bonus zone it has never entered; its Adam moments are cleared with it
            lo = 0
            ppo.policy.views["log_std"].fill_(float(ph["set_log_std"]))
            ppo.adam_m[lo:lo + 7].zero_()
            ppo.adam_v[lo:lo + 7].zero_()
            ppo._mlp.pack(ppo.policy.flat)
            print("log_std set to", float(ph["set_log_std"]), flush=True)
        else:
            # same env: only the hyper-parameters move (the update graph is re-captured when they change)
            ppo.cfg.learning_rate = float(state.get("lr", ppo.cfg.learning_rate))
            ppo.cfg.n_epochs = int(state.get("epochs", ppo.cfg.n_epochs))
            ppo.cfg.clip_range = float(state.get("clip", ppo.cfg.clip_range))
            ppo.cfg.ent_coef = float(state.get("ent", ppo.cfg.ent_coef))
            if int(state.get("batch", ppo.cfg.batch_size)) != ppo.cfg.batch_size:
                ppo.cfg.batch_size = int(state["batch"])
        steps = int(float(ph["steps"]))
        start = ppo.num_timesteps
        t0 = time.time()
        it = 0
        log = []
        while ppo.num_timesteps - start < steps:
            ppo.collect_rollouts()
            ppo.train()
            it += 1
            if it % max(int(state.get("log_every", 50)), 1) == 0:
                st = curriculum.read()
                done = ppo.done_buf
                rec = {"it": it, "steps": ppo.num_timesteps - start, "wall_s": round(time.time() - t0, 1), "stage": int(st.stage_index),
                       "rollout_success": round(float((((done & 4) != 0) & ((done & 3) != 0)).sum()) / max(float(((done & 3) != 0).sum()), 1.0), 4),
                       "success_step_frac": round(float(((done & 4) != 0).float().mean()), 4),
                       "log_std_mean": round(float(ppo.policy.views["log_std"].mean()), 4),
                       **{m: round(v, 5) if isinstance(v, float) else v for m, v in ppo.last_stats.items()}}
                log.append(rec)
                print(json.dumps(rec), flush=True)
        torch.cuda.synchronize()
        wall = time.time() - t0
        metrics = evaluate(ppo, env_cfg, cfg_dict)
        entry = {"phase": k, "label": ph.get("label", stem), "config": stem, "hyper": {"lr": ppo.cfg.learning_rate, "epochs": ppo.cfg.n_epochs, "batch": ppo.cfg.batch_size,
                                                                                         "clip": ppo.cfg.clip_range, "ent": ppo.cfg.ent_coef, "gamma": ppo.cfg.gamma},
                 "env_steps": ppo.num_timesteps - start, "wall_s": round(wall, 1), "env_steps_per_s": round((ppo.num_timesteps - start) / max(wall, 1e-9)),
                 "final_stage": int(curriculum.read().stage_index), "log_std_mean": float(ppo.policy.views["log_std"].mean()), "eval": metrics, "log": log[-6:]}
        report.append(entry)
        print("EVAL", json.dumps({"phase": k, "label": entry["label"], "stage5": metrics.get("5"), "stage0": metrics.get("0"), "log_std": entry["log_std_mean"],
                                  "wall_s": entry["wall_s"]}), flush=True)
        if save_prefix:
            print("saved", checkpoint.save(f"{save_prefix}_phase{k}", ppo, env_cfg), flush=True)
        out_path.parent.mkdir(parents=True, exist_ok=True)
        out_path.write_text(json.dumps({"phases": report, "total_wall_s": round(time.time() - t_all, 1), "n_envs": N_ENVS}, indent=1))
    print("done", round(time.time() - t_all, 1), "s")


if __name__ == "__main__":
    main()
