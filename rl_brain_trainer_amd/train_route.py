"""Route-curriculum trainer on one MI355X.

Mirror of ``kinematic_phase1/train_route_curriculum.py:69-199``: same YAML chain (approach_default <- ppo_default <- overlay), same
CLI flags, same artefacts (``model_latest.zip``, ``curriculum_history.json``, ``route_eval_sequential/``, ``route_gate/``,
``model_sequential_gate_accepted.zip``, ``training_summary.json``), with the reference's three callbacks mapped onto the device
engine: periodic checkpoints, the prefix curriculum (``RoutePrefixCurriculumDevice``: a one-wave tracker after every env step, inside the
rollout hipGraph) and the teacher-anchor imitation step between rollout and update (``RouteTeacherAnchor``).  The envs are ``RouteVecEnv`` lanes (single or
sequence wrapper, 56- or 80-float observation as the YAML says); the PPO update runs on the MFMA kernels.

    python -m rl_brain_trainer_amd.train_route --config <route yaml> --route-path <route_q_dense.json> --run-id route \
        --output-dir /tmp/route --total-timesteps 1000000 --n-envs 1024
"""
from __future__ import annotations

import argparse
import json
import shutil
import time
from pathlib import Path
from typing import Any

import torch

from . import checkpoint
from . import config as kcfg
from . import route_config as rcfg
from .ppo import PPO, Dist, PPOConfig
from .route_curriculum import RoutePrefixCurriculumDevice, evaluate_route_gate, evaluate_sequential_route
from .route_env import RouteVecEnv
from .teacher_anchor import RouteTeacherAnchor, TeacherAnchorConfig


def load_route_training_config(path: str | Path | None) -> dict[str, Any]:
    """train_route_curriculum.py:41-45"""
    cfg = kcfg.deep_merge(kcfg.load_yaml_file(kcfg.builtin_config_dir() / "approach_default.yaml"), kcfg.load_yaml_file(kcfg.builtin_config_dir() / "ppo_default.yaml"))
    if path:
        cfg = kcfg.deep_merge(cfg, kcfg.load_yaml_file(Path(path)))
    return cfg


def build_arg_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Train route curriculum policy (MI355X engine).")
    p.add_argument("--config", required=True)
    p.add_argument("--route-path")
    p.add_argument("--init-checkpoint")
    p.add_argument("--run-id", default="route_curriculum")
    p.add_argument("--output-dir")
    p.add_argument("--total-timesteps", type=int)
    p.add_argument("--seed", type=int)
    p.add_argument("--n-envs", type=int, default=0, help="device environments (0 = training.n_envs of the YAML)")
    p.add_argument("--n-steps", type=int, default=0, help="rollout length (0 = the YAML's n_steps)")
    p.add_argument("--batch-size", type=int, default=0, help="minibatch (0 = the YAML's batch_size)")
    p.add_argument("--hidden", type=int, default=256)
    p.add_argument("--device", type=int, default=0)
    p.add_argument("--log-every", type=int, default=0)
    return p


def main(argv: list[str] | None = None) -> dict[str, Any]:
    args = build_arg_parser().parse_args(argv)
    cfg = load_route_training_config(args.config)
    route_cfg = cfg.get("route", {}) or {}
    route_path = Path(args.route_path or route_cfg["route_path"])
    init_checkpoint = args.init_checkpoint or route_cfg.get("init_checkpoint")
    root = Path(args.output_dir) if args.output_dir else kcfg.repo_root() / "artifacts" / "kinematic_phase1" / "route_curriculum" / args.run_id
    root.mkdir(parents=True, exist_ok=True)

    route_q = rcfg.load_route_q(route_path)
    W = int(route_q.shape[0])
    prefixes = rcfg.prefix_stages(cfg, W)
    env_cfg = kcfg.to_env_config(cfg)
    runtime_cfg = cfg.get("training", {}) or {}
    algo = kcfg.to_algorithm_kwargs(cfg, "ppo")
    if args.total_timesteps is not None:
        algo["total_timesteps"] = args.total_timesteps
    if args.seed is not None:
        algo["seed"] = args.seed
    seed = int(algo.get("seed") or 0)
    n_envs = int(args.n_envs or runtime_cfg.get("n_envs", 1))
    torch.cuda.set_device(args.device)
    env = RouteVecEnv(env_cfg, rcfg.route_config_from_dict(cfg, max_route_index=prefixes[0]), route_q, n_envs, device=args.device, seed=seed)

    total = int(algo.get("total_timesteps", 100_000))
    model_kwargs = {k: v for k, v in algo.items() if k not in ("total_timesteps", "n_steps", "batch_size")}
    n_steps = int(args.n_steps or algo.get("n_steps", 2048))
    batch = int(args.batch_size or algo.get("batch_size", 64))
    pcfg = PPOConfig.from_algo_kwargs(model_kwargs, n_steps=n_steps, batch_size=batch, hidden=checkpoint.hidden_for_run(args.hidden, init_checkpoint))
    # the prefix curriculum: a device tracker after every env step (the rollout stays one hipGraph replay)
    curriculum = RoutePrefixCurriculumDevice.from_config(cfg, W)
    ppo = PPO(env, pcfg, curriculum=curriculum, dist=Dist(), backend="hip")
    if init_checkpoint:
        # PPO.load(..., env=vec_env) + learn(reset_num_timesteps=False): weights, Adam state, step clock; the YAML's learning rate wins
        ppo.load_checkpoint(init_checkpoint, restore_timesteps=True, restore_hyperparameters=True)
        print(f"Resuming route policy from {init_checkpoint}")

    anchor = None
    anchor_cfg = TeacherAnchorConfig(**(route_cfg.get("teacher_anchor", {}) or {}))
    if anchor_cfg.enabled:
        anchor = RouteTeacherAnchor(anchor_cfg)
        anchor.on_training_start(ppo)
    checkpoint_freq = max(int(runtime_cfg.get("checkpoint_freq", 250_000)), 1)
    next_checkpoint = checkpoint_freq

    t0 = time.time()
    start_steps, it = ppo.num_timesteps, 0
    while ppo.num_timesteps - start_steps < total:
        ppo.collect_rollouts()
        if anchor is not None:
            anchor.on_rollout_end(ppo)            # BaseCallback._on_rollout_end: after the rollout, before the update
        ppo.train()
        it += 1
        if ppo.num_timesteps - start_steps >= next_checkpoint:   # PeriodicCheckpointCallback (callbacks.py)
            checkpoint.save(root / "checkpoints" / f"model_{ppo.num_timesteps - start_steps}_steps", ppo, env_cfg)
            next_checkpoint += checkpoint_freq
        if args.log_every and it % args.log_every == 0:
            s = curriculum.summary()
            print(f"[route] it={it} steps={ppo.num_timesteps} fps={(ppo.num_timesteps - start_steps) / (time.time() - t0):,.0f} prefix={s['prefix_end_index']} "
                  f"succ={s['recent_success_rate']:.3f} rew={ppo.rew_buf.mean().item():.4f} anchor={anchor.last_loss if anchor else 0.0:.5f}", flush=True)
    torch.cuda.synchronize()
    wall = time.time() - t0
    latest = root / "model_latest"
    checkpoint.save(latest, ppo, env_cfg)
    curriculum_summary = curriculum.summary()
    (root / "curriculum_history.json").write_text(json.dumps(curriculum_summary, indent=2))

    def policy(obs: torch.Tensor) -> torch.Tensor:
        return ppo.predict(obs.float().contiguous(), deterministic=True)

    def evaluate(*, artifact_root: Path, start_index: int, end_index: int) -> dict[str, Any]:
        out = evaluate_sequential_route(policy=policy, cfg=cfg, route_q=route_q, artifact_root=artifact_root, start_index=start_index, end_index=end_index,
                                        device=args.device)
        return {k: v for k, v in out.items() if k not in ("rows", "chunk_metrics", "final_q")}

    eval_end = min(int(curriculum_summary["prefix_end_index"]), W - 1)
    eval_summary = evaluate(artifact_root=root / "route_eval_sequential", start_index=1, end_index=eval_end)
    gate_summary: dict[str, Any] = {"enabled": False}
    gate_cfg = route_cfg.get("sequential_gate", {}) or {}
    if bool(gate_cfg.get("enabled", False)):
        gate_summary = evaluate_route_gate(evaluate=evaluate, artifact_root=root / "route_gate", prefixes=[int(x) for x in gate_cfg.get("prefixes", [20, 40, 80, 120, 180])],
                                           full_end_index=gate_cfg.get("full_end_index"),
                                           min_prefix120_success_rate=float(gate_cfg.get("min_prefix120_success_rate", 0.98)),
                                           best_full_longest_prefix=int(gate_cfg.get("best_full_longest_prefix", 170)),
                                           full_prefix_tolerance=int(gate_cfg.get("full_prefix_tolerance", 20)), checkpoint=str(latest), config=str(args.config),
                                           route_path=str(route_path))
        if bool(gate_summary.get("accepted", False)):
            src = Path(str(latest) + ".zip")
            dst = root / "model_sequential_gate_accepted.zip"
            if src.exists():
                shutil.copy2(src, dst)
                gate_summary["accepted_model_path"] = str(dst)
    summary = {
        "schema_version": "v5.route_curriculum.training_summary.v1", "run_id": args.run_id, "route_path": str(route_path),
        "init_checkpoint": str(init_checkpoint) if init_checkpoint else None, "checkpoint_format": {"layout": "stable-baselines3 zip", "sb3_loadable": False, "finish_with": "tools/finish_sb3_zip.py (needs stable-baselines3==2.8.0)"}, "model_path": str(latest), "n_envs": n_envs, "device": "MI355X",
        "curriculum_summary": curriculum_summary, "teacher_anchor_summary": anchor.summary() if anchor is not None else {"enabled": False},
        "route_eval_sequential_summary": eval_summary, "route_gate_summary": gate_summary, "config": cfg,
        "num_timesteps": int(ppo.num_timesteps), "wall_seconds": wall, "env_steps_per_second": (ppo.num_timesteps - start_steps) / max(wall, 1e-9),
        "observation_dim": int(ppo.obs_dim),
    }
    (root / "training_summary.json").write_text(json.dumps(summary, indent=2, default=str))
    print(json.dumps({"run_id": args.run_id, "artifact_root": str(root), "model_latest": str(latest) + ".zip", "prefix_end_index": curriculum_summary["prefix_end_index"],
                      "env_steps_per_second": summary["env_steps_per_second"]}, indent=2))
    env.close()
    return summary


if __name__ == "__main__":
    main()
