"""Finisher (dock-mode) trainer on one MI355X or one 8-GPU node.

Mirror of kinematic_phase1/training/train_dock_policy.py:39-131 (same YAML chain dock_default <- ppo_default <- overlay, same CLI
flags, same artefact names) on the device engine, plus the dock reverse curriculum the reference wires into its TD3 dock trainer
(train_dock_td3_policy.py:121-129: ``training.dock_reverse_curriculum``), running here as a device tracker inside the rollout graph.

    python -m rl_brain_trainer_amd.train_dock --config rl_brain_trainer_amd/configs/dock_workspace_handoff_noop_ft_12env.yaml \
        --run-id finisher --artifact-root /tmp/finisher --total-timesteps 2000000 --n-envs 4096
"""
from __future__ import annotations

import argparse
import json
import os
import time
from pathlib import Path
from typing import Any

import torch

from . import checkpoint
from . import config as kcfg
from .finisher_tools import DockReverseCurriculum
from .ppo import PPO, Dist, PPOConfig
from .vec_env import ArmKinematicVecEnv


def build_arg_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Train the Phase 1B dock policy (MI355X engine).")
    p.add_argument("--config")
    p.add_argument("--run-id", default="dock_policy")
    p.add_argument("--artifact-root")
    p.add_argument("--total-timesteps", type=int)
    p.add_argument("--seed", type=int)
    p.add_argument("--resume-from")
    p.add_argument("--n-envs", type=int, default=4096, help="environments per GPU")
    p.add_argument("--n-steps", type=int, default=64)
    p.add_argument("--batch-size", type=int, default=0, help="global minibatch; 0 = n_envs*n_steps*world/64")
    p.add_argument("--hidden", type=int, default=256)
    p.add_argument("--eval-episodes", type=int, default=512)
    p.add_argument("--log-every", type=int, default=1)
    return p


def evaluate_dock(ppo: PPO, env_cfg: kcfg.EnvConfig, *, episodes: int, seed: int, device: int) -> dict[str, Any]:
    """Deterministic dock evaluation on freshly sampled dock resets (eval_dock.py's summary keys that downstream scripts read)."""
    from . import evaluate as ev

    env = ArmKinematicVecEnv(env_cfg, episodes, device=device, seed=seed)
    if ppo.obs_w != 56:
        env.set_obs_stride(ppo.obs_w)
    res, _ = ev.run_episodes(env, ppo.predict, None)
    env.close()
    succ = res["success"].float()
    return {"episodes": int(episodes), "success_rate": float(succ.mean()), "mean_final_position_error": float(res["final_position_error"].mean()),
            "mean_final_orientation_error": float(res["final_orientation_error"].mean()), "mean_min_position_error": float(res["min_position_error"].mean()),
            "mean_episode_length": float(res["step_count"].float().mean())}


def main(argv: list[str] | None = None) -> dict[str, Any]:
    args = build_arg_parser().parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    cfg = kcfg.load_dock_config(args.config)
    base_dirs = (Path(args.config).parent,) if args.config else ()
    env_cfg = kcfg.to_env_config(cfg, handoff_base_dirs=base_dirs)
    algo = kcfg.to_algorithm_kwargs(cfg, "ppo")
    runtime = cfg.get("training", {}) or {}
    if args.total_timesteps is not None:
        algo["total_timesteps"] = args.total_timesteps
    if args.seed is not None:
        algo["seed"] = args.seed
    seed = int(algo.get("seed", 0))
    root = Path(args.artifact_root) if args.artifact_root else kcfg.repo_root() / "artifacts/kinematic_phase1/phase1b_dock" / args.run_id
    if rank == 0:
        root.mkdir(parents=True, exist_ok=True)

    n_envs = args.n_envs
    env = ArmKinematicVecEnv(env_cfg, n_envs, device=local_rank, seed=seed, first_env_id=rank * n_envs)
    batch = args.batch_size or max(n_envs * args.n_steps * world // 64, 64)
    model_kwargs = {k: v for k, v in algo.items() if k not in ("total_timesteps", "n_steps", "batch_size")}
    pcfg = PPOConfig.from_algo_kwargs(model_kwargs, n_steps=args.n_steps, batch_size=batch, hidden=checkpoint.hidden_for_run(args.hidden, args.resume_from))
    # training.dock_reverse_curriculum (train_dock_td3_policy.py:121-129): a device tracker after every env step, inside the rollout hipGraph
    curriculum = None
    cur_cfg = runtime.get("dock_reverse_curriculum", {}) or {}
    if bool(cur_cfg.get("enabled", False)):
        curriculum = DockReverseCurriculum(stages=list(cur_cfg.get("stages", [])), window_episodes=int(cur_cfg.get("window_episodes", 100)),
                                           handoff_base_dirs=base_dirs)
    ppo = PPO(env, pcfg, curriculum=curriculum, dist=Dist(), backend="hip")
    if args.resume_from and Path(args.resume_from).exists():
        # PPO.load(resume, env=vec_env) + learn(reset_num_timesteps=False) (train_dock_policy.py:89-102)
        ppo.load_checkpoint(args.resume_from, restore_timesteps=True, restore_hyperparameters=True)
        if rank == 0:
            print(f"Resuming dock policy from {args.resume_from}")

    total = int(algo.get("total_timesteps", 100_000))
    t0 = time.time()
    start, it = ppo.num_timesteps, 0
    while ppo.num_timesteps - start < total:
        ppo.collect_rollouts()
        ppo.train()
        it += 1
        if args.log_every and it % args.log_every == 0 and rank == 0:
            stage = curriculum.current_stage_index if curriculum is not None else -1
            print(f"[ppo-dock] it={it} steps={ppo.num_timesteps} fps={(ppo.num_timesteps - start) / (time.time() - t0):,.0f} stage={stage} "
                  f"rew={ppo.rew_buf.mean().item():.4f} {ppo.last_stats}", flush=True)
    torch.cuda.synchronize()
    wall = time.time() - t0
    summary: dict[str, Any] = {}
    if rank == 0:
        latest = root / "model_latest"
        checkpoint.save(latest, ppo, env_cfg)
        eval_summary = evaluate_dock(ppo, env_cfg, episodes=args.eval_episodes, seed=seed + 10_000, device=local_rank)
        (root / "dock_eval").mkdir(exist_ok=True)
        (root / "dock_eval" / "dock_eval_summary.json").write_text(json.dumps(eval_summary, indent=2))
        summary = {"policy_type": "dock", "algorithm": "ppo", "run_id": args.run_id, "checkpoint_format": {"layout": "stable-baselines3 zip", "sb3_loadable": False, "finish_with": "tools/finish_sb3_zip.py (needs stable-baselines3==2.8.0)"}, "config": cfg, "model_path": str(latest) + ".zip",
                   "resume_from": str(args.resume_from) if args.resume_from else None, "n_envs": n_envs * world, "device": f"{world}x MI355X",
                   "dock_eval_summary": eval_summary, "dock_reverse_curriculum": curriculum.summary() if curriculum is not None else None,
                   "num_timesteps": ppo.num_timesteps, "wall_seconds": wall, "env_steps_per_second": ppo.num_timesteps / wall}
        (root / "training_summary.json").write_text(json.dumps(summary, indent=2))
        print(json.dumps(eval_summary, indent=2))
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        ppo.dist.close()
        dist.destroy_process_group()
    return summary


if __name__ == "__main__":  # pragma: no cover
    main()
