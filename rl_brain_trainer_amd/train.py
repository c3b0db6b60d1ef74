"""Stage 0-11 Approach trainer on one MI355X or one 8-GPU node.

Mirror of the reference entry point kinematic_phase1/train_workspace_expansion.py:144-270 (same YAML chain, same CLI
flags, same artefact names) with the SubprocVecEnv + SB3 loop replaced by the device-resident engine:

    python -m rl_brain_trainer_amd.train --config rl_brain_trainer_amd/configs/workspace_expansion_bigtrain.yaml \
        --run-id demo --artifact-root /tmp/run --total-timesteps 2000000 --n-envs 4096
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m rl_brain_trainer_amd.train ...

Differences forced by scale, all explicit flags: ``--n-envs`` (per GPU, default 4096 instead of the YAML's 16),
``--n-steps`` / ``--batch-size`` (defaults keep the reference's 64 minibatches per epoch), ``--hidden`` (256, BASELINE
config 2; the reference never sets net_arch, i.e. SB3's 64).
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import time
from pathlib import Path
from typing import Any

import torch

from . import checkpoint
from . import config as kcfg
from .curriculum import PointCurriculum
from .ppo import PPO, Dist, PPOConfig
from .vec_env import ArmKinematicVecEnv


def build_arg_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Workspace Expansion Curriculum PPO training (MI355X engine).")
    p.add_argument("--config", required=True)
    p.add_argument("--run-id", default="workspace_expand_mi355x_001")
    p.add_argument("--artifact-root")
    p.add_argument("--total-timesteps", type=int)
    p.add_argument("--seed", type=int)
    p.add_argument("--resume-from")
    p.add_argument("--no-gate-callback", action="store_true")
    p.add_argument("--n-envs", type=int, default=4096, help="environments per GPU")
    p.add_argument("--n-steps", type=int, default=128)
    p.add_argument("--batch-size", type=int, default=0, help="global minibatch; 0 = n_envs*n_steps*world/64")
    p.add_argument("--hidden", type=int, default=256)
    p.add_argument("--learning-rate", type=float)
    p.add_argument("--log-every", type=int, default=1)
    return p


def write_json(path: Path, payload: dict[str, Any]) -> None:
    path.parent.mkdir(parents=True, exist_ok=True)
    path.write_text(json.dumps(payload, indent=2))


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

    cfg = kcfg.load_workspace_expansion_config(args.config)
    env_cfg = kcfg.to_env_config(cfg)
    algo = kcfg.to_algorithm_kwargs(cfg, "ppo")
    ws = cfg.get("workspace_expansion", {})
    if args.total_timesteps is not None:
        algo["total_timesteps"] = args.total_timesteps
    if args.seed is not None:
        algo["seed"] = args.seed
    if args.learning_rate is not None:
        algo["learning_rate"] = args.learning_rate
    seed = int(algo.get("seed", 0))

    root = Path(args.artifact_root) if args.artifact_root else kcfg.repo_root() / "artifacts/kinematic_phase1/workspace_expansion" / args.run_id
    if rank == 0:
        root.mkdir(parents=True, exist_ok=True)
        (root / "latest_checkpoint").mkdir(exist_ok=True)
        shutil.copyfile(args.config, root / "config_resolved.yaml")
        write_json(root / "training_launch_summary.json", {"run_id": args.run_id, "config": cfg})

    n_envs = args.n_envs
    env = ArmKinematicVecEnv(env_cfg, n_envs, device=local_rank, seed=seed, first_env_id=rank * n_envs)
    cur = cfg["env"].get("curriculum", {})
    curriculum = None
    if env_cfg.c.curriculum_enabled and env_cfg.n_stages:
        curriculum = PointCurriculum(success_rate_threshold=float(cur.get("success_rate_threshold", 0.80)),
                                     window_episodes=int(cur.get("window_episodes", 20)),
                                     min_episodes_per_stage=int(cur.get("min_episodes_per_stage", 30)),
                                     max_stage_index=env_cfg.n_stages - 1,
                                     initial_stage_index=int(ws.get("start_stage_index", 0)), device=local_rank)
    batch = args.batch_size or max(n_envs * args.n_steps * world // 64, 64)
    model_kwargs = {k: v for k, v in algo.items() if k not in ("total_timesteps", "n_steps", "batch_size")}
    pcfg = PPOConfig.from_algo_kwargs(model_kwargs, n_steps=args.n_steps, batch_size=batch, hidden=args.hidden)
    ppo = PPO(env, pcfg, curriculum=curriculum, dist=Dist(), backend="hip" if args.hidden in (128, 256) else "torch")
    resume = args.resume_from or ws.get("init_approach_checkpoint", "")
    if resume and Path(resume).exists():
        sd = checkpoint.load_policy_state_dict(resume)
        ppo.policy.load_state_dict(sd)
        if ppo._mlp is not None:
            ppo._mlp.pack(ppo.policy.flat)
        if rank == 0:
            print(f"Resuming workspace expansion from {resume}")

    t0 = time.time()
    total = int(algo.get("total_timesteps", 100_000))
    ppo.learn(total, log_every=args.log_every)
    torch.cuda.synchronize()
    wall = time.time() - t0
    summary: dict[str, Any] = {}
    if rank == 0:
        latest = root / "latest_checkpoint" / "model_latest"
        checkpoint.save(latest, ppo, env_cfg)
        checkpoint.save(root / "model_latest", ppo, env_cfg)
        summary = {
            "policy_type": "approach", "algorithm": "ppo", "run_id": args.run_id, "model_path": str(latest) + ".zip",
            "resume_from": str(resume) if resume else None, "n_envs": n_envs * world, "device": f"{world}x MI355X",
            "curriculum_summary": curriculum.summary() if curriculum is not None else None,
            "final_workspace_eval": None, "num_timesteps": ppo.num_timesteps, "wall_seconds": wall,
            "env_steps_per_second": ppo.num_timesteps / wall, "last_update_stats": ppo.last_stats,
        }
        write_json(root / "training_summary.json", summary)
        print(json.dumps({"run_id": args.run_id, "artifact_root": str(root), "model_latest": str(latest) + ".zip",
                          "env_steps_per_second": summary["env_steps_per_second"]}, indent=2))
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()
    return summary


if __name__ == "__main__":  # pragma: no cover
    main()
