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


class WorkspaceEvalGate:
    """WorkspaceEvalGateCallback (train_workspace_expansion.py:54-129): every ``eval_interval`` timesteps save a candidate, run the
    deterministic Approach -> Finisher stage suites on it, append the gated selection to eval_history.jsonl and keep the best
    retention-ok candidate as best_checkpoint/model_best_by_gate.zip.  The reference checks after every env step of its 12-16
    envs; here the check runs after every PPO iteration (n_envs * n_steps timesteps), so evaluations land on iteration boundaries."""

    def __init__(self, *, artifact_root: Path, approach_cfg: kcfg.EnvConfig, finisher_policy, finisher_cfg: kcfg.EnvConfig | None, eval_interval: int,
                 episodes: int, seed: int, stage_indices: list[int], gate_config: dict[str, Any], device: int = 0) -> None:
        self.artifact_root = artifact_root
        self.approach_cfg, self.finisher_policy, self.finisher_cfg = approach_cfg, finisher_policy, finisher_cfg
        self.eval_interval = max(int(eval_interval), 1)
        self.episodes = max(int(episodes), 1)
        self.seed = int(seed)
        self.stage_indices = list(stage_indices)
        self.gate_config = dict(gate_config)
        self.device = device
        self.candidates_dir = artifact_root / "gate_candidates"
        self.eval_dir = artifact_root / "gate_evals"
        self.best_dir = artifact_root / "best_checkpoint"
        self.eval_history_path = artifact_root / "eval_history.jsonl"
        for d in (self.candidates_dir, self.eval_dir, self.best_dir):
            d.mkdir(parents=True, exist_ok=True)
        self.best_score = float("-inf")
        self.next_eval_timesteps = self.eval_interval

    def on_iteration(self, ppo: PPO, env_cfg: kcfg.EnvConfig) -> dict[str, Any] | None:
        from . import evaluate as ev

        if ppo.num_timesteps < self.next_eval_timesteps:
            return None
        while self.next_eval_timesteps <= ppo.num_timesteps:
            self.next_eval_timesteps += self.eval_interval
        candidate = self.candidates_dir / f"candidate_step_{ppo.num_timesteps}"
        checkpoint.save(candidate, ppo, env_cfg)
        summary = ev.evaluate_workspace_expansion(approach_policy=ppo.predict, finisher_policy=self.finisher_policy, approach_cfg=self.approach_cfg,
                                                  finisher_cfg=self.finisher_cfg, episodes=self.episodes, seed=self.seed, stage_indices=self.stage_indices,
                                                  gate_config=self.gate_config, artifact_root=self.eval_dir / f"eval_step_{ppo.num_timesteps}",
                                                  device=self.device, obs_stride=ppo.obs_w)
        selection = summary["best_model_selection"]
        record = {"timesteps": int(ppo.num_timesteps), "candidate": str(candidate) + ".zip", **selection}
        with self.eval_history_path.open("a", encoding="utf-8") as handle:
            handle.write(json.dumps(record) + "\n")
        score = float(selection["score"])
        if bool(selection["retention_ok"]) and score > self.best_score:
            self.best_score = score
            checkpoint.save(self.best_dir / "model_best_by_gate", ppo, env_cfg)
            write_json(self.artifact_root / "best_model_selection_summary.json", record)
        return record


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
    resume = args.resume_from or ws.get("init_approach_checkpoint", "")
    hidden = checkpoint.hidden_for_run(args.hidden, resume)     # a reference-trained zip is 2x64: the model takes the checkpoint's width
    pcfg = PPOConfig.from_algo_kwargs(model_kwargs, n_steps=args.n_steps, batch_size=batch, hidden=hidden)
    ppo = PPO(env, pcfg, curriculum=curriculum, dist=Dist(), backend="hip")     # 2x64 / 2x128 / 2x256 all run on the MFMA kernels
    if resume and Path(resume).exists():
        # PPO.load(resume, env=vec_env): weights, Adam state and the saved algorithm constants; the YAML's learning rate is re-applied
        ppo.load_checkpoint(resume, restore_hyperparameters=True)
        if rank == 0:
            print(f"Resuming workspace expansion from {resume}")

    gate = None
    finisher_policy = finisher_cfg = None
    gate_cfg = dict(ws.get("gate", {}) or {})
    if ws.get("finisher_checkpoint") and Path(str(ws["finisher_checkpoint"])).exists():
        from .ppo import InferencePolicy

        finisher_policy = InferencePolicy.load(str(ws["finisher_checkpoint"]), device=local_rank)
        finisher_cfg = kcfg.to_env_config(kcfg.load_yaml_file(ws["finisher_config"]), handoff_base_dirs=(Path(str(ws["finisher_config"])).parent,))
    if rank == 0 and not args.no_gate_callback and finisher_policy is not None:
        gate = WorkspaceEvalGate(artifact_root=root, approach_cfg=env_cfg, finisher_policy=finisher_policy, finisher_cfg=finisher_cfg,
                                 eval_interval=int(ws.get("eval_interval", 200_000)), episodes=int(ws.get("gate_eval_episodes", 24)),
                                 seed=int(ws.get("eval_seed", 700001)), stage_indices=list(range(env_cfg.n_stages)), gate_config=gate_cfg,
                                 device=local_rank)

    t0 = time.time()
    total = int(algo.get("total_timesteps", 100_000))
    start_steps, it = ppo.num_timesteps, 0
    while ppo.num_timesteps - start_steps < total:     # PPO.learn, one iteration at a time so the gate can look in between
        ppo.collect_rollouts()
        ppo.train()
        it += 1
        if gate is not None:
            gate.on_iteration(ppo, env_cfg)
        if args.log_every and it % args.log_every == 0 and rank == 0:
            stage = curriculum.read().stage_index if curriculum is not None else -1
            print(f"[ppo] it={it} steps={ppo.num_timesteps} fps={(ppo.num_timesteps - start_steps) / (time.time() - t0):,.0f} stage={stage} "
                  f"rew={ppo.rew_buf.mean().item():.4f} {ppo.last_stats}", flush=True)
    torch.cuda.synchronize()
    wall = time.time() - t0
    summary: dict[str, Any] = {}
    if rank == 0:
        latest = root / "latest_checkpoint" / "model_latest"
        checkpoint.save(latest, ppo, env_cfg)
        checkpoint.save(root / "model_latest", ppo, env_cfg)
        final_eval = None
        if finisher_policy is not None:   # train_workspace_expansion.py:243-259
            from . import evaluate as ev

            final_eval = ev.evaluate_workspace_expansion(approach_policy=ppo.predict, finisher_policy=finisher_policy, approach_cfg=env_cfg,
                                                         finisher_cfg=finisher_cfg, episodes=int(ws.get("final_eval_episodes", 80)),
                                                         seed=int(ws.get("eval_seed", 700001)), stage_indices=list(range(env_cfg.n_stages)),
                                                         gate_config=gate_cfg, artifact_root=root / "final_eval", device=local_rank, obs_stride=ppo.obs_w)
            final_eval = {k: v for k, v in final_eval.items() if k != "target_rows"}
            for name in ("stage_metrics.json", "workspace_failure_report.json", "best_model_selection_summary.json"):
                if (root / "final_eval" / name).exists():
                    shutil.copyfile(root / "final_eval" / name, root / name)
        summary = {
            "policy_type": "approach", "algorithm": "ppo", "run_id": args.run_id, "checkpoint_format": {"layout": "stable-baselines3 zip", "sb3_loadable": False, "finish_with": "tools/finish_sb3_zip.py (needs stable-baselines3==2.8.0)"}, "model_path": str(latest) + ".zip",
            "resume_from": str(resume) if resume else None, "n_envs": n_envs * world, "device": f"{world}x MI355X",
            "curriculum_summary": curriculum.summary() if curriculum is not None else None,
            "final_workspace_eval": final_eval, "num_timesteps": ppo.num_timesteps, "wall_seconds": wall,
            "env_steps_per_second": ppo.num_timesteps / wall, "last_update_stats": ppo.last_stats,
        }
        write_json(root / "training_summary.json", summary)
        print(json.dumps({"run_id": args.run_id, "artifact_root": str(root), "model_latest": str(latest) + ".zip",
                          "env_steps_per_second": summary["env_steps_per_second"]}, indent=2))
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        ppo.dist.close()
        dist.destroy_process_group()
    return summary


if __name__ == "__main__":  # pragma: no cover
    main()
