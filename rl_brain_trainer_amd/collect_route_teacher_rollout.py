"""CLI of the teacher-anchor dataset recorder (reference: ``kinematic_phase1/route/collect_route_teacher_rollout.py:126-147``, same flags).

    python -m rl_brain_trainer_amd.collect_route_teacher_rollout --checkpoint model_latest.zip --config route.yaml \
        --route-path route_q_dense.json --artifact-root /tmp/teacher --end-index 120
"""
from __future__ import annotations

import argparse
import json
from pathlib import Path

import torch

from . import route_config as rcfg
from .ppo import InferencePolicy
from .route_curriculum import collect_teacher_rollout
from .train_route import load_route_training_config


def build_arg_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Collect route teacher anchor dataset (MI355X engine).")
    p.add_argument("--checkpoint", required=True)
    p.add_argument("--config", required=True)
    p.add_argument("--route-path", required=True)
    p.add_argument("--artifact-root", required=True)
    p.add_argument("--start-index", type=int, default=1)
    p.add_argument("--end-index", type=int, default=120)
    p.add_argument("--device", type=int, default=0)
    return p


def main(argv: list[str] | None = None) -> dict:
    args = build_arg_parser().parse_args(argv)
    torch.cuda.set_device(args.device)
    model = InferencePolicy.load(args.checkpoint, device=args.device)
    summary = collect_teacher_rollout(policy=lambda obs: model.predict(obs.float().contiguous()), cfg=load_route_training_config(args.config),
                                      route_q=rcfg.load_route_q(args.route_path), artifact_root=Path(args.artifact_root), start_index=args.start_index,
                                      end_index=args.end_index, device=args.device, checkpoint=args.checkpoint, config=args.config, route_path=args.route_path)
    print(json.dumps(summary, indent=2))
    return summary


if __name__ == "__main__":
    main()
