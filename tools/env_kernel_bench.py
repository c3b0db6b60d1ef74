#!/usr/bin/env python3
"""BASELINE configs[2] kernel-only workload for rocprofv3: 32768 stage-11 envs (workspace_expansion_1h_extend), kp1_step_kernel + fused
auto-reset on i.i.d. U(-1, 1) actions.  bench.py's config3_env_kernel block times the same launches with HIP events; this script exists so
that `rocprofv3 --kernel-trace --stats` and the --pmc passes see that kernel at that size alone.

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cfg3/stats -o env -- python3 tools/env_kernel_bench.py
    python3 tools/env_kernel_bench.py [--envs 32768] [--stage 11] [--launches 300]
"""
import argparse
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import bench  # noqa: E402
from rl_brain_trainer_amd import config as kcfg  # noqa: E402
from rl_brain_trainer_amd.vec_env import ArmKinematicVecEnv  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=32768)
ap.add_argument("--stage", type=int, default=11)
ap.add_argument("--launches", type=int, default=300)
args = ap.parse_args()
cfg = kcfg.to_env_config(kcfg.load_workspace_expansion_config(kcfg.builtin_config_dir() / "workspace_expansion_1h_extend.yaml"))
env = ArmKinematicVecEnv(cfg, args.envs, seed=806)
env.set_curriculum_stage(args.stage)
env.set_obs_stride(64)
us = bench.time_env_kernel(env, 64, launches=args.launches)
b = bench.ENV_STEP_BYTES * args.envs
print(json.dumps({"envs": args.envs, "stage": args.stage, "launch_us": us, "algorithmic_bytes": b, "achieved_GBs": b / (us * 1e-6) / 1e9,
                  "frac_of_8TBs": b / (us * 1e-6) / 1e9 / bench.PEAK_HBM_GBS}))
env.close()
