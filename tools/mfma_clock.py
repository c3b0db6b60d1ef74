#!/usr/bin/env python3
"""Shader clock the two matrix kernels of the optimiser step actually run at (developer tool, GPU box only; the guide's DVFS check).

Uses a library built with -DKP1_CLK_TRACE (tools/ab_build.sh clk "-DKP1_CLK_TRACE" -> build_ab/libkp1_clk.so): thread 0 of every workgroup stamps
s_memtime (shader cycles) and s_memrealtime (100 MHz) at its start and end.  Runs the bench workload's update epochs back to back for a few seconds
(graph replay, random data from a real rollout), then reads the stamps of the last launches: clock = d(cycles) / d(ticks) x 100 MHz per workgroup.

    python3 tools/mfma_clock.py [seconds]
"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from rl_brain_trainer_amd import native

native.LIB_PATH = type(native.LIB_PATH)(os.path.join(ROOT, os.environ.get("KP1_TRACE_LIB", "build_ab/libkp1_clk.so")))
from rl_brain_trainer_amd import config as kcfg
from rl_brain_trainer_amd import ppo as P
from rl_brain_trainer_amd.vec_env import ArmKinematicVecEnv

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
cfg_dict = kcfg.load_workspace_expansion_config(kcfg.builtin_config_dir() / "workspace_expansion_bigtrain.yaml")
algo = kcfg.to_algorithm_kwargs(cfg_dict)
env = ArmKinematicVecEnv(kcfg.to_env_config(cfg_dict), 4096, seed=806)
env.set_curriculum_stage(5)
pcfg = P.PPOConfig(learning_rate=algo["learning_rate"], n_steps=128, batch_size=8192, n_epochs=8, gamma=algo["gamma"], gae_lambda=algo["gae_lambda"],
                   clip_range=algo["clip_range"], ent_coef=algo["ent_coef"], seed=806, hidden=256)
ppo = P.PPO(env, pcfg, backend="hip")
ppo.collect_rollouts()
ppo.train()
ppo.collect_rollouts()
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < seconds:       # update epochs only: 512 optimiser steps per call, graph replay
    ppo.train()
    n += 1
torch.cuda.synchronize()
L = native.load()
L.kp1_debug_clk_trace.argtypes = [C.c_void_p]
buf = np.zeros(2 * 1024 * 4, dtype=np.uint64)
L.kp1_debug_clk_trace(buf.ctypes.data_as(C.c_void_p))
t = buf.reshape(2, 1024, 4).astype(np.int64)
out = {"update_calls": n, "seconds": round(time.perf_counter() - t0, 2)}
for kern, name in ((0, "mlp_tile_kernel<true>"), (1, "gemm_tn_split_kernel")):
    rows = t[kern][(t[kern][:, 3] > t[kern][:, 1])]
    ghz = (rows[:, 2] - rows[:, 0]) / ((rows[:, 3] - rows[:, 1]) * 10.0)      # cycles per 10 ns tick -> GHz
    life = (rows[:, 3] - rows[:, 1]) / 100.0
    out[name] = {"workgroups": int(len(rows)), "clock_GHz_median": round(float(np.median(ghz)), 3), "clock_GHz_p10_p90": [round(float(x), 3) for x in np.percentile(ghz, [10, 90])],
                 "workgroup_lifetime_us_median": round(float(np.median(life)), 2)}
print(json.dumps(out))
env.close()
