#!/usr/bin/env python3
"""Per-wave phase timeline of kp1_step_kernel<float, approach> (developer tool, GPU box only).

Uses a library built with -DKP1_ENV_TRACE (tools/ab_build.sh trace "-DKP1_ENV_TRACE" -> build_ab/libkp1_trace.so): lane 0 of every wave stamps the
shader clock at the phase boundaries.  Prints the median / min / max over waves of each phase in shader cycles and, with the clock measured from
the launch time, in microseconds.

    python3 tools/env_timeline.py [envs] [stage]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from rl_brain_trainer_amd import native

native.LIB_PATH = type(native.LIB_PATH)(os.path.join(ROOT, os.environ.get("KP1_TRACE_LIB", "build_ab/libkp1_trace.so")))
from rl_brain_trainer_amd import config as kcfg
from rl_brain_trainer_amd.vec_env import ArmKinematicVecEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
stage = int(sys.argv[2]) if len(sys.argv) > 2 else 11
cfg = kcfg.to_env_config(kcfg.load_workspace_expansion_config(kcfg.builtin_config_dir() / "workspace_expansion_1h_extend.yaml"))
env = ArmKinematicVecEnv(cfg, n, seed=806)
env.set_curriculum_stage(stage)
env.set_obs_stride(64)
dev = env.device
obs = torch.zeros((n, 64), device=dev)
tobs = torch.zeros_like(obs)
rew = torch.zeros(n, device=dev)
done = torch.zeros(n, dtype=torch.uint8, device=dev)
acts = torch.rand((8, n, 7), device=dev) * 2 - 1
env.use_current_stream()
env.reset()
L = native.load()
L.kp1_debug_env_trace.argtypes = [C.c_void_p, C.c_int]
for k in range(20):
    env.step_into(acts[k % 8], obs, rew, done, tobs, True)
torch.cuda.synchronize()
L.kp1_debug_env_trace(None, 1)
st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
st.record()
env.step_into(acts[0], obs, rew, done, tobs, True)
en.record()
torch.cuda.synchronize()
SLOTS, WAVES = 16, 1024
buf = np.zeros(SLOTS * WAVES, dtype=np.uint64)
L.kp1_debug_env_trace(buf.ctypes.data_as(C.c_void_p), 0)
t = buf.reshape(WAVES, SLOTS)[: min(WAVES, (n + 63) // 64)].astype(np.int64)
t = t[t[:, 0] > 0]
names = ["entry -> cfg warm-up issued+landed", "state loads -> prev pose error", "action / q integration", "FK (fp64 chain) + Euler", "pose error, counters, termination",
         "reward", "observation build", "reward/done/state/obs stores issued", "stores acknowledged"]
print(f"{n} envs, stage {stage}: {t.shape[0]} waves traced; host-timed launch {st.elapsed_time(en) * 1e3:.1f} us")
# slots 11 / 12: the chip-wide 100 MHz clock at wave entry / exit (10 ns ticks); the shader clock of the phase stamps is per XCD
t0r, t1r = t[:, 11], t[:, 12]
print(f"chip clock: first wave start -> last wave START {(t0r.max() - t0r.min()) * 0.01:.2f} us, -> last wave END {(t1r.max() - t0r.min()) * 0.01:.2f} us; "
      f"wave lifetime median {np.median(t1r - t0r) * 0.01:.2f} us ({np.median(t[:, 9] - t[:, 0]):.0f} shader cycles); start-time deciles (us) "
      f"{np.round(np.percentile(t0r - t0r.min(), [10, 30, 50, 70, 90]) * 0.01, 2)}")
for k, nm in enumerate(names):
    d = t[:, k + 1] - t[:, k]
    print(f"  {nm:46s} median {np.median(d):8.0f}  min {d.min():8d}  max {d.max():8d} cycles")
if (t[:, 10] > 0).all():   # slot 10 = end of step_env_lane (reward / done / state stores issued), before the observation tile store
    a_, b_ = t[:, 10] - t[:, 7], t[:, 8] - t[:, 10]
    print(f"  of the store phase: reward/done/state stores {np.median(a_):6.0f}, observation tile (LDS transpose + 16 stores) {np.median(b_):6.0f} cycles")
env.close()
