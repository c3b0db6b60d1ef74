"""Per-workgroup phase timeline of gemm_nt_kernel (developer tool, GPU box only).

Builds a private copy of the library with -DKP1_NT_TRACE (thread 0 of every workgroup stamps the 100 MHz wall clock
at the phase boundaries), runs the layer-2 forward GEMM at the bench minibatch and prints, in microseconds relative to
the first workgroup's start: when workgroups start, how long the prologue, each pair of W stages and the epilogue take.

    python tools/nt_timeline.py [rows]            # layer-wise gemm_nt (the last NT launch of time_kernels = bwd dZ1)
    python tools/nt_timeline.py [rows] fused      # mlp_tile_kernel<true> (one loss_grad call)
    python tools/nt_timeline.py [rows] tnfrag     # gemm_tn_frag_kernel (the launch after it in the same loss_grad call)
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out_dir = os.path.join(ROOT, "gpurun_out")
os.makedirs(out_dir, exist_ok=True)
lib = os.path.join(out_dir, "libkp1_trace.so")
srcs = [os.path.join(ROOT, "rl_brain_trainer_amd/csrc", f) for f in ("kp1_env.hip", "kp1_ppo.hip", "kp1_mlp.hip")]
extra = [f"-D{d}" for d in os.environ.get("KP1_TRACE_DEFS", "").split() if d]   # e.g. KP1_TNF_NOMFMA, KP1_TNF_NOLOAD
if os.environ.get("KP1_TRACE_LIB"):      # a library already built with -DKP1_NT_TRACE (tools/ab_local.sh build trace=-DKP1_NT_TRACE)
    lib = os.path.join(ROOT, os.environ["KP1_TRACE_LIB"])
else:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DKP1_NT_TRACE", "-shared", "-o", lib] + extra + srcs)

import numpy as np
import torch

from rl_brain_trainer_amd import native

native.LIB_PATH = type(native.LIB_PATH)(lib)
from rl_brain_trainer_amd import ppo as P
from rl_brain_trainer_amd.mlp import MlpKernels

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda", 0)
pol = P.ActorCritic(256, dev, seed=1)
k = MlpKernels(256, dev, max_batch=max(n, 8192))
k.pack(pol.flat)
obs = torch.zeros((n, 64), device=dev)
obs[:, :56] = torch.rand((n, 56), device=dev) * 2 - 1
L = native.load()
L.kp1_debug_nt_trace.argtypes = [C.c_void_p, C.c_int]
fused = len(sys.argv) > 2 and sys.argv[2] == "fused"
tn = len(sys.argv) > 2 and sys.argv[2] == "tn"
tnfrag = len(sys.argv) > 2 and sys.argv[2] == "tnfrag"
fused = fused or tnfrag
if fused:
    act = torch.randn((n, 7), device=dev)
    z = torch.zeros(n, device=dev)
    grad = torch.empty(k.num_params, device=dev)
    for _ in range(3):
        k.loss_grad(obs, None, n, act, z - 7.0, torch.randn(n, device=dev), z, clip_range=0.2, ent_coef=0.0, vf_coef=0.5, inv_count=1.0 / n,
                    grad_out=grad, stats_out=None)
    torch.cuda.synchronize()
else:
    k.time_kernels(obs, n, iters=3)   # last launch = gemm_tn dW2
buf = np.zeros(16 * 1024, dtype=np.uint64)
# time_kernels runs layer 1 (64-deep) first, then fwd L2, then bwd: the buffer holds the LAST NT launch = bwd dZ1
L.kp1_debug_nt_trace(buf.ctypes.data, 0)
t = buf.reshape(1024, 16).astype(np.int64)
base = 12 if tnfrag else 0      # the TN-frag launch stamps slots 12..15, everything else starts at slot 0
t = t[t[:, base] > 0]
t0 = t[:, base].min()
us = (t - t0) / 100.0
print("workgroups traced:", len(t))
if tnfrag:
    # the tile kernel's stamps are overwritten by the TN launch that follows it (same buffer, block-linear index)
    hw = t[:, 11]
    cu = ((hw >> 32) & 0xF) * 64 + ((hw >> 13) & 0x7) * 16 + ((hw >> 8) & 0xF)      # (xcc, se, cu) -> one id per CU
    kinds = {}
    for b in range(min(512, len(t))):
        kinds.setdefault(int(cu[b]), []).append("dW2" if b < 256 else "dW1")
    mix = {}
    for v in kinds.values():
        mix["+".join(sorted(v))] = mix.get("+".join(sorted(v)), 0) + 1
    print("CUs used:", len(kinds), " workgroup kinds per CU:", mix)
    for nm, sel in (("dW2 workgroups", us[:256]), ("dW1 workgroups", us[256:512])):
        print(nm, "start / loop done / acc in LDS / stored (median us):", np.round(np.median(sel[:, 12:16], axis=0), 2),
              " max stored", round(float(sel[:, 15].max()), 2), " latest start", round(float(sel[:, 12].max()), 2))
    sys.exit(0)
if tn:
    names = ["start", "stage0_in_lds", "st0", "st1", "st2", "st3", "st4", "st5", "st6", "st7", "st8+", "acc_in_lds", "stores_issued"]
    for i, nm in enumerate(names):
        if (t[:, i] == 0).all():
            continue
        col = us[:, i]
        print(f"{nm:>18}: min {col.min():7.2f}  median {np.median(col):7.2f}  max {col.max():7.2f}")
    sys.exit(0)
if fused:
    names = ["start", "prologue_done", "h1_in_lds", "h2_in_lds", "heads_done", "dz1_done", "stores_issued"]
    for i, nm in enumerate(names):
        col = us[:, i]
        print(f"{nm:>18}: min {col.min():7.2f}  median {np.median(col):7.2f}  max {col.max():7.2f}")
    for zz in (0, 1):
        sel = us[zz * (len(us) // 2):(zz + 1) * (len(us) // 2)]
        d = np.diff(sel[:, :7], axis=1)
        print(f"net {zz}: per-phase medians (prologue, G1+tanh, G2+tanh, heads, G3+dtanh, store):", np.round(np.median(d, axis=0), 2),
              " first-round workgroups:", int((sel[:, 0] < 1.0).sum()))
    if "KP1_TR_HEADS" in os.environ.get("KP1_TRACE_DEFS", ""):
        hd = us[:, [3, 7, 8, 9, 10, 4]]
        for zz in (0, 1):
            sel = hd[zz * (len(us) // 2):(zz + 1) * (len(us) // 2)]
            print(f"net {zz} head phase medians (dot products + barrier, loss math, barrier, dZ2 + head grads, barrier):", np.round(np.median(np.diff(sel, axis=1), axis=0), 2))
    elif (t[:, 7] > 0).any():
        g2 = us[:, [2] + list(range(7, 11))]     # slots 11.. are overwritten by the TN-frag launch that follows
        print("G2 first stages, medians (wave 0 of each workgroup, us):", np.round(np.median(np.diff(g2, axis=1), axis=0), 2))
    sys.exit(0)
names = ["start", "prologue_done"] + [f"stages_{2 * i}-{2 * i + 1}_done" for i in range(8)] + ["acc_in_lds", "stores_issued"]
for i, nm in enumerate(names):
    col = us[:, i]
    if (t[:, i] == 0).all():
        continue
    print(f"{nm:>18}: min {col.min():7.2f}  median {np.median(col):7.2f}  max {col.max():7.2f}")
d = np.diff(us[:, [0, 1, 2, 3, 4, 5, 10, 11]], axis=1)
print("per-phase medians (prologue, st01, st23, st45, st67, acc->lds, epilogue):", np.round(np.median(d, axis=0), 2))
print("late starters (start > 1 us):", int((us[:, 0] > 1.0).sum()))
