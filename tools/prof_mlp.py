"""Small driver for rocprofv3 counter passes: runs the MFMA GEMM kernels at the bench minibatch size."""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rl_brain_trainer_amd.mlp import MlpKernels
from rl_brain_trainer_amd import ppo as P
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda", 0)
pol = P.ActorCritic(256, dev, seed=1)
k = MlpKernels(256, dev, max_batch=max(n, 8192))
k.pack(pol.flat)
obs = torch.zeros((n, 64), device=dev)
obs[:, :56] = torch.rand((n, 56), device=dev) * 2 - 1
for _ in range(2):
    out = k.time_kernels(obs, n, iters=int(sys.argv[2]) if len(sys.argv) > 2 else 5)
torch.cuda.synchronize()
print({kk: (round(v["ms"] * 1e3, 1), round(v["tflops"], 1)) for kk, v in out.items()})
