"""developer check: the f64 route env replay of one golden trace, printing the step index (KP1_SYNC_CHECK=1: every ABI call waits for its kernels)"""
import os
import sys

os.environ.setdefault("KP1_SYNC_CHECK", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import test_route_env_gpu as T
from rl_brain_trainer_amd import config as kcfg
from rl_brain_trainer_amd import route_config as rcfg
from rl_brain_trainer_amd.route_env import RouteVecEnv

name, cfg_name, max_index = sys.argv[1], sys.argv[2], int(sys.argv[3])
comps = len(sys.argv) < 5 or sys.argv[4] != "nocomps"
g = np.load(T.GOLDEN / f"route_trace_{name}.npz")
cfgd = T._cfg_dict(cfg_name)
route_q = rcfg.load_route_q(T.GOLDEN / "synthetic_route.json")
env = RouteVecEnv(kcfg.to_env_config(cfgd), rcfg.route_config_from_dict(cfgd, max_route_index=max_index), route_q, 1, seed=int(g["seed"]), real="f64",
                  reward_components=comps)
print("created", flush=True)
env.reset()
print("reset ok", flush=True)
Tn = int(np.sum(~np.isnan(g["reward"])))
row = 1
for t in range(Tn):
    a = torch.tensor(g["action"][row][None], dtype=torch.float64, device="cuda")
    print("step", t, "row", row, flush=True)
    o, r, d = env.step(a)
    torch.cuda.synchronize()
    d = int(d[0])
    row += 2 if (d & 3) else 1
    if row >= g["action"].shape[0]:
        break
print("done", flush=True)
