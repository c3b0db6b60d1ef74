import sys, json, numpy as np, torch
sys.path.insert(0, ".")
from tests.conftest import load_golden_config
from rl_brain_trainer_amd.vec_env import ArmKinematicVecEnv
real = sys.argv[1]; n = int(sys.argv[2]); steps = int(sys.argv[3]); comps = sys.argv[4] == "1"
cfg = load_golden_config("workspace_expansion_bigtrain")
env = ArmKinematicVecEnv(cfg, n, seed=806, real=real, reward_components=comps)
env.set_curriculum_stage(5)
env.reset(); torch.cuda.synchronize(); print("reset ok", flush=True)
rng = np.random.default_rng(0)
for t in range(steps):
    a = torch.tensor(rng.uniform(-1, 1, size=(n, 7)), dtype=env.dtype, device="cuda")
    env.step(a); torch.cuda.synchronize()
    if t % 16 == 0 or t > 90: print("step", t, "ok", int(env.done.sum().item()), flush=True)
print("done", real, n, flush=True)
