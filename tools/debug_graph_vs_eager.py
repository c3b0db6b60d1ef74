"""developer check: a captured (hipGraph) training run against the same run with every launch eager, single process; prints the first buffer that differs"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

from conftest import load_golden_config
from rl_brain_trainer_amd import ppo as P
from rl_brain_trainer_amd.curriculum import PointCurriculum
from rl_brain_trainer_amd.vec_env import ArmKinematicVecEnv

cfg = load_golden_config("workspace_expansion_bigtrain")
N, T = 192, 32


def run(use_graphs):
    env = ArmKinematicVecEnv(cfg, N, seed=806)
    cur = PointCurriculum(success_rate_threshold=0.0, window_episodes=8, min_episodes_per_stage=16, max_stage_index=11, initial_stage_index=2)
    pcfg = P.PPOConfig(n_steps=T, batch_size=1024, n_epochs=3, hidden=256, learning_rate=1e-3, seed=806, clip_range=0.1, ent_coef=3e-4)
    ppo = P.PPO(env, pcfg, curriculum=cur, backend="hip", use_graphs=use_graphs)
    snaps = []
    for it in range(3):
        ppo.collect_rollouts()
        torch.cuda.synchronize()
        snaps.append({f"it{it}.{k}": getattr(ppo, k).clone() for k in ("obs_buf", "act_buf", "logp_buf", "val_buf", "rew_buf", "done_buf", "adv_buf", "ret_buf")})
        ppo.train()
        torch.cuda.synchronize()
        snaps.append({f"it{it}.flat": ppo.policy.flat.clone(), f"it{it}.adam_m": ppo.adam_m.clone()})
    cur.close()
    env.close()
    return snaps


a, b = run(True), run(False)
for sa, sb in zip(a, b):
    for k in sa:
        same = torch.equal(sa[k], sb[k])
        d = (sa[k].double() - sb[k].double()).abs().max().item()
        print(f"{k:18s} {'same' if same else 'DIFF'}  max|d| {d:.3e}")
