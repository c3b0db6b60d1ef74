"""in-situ HIP-event duration of each optimiser-step kernel at a small minibatch (developer tool): python tools/prof_small_batch.py [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rl_brain_trainer_amd import config as kcfg
from rl_brain_trainer_amd.ppo import PPO, PPOConfig
from rl_brain_trainer_amd.vec_env import ArmKinematicVecEnv

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cfg = kcfg.to_env_config(kcfg.load_workspace_expansion_config(kcfg.builtin_config_dir() / "workspace_expansion_bigtrain.yaml"))
env = ArmKinematicVecEnv(cfg, 512, seed=1)
env.set_curriculum_stage(5)
ppo = PPO(env, PPOConfig(n_steps=32, batch_size=rows, n_epochs=1, hidden=256, seed=1), backend="hip", use_graphs=False)
ppo.collect_rollouts()
total = 32 * 512
obs = ppo.obs_buf[:32].view(total, ppo.obs_w)
act, old_logp, adv, ret = ppo.act_buf.view(total, 7), ppo.logp_buf.view(total), ppo.adv_buf.view(total), ppo.ret_buf.view(total)
perm = torch.randperm(total, device=ppo.device)
stats = ppo._epoch_adv_stats(adv, perm, total, rows)
for rep in range(3):
    ppo._mlp.set_profile(True)
    for i in range(total // rows):
        ppo._hip_minibatch_step(obs, perm[i * rows:(i + 1) * rows], act, old_logp, adv, ret, device_step=True, adv_stats=stats[i])
    torch.cuda.synchronize()
    out = ppo._mlp.profile_read()
    ppo._mlp.set_profile(False)
print(rows, {k: round(v["us"], 2) for k, v in out.items()}, "sum", round(sum(v["us"] for v in out.values()), 2))
