#!/usr/bin/env python3
"""Round-3 EXPERIMENT, not the product path: the weight-gradient GEMMs on bf16 x 3 operands (KP1_MLP_OPT_BF16X3_WGRAD) against the exact fp32 MFMA kernel.

On one real minibatch of the bench workload (4096 stage-5 envs x 128 steps, 8192 rows, 2x256): the flat gradient of the exact path, of the experiment
and of an fp64 torch-autograd restatement of SB3's loss; per-tensor max |g - ref64| / max |ref64| for both kernels and max |g_bf16x3 - g_exact|;
then the in-situ HIP-event time of every optimiser-step kernel with the option off and on.  Prints one JSON object.

    python3 tools/bf16x3_experiment.py
"""
import json
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rl_brain_trainer_amd import config as kcfg
from rl_brain_trainer_amd import ppo as P
from rl_brain_trainer_amd.vec_env import ArmKinematicVecEnv

cfg_dict = kcfg.load_workspace_expansion_config(kcfg.builtin_config_dir() / "workspace_expansion_bigtrain.yaml")
algo = kcfg.to_algorithm_kwargs(cfg_dict)
env = ArmKinematicVecEnv(kcfg.to_env_config(cfg_dict), 4096, seed=806)
env.set_curriculum_stage(5)
pcfg = P.PPOConfig(learning_rate=algo["learning_rate"], n_steps=128, batch_size=8192, n_epochs=1, gamma=algo["gamma"], gae_lambda=algo["gae_lambda"],
                   clip_range=algo["clip_range"], ent_coef=algo["ent_coef"], seed=806, hidden=256)
ppo = P.PPO(env, pcfg, backend="hip", use_graphs=False)
for _ in range(3):                      # a few updates so that the policy is not at its initialisation (ratios spread, clipping occurs)
    ppo.collect_rollouts()
    ppo.train()
ppo.collect_rollouts()
T, N = 128, 4096
total = T * N
obs = ppo.obs_buf[:T].view(total, ppo.obs_w)
act, old_logp, adv, ret = ppo.act_buf.view(total, 7), ppo.logp_buf.view(total), ppo.adv_buf.view(total), ppo.ret_buf.view(total)
perm = torch.randperm(total, device=ppo.device, generator=torch.Generator(device=ppo.device).manual_seed(3))
idx = perm[:8192]
stats = ppo._epoch_adv_stats(adv, perm, total, 8192)[0]
k = ppo._mlp


def grad(bf16: bool) -> torch.Tensor:
    k.set_bf16x3_wgrad(bf16)
    g = torch.empty(k.num_params, device=ppo.device)
    k.loss_grad(obs, idx, 8192, act, old_logp, adv, ret, clip_range=pcfg.clip_range, ent_coef=pcfg.ent_coef, vf_coef=pcfg.vf_coef, inv_count=1.0 / 8192,
                grad_out=g, stats_out=None, adv_stats=stats)
    torch.cuda.synchronize()
    return g.clone()


g_exact, g_bf = grad(False), grad(True)
g_bf2 = grad(True)
# fp64 reference of the same minibatch (SB3's loss through torch autograd)
flat = ppo.policy.flat.detach().double().clone().requires_grad_(True)
Pv, off = {}, 0
for name, shape in ppo.policy.spec:
    cnt = math.prod(shape)
    Pv[name] = flat[off:off + cnt].view(shape)
    off += cnt
o64 = obs[idx, :56].double()
mean, value = P.mlp_forward(Pv, o64)
logp = P.gaussian_log_prob(act[idx].double(), mean, Pv["log_std"])
a = (adv[idx].double() - stats[0].double()) * stats[1].double()
ratio = torch.exp(logp - old_logp[idx].double())
c = pcfg.clip_range
pl = -torch.min(a * ratio, a * torch.clamp(ratio, 1 - c, 1 + c)).mean()
vl = torch.nn.functional.mse_loss(ret[idx].double(), value)
ent = (0.5 + 0.5 * math.log(2 * math.pi) + Pv["log_std"]).sum()
(ref,) = torch.autograd.grad(pl + pcfg.vf_coef * vl - pcfg.ent_coef * ent, flat)
table, off = {}, 0
for name, shape in ppo.policy.spec:
    cnt = math.prod(shape)
    r = ref[off:off + cnt]
    scale = r.abs().max().item() + 1e-300
    table[name] = {"scale": scale, "exact_vs_f64": (g_exact[off:off + cnt].double() - r).abs().max().item() / scale,
                   "bf16x3_vs_f64": (g_bf[off:off + cnt].double() - r).abs().max().item() / scale,
                   "bf16x3_vs_exact": (g_bf[off:off + cnt] - g_exact[off:off + cnt]).abs().max().item() / scale}
    off += cnt
wg = [n for n in table if n.endswith(".weight") and ("mlp_extractor" in n)]       # the tensors the experiment computes differently


def step_times(bf16: bool) -> dict:
    k.set_bf16x3_wgrad(bf16)
    out = {}
    for _rep in range(3):
        k.set_profile(True)
        for i in range(64):
            ppo._hip_minibatch_step(obs, perm[i * 8192:(i + 1) * 8192], act, old_logp, adv, ret, device_step=True, adv_stats=stats)
        torch.cuda.synchronize()
        out = k.profile_read()
        k.set_profile(False)
    return {kk: round(v["us"], 2) for kk, v in out.items()}


t_exact, t_bf = step_times(False), step_times(True)
k.set_bf16x3_wgrad(False)
print(json.dumps({
    "experiment": "bf16x3 weight-gradient GEMMs (KP1_MLP_OPT_BF16X3_WGRAD), NOT the measured product path",
    "arithmetic": "operands split by the tile kernel into hi + mid + lo bf16 pieces; lo*hi + hi*lo + mid*mid + hi*mid + mid*hi + hi*hi on v_mfma_f32_32x32x16_bf16, fp32 accumulate",
    "minibatch": "8192 rows of a stage-5 rollout after 3 PPO iterations, 2x256",
    "reproducible_bitwise": bool(torch.equal(g_bf, g_bf2)),
    "max_rel_error_weight_tensors": {"exact_vs_f64": max(table[n]["exact_vs_f64"] for n in wg), "bf16x3_vs_f64": max(table[n]["bf16x3_vs_f64"] for n in wg),
                                     "bf16x3_vs_exact": max(table[n]["bf16x3_vs_exact"] for n in wg)},
    "per_tensor": table,
    "optimizer_step_kernels_us_in_situ": {"exact": t_exact, "bf16x3": t_bf, "exact_sum": round(sum(t_exact.values()), 2), "bf16x3_sum": round(sum(t_bf.values()), 2)},
}))
env.close()
