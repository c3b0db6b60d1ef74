"""CPU restatement of the PPO side of the hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/ and bench.py's cpu_baseline leg import this module; the product package never does.

What it restates: stable-baselines3 2.8.0 ``PPO`` with ``MultiInputPolicy`` as the reference configures it
(``training/train_ppo.py``, ``configs/ppo_default.yaml``): CombinedExtractor (flatten, sorted keys) -> separate tanh MLPs
``mlp_extractor.policy_net`` / ``value_net`` -> ``action_net`` / ``value_net`` heads, state-independent ``log_std``;
``collect_rollouts`` (sample, clip for the env, log-prob, value), GAE(lambda) with the time-limit bootstrap, and the ``train``
inner loop (advantage normalisation per minibatch, clipped surrogate, value MSE, entropy bonus, ``clip_grad_norm_``, Adam).
SB3 is a pip dependency of the reference, absent from /root/reference and from this image: **parity unpinned** -- the
semantics follow SB3's published source for that version, nothing here was checked against SB3 itself.

Plain torch on CPU tensors (fp32, like the reference's own CPU runs with ``device: cpu``), so bench.py can time the whole loop --
env (oracle/kp1_oracle.c) + this -- on the host cores of the GPU box.
"""
from __future__ import annotations

import math

import torch
from torch import nn

LOG_SQRT_2PI = 0.5 * math.log(2.0 * math.pi)


class ActorCriticCPU(nn.Module):
    def __init__(self, obs_dim: int = 56, hidden: int = 256, act_dim: int = 7, seed: int = 0) -> None:
        super().__init__()
        torch.manual_seed(seed)
        self.policy_net = nn.Sequential(nn.Linear(obs_dim, hidden), nn.Tanh(), nn.Linear(hidden, hidden), nn.Tanh())
        self.value_net = nn.Sequential(nn.Linear(obs_dim, hidden), nn.Tanh(), nn.Linear(hidden, hidden), nn.Tanh())
        self.action_head = nn.Linear(hidden, act_dim)
        self.value_head = nn.Linear(hidden, 1)
        self.log_std = nn.Parameter(torch.zeros(act_dim))
        for mod, gain in ((self.policy_net, math.sqrt(2.0)), (self.value_net, math.sqrt(2.0)), (self.action_head, 0.01), (self.value_head, 1.0)):
            for m in mod.modules():
                if isinstance(m, nn.Linear):
                    nn.init.orthogonal_(m.weight, gain=gain)
                    nn.init.zeros_(m.bias)

    def forward(self, obs: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        return self.action_head(self.policy_net(obs)), self.value_head(self.value_net(obs)).squeeze(-1)

    def log_prob(self, actions: torch.Tensor, mean: torch.Tensor) -> torch.Tensor:
        z = (actions - mean) * torch.exp(-self.log_std)
        return (-0.5 * z * z - self.log_std - LOG_SQRT_2PI).sum(-1)

    def entropy(self) -> torch.Tensor:
        return (0.5 + LOG_SQRT_2PI + self.log_std).sum()


@torch.no_grad()
def rollout_step(model: ActorCriticCPU, obs: torch.Tensor, gen: torch.Generator | None = None):
    """policy.forward of collect_rollouts: (action, clipped action for the env, value, log_prob)"""
    mean, value = model(obs)
    noise = torch.randn(mean.shape, generator=gen)
    action = mean + torch.exp(model.log_std) * noise
    return action, action.clamp(-1.0, 1.0), value, model.log_prob(action, mean)


def gae(rewards: torch.Tensor, values: torch.Tensor, dones: torch.Tensor, last_values: torch.Tensor, gamma: float, lam: float):
    """RolloutBuffer.compute_returns_and_advantage: rewards/values/dones [T, N] (dones[t] = episode ended AT step t)."""
    T = rewards.shape[0]
    adv = torch.zeros_like(rewards)
    last = torch.zeros_like(last_values)
    for t in reversed(range(T)):
        nonterminal = 1.0 - dones[t].float()
        next_v = last_values if t == T - 1 else values[t + 1]
        delta = rewards[t] + gamma * next_v * nonterminal - values[t]
        last = delta + gamma * lam * nonterminal * last
        adv[t] = last
    return adv, adv + values


def train_minibatch(model: ActorCriticCPU, opt: torch.optim.Optimizer, obs, actions, old_log_prob, advantages, returns, *, clip_range: float,
                    ent_coef: float, vf_coef: float, max_grad_norm: float, normalize_advantage: bool = True) -> dict[str, float]:
    """One optimiser step of PPO.train's inner loop."""
    if normalize_advantage and len(advantages) > 1:
        advantages = (advantages - advantages.mean()) / (advantages.std() + 1e-8)
    mean, values = model(obs)
    log_prob = model.log_prob(actions, mean)
    ratio = torch.exp(log_prob - old_log_prob)
    policy_loss = -torch.min(advantages * ratio, advantages * torch.clamp(ratio, 1 - clip_range, 1 + clip_range)).mean()
    value_loss = torch.nn.functional.mse_loss(returns, values)
    entropy_loss = -model.entropy()
    loss = policy_loss + ent_coef * entropy_loss + vf_coef * value_loss
    opt.zero_grad()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), max_grad_norm)
    opt.step()
    return {"policy_loss": float(policy_loss.detach()), "value_loss": float(value_loss.detach()), "entropy": float(-entropy_loss.detach())}


def flat_params_sb3_order(model: ActorCriticCPU) -> torch.Tensor:
    """Parameters in SB3 state_dict order (log_std, policy_net.0/2, value_net.0/2, action_net, value_net) = the engine's flat vector."""
    parts = [model.log_std, model.policy_net[0].weight, model.policy_net[0].bias, model.policy_net[2].weight, model.policy_net[2].bias,
             model.value_net[0].weight, model.value_net[0].bias, model.value_net[2].weight, model.value_net[2].bias, model.action_head.weight,
             model.action_head.bias, model.value_head.weight, model.value_head.bias]
    return torch.cat([p.detach().reshape(-1) for p in parts])
