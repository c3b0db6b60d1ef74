"""Teacher-anchor imitation step of the route trainer (reference: ``kinematic_phase1/route/teacher_anchor.py:15-102``).

After every ``every_rollouts``-th rollout the reference takes ``gradient_steps`` optimiser steps on
``loss_weight * mse(policy._predict(obs, deterministic=True), teacher_actions)`` over a batch drawn with
``default_rng(0).integers(0, M, size=batch_size)`` from a recorded teacher dataset, clipping the gradient norm at 0.5, with the
policy's own Adam optimiser.  Only the tensors the loss reaches get a gradient (the policy MLP and ``action_net``); torch.optim.Adam
skips the rest, so those tensors' per-tensor step counts run ahead of the others -- mirrored here by ``actor_extra_steps``, which
the HIP Adam kernel adds to the common step count for exactly those tensors (KP1_MLP_OPT_ACTOR_EXTRA_STEPS).

The step itself is 256 rows once per rollout: it runs as a handful of torch ops on the device-resident flat parameter vector and its
Adam moments (no host copy), then the kernel-format weights are repacked.  SB3 is absent from this image, so the SB3 side is
restated from its published semantics ('parity unpinned'); tests compare the update with torch.optim.Adam on the same tensors.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from pathlib import Path
from typing import Any

import numpy as np
import torch

from . import config as kcfg
from . import route_config as rcfg


@dataclass(frozen=True)
class TeacherAnchorConfig:
    enabled: bool = False
    dataset_path: str = ""
    loss_weight: float = 0.02
    batch_size: int = 256
    gradient_steps: int = 1
    every_rollouts: int = 1
    max_route_index: int = 120


ACTOR_TENSORS = ("mlp_extractor.policy_net.0.weight", "mlp_extractor.policy_net.0.bias", "mlp_extractor.policy_net.2.weight",
                 "mlp_extractor.policy_net.2.bias", "action_net.weight", "action_net.bias")


def flatten_observation(obs: dict[str, np.ndarray], obs_dim: int) -> np.ndarray:
    """Dict observation -> the flat row SB3's CombinedExtractor feeds the MLP (keys in sorted order = the layout tables)."""
    layout = rcfg.ROUTE_OBS_LAYOUT if obs_dim == rcfg.ROUTE_OBS_DIM else kcfg.OBS_LAYOUT
    missing = [k for k in layout if k not in obs]
    if missing:
        raise ValueError(f"teacher-anchor dataset lacks observation keys {missing}")
    n = len(next(iter(obs.values())))
    flat = np.zeros((n, obs_dim), dtype=np.float32)
    for key, (off, width) in layout.items():
        flat[:, off:off + width] = np.asarray(obs[key], dtype=np.float32).reshape(n, width)
    return flat


def load_anchor_dataset(path: str | Path, max_route_index: int, obs_dim: int) -> tuple[np.ndarray, np.ndarray]:
    """``teacher_route_anchor_dataset.npz`` (collect_route_teacher_rollout) -> (flat observations [M, obs_dim], actions [M, 7]) of the
    samples whose waypoint lies inside the protected prefix."""
    with np.load(Path(path), allow_pickle=False) as data:
        inside = np.asarray(data["route_index"], dtype=np.int32) <= max_route_index
        actions = np.asarray(data["actions"], dtype=np.float32)[inside]
        keyed = {name[len("obs__"):]: np.asarray(data[name], dtype=np.float32)[inside] for name in data.files if name.startswith("obs__")}
    if not len(actions):
        raise ValueError(f"No teacher-anchor samples left after max_route_index={max_route_index}")
    return flatten_observation(keyed, obs_dim), actions


class RouteTeacherAnchor:
    def __init__(self, config: TeacherAnchorConfig) -> None:
        if not config.dataset_path:
            raise ValueError("TeacherAnchorConfig.dataset_path is required when enabled")
        self.config = config
        self._rng = np.random.default_rng(0)
        self._rollout_count = 0
        self._obs: torch.Tensor | None = None
        self._actions: torch.Tensor | None = None
        self.actor_extra_steps = 0
        self.last_loss = 0.0

    def on_training_start(self, ppo) -> None:
        """Put the protected-prefix part of the recorded dataset on the device, flattened to the policy's input rows."""
        flat_obs, actions = load_anchor_dataset(self.config.dataset_path, int(self.config.max_route_index), ppo.obs_dim)
        self.actor_extra_steps = int(getattr(ppo, "actor_extra_steps", 0))   # a resumed teacher-anchored run carries its count
        self._obs = torch.as_tensor(flat_obs, device=ppo.device)
        self._actions = torch.as_tensor(actions, device=ppo.device)

    def sample_indices(self) -> np.ndarray:
        """the next batch's row indices: ``rng.integers(0, M, size=min(batch_size, M))`` on the callback's default_rng(0) stream"""
        rows = int(self._actions.shape[0])
        return self._rng.integers(0, rows, size=min(int(self.config.batch_size), rows))

    def on_rollout_end(self, ppo) -> None:
        self._rollout_count += 1
        period = max(int(self.config.every_rollouts), 1)
        if self._rollout_count % period:
            return
        for _step in range(max(int(self.config.gradient_steps), 1)):
            pick = torch.as_tensor(self.sample_indices(), device=ppo.device)
            self.last_loss = self.gradient_step(ppo, self._obs.index_select(0, pick), self._actions.index_select(0, pick))

    def gradient_step(self, ppo, obs: torch.Tensor, teacher_actions: torch.Tensor) -> float:
        """One clip_grad_norm_(0.5) + Adam step of the imitation loss on the actor tensors of ``ppo`` (in place)."""
        from .ppo import mlp_forward

        cfg = ppo.cfg
        views = ppo.policy.views
        leaves = {name: views[name].detach().clone().requires_grad_(True) for name in ACTOR_TENSORS}
        P = {**{k: v for k, v in views.items()}, **leaves}
        mean, _ = mlp_forward(P, obs)                     # distribution mode of the diagonal Gaussian = the mean, unclipped
        loss = torch.nn.functional.mse_loss(mean, teacher_actions) * float(self.config.loss_weight)
        grads = torch.autograd.grad(loss, [leaves[n] for n in ACTOR_TENSORS])
        with torch.no_grad():
            norm = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads]))
            scale = torch.clamp(0.5 / (norm + 1e-6), max=1.0)
            step = ppo.adam_t + self.actor_extra_steps + 1   # per-tensor step count of the actor tensors after this step
            b1, b2 = 0.9, 0.999
            bc1, bc2 = 1.0 - b1 ** step, 1.0 - b2 ** step
            off = {}
            o = 0
            for name, shape in ppo.policy.spec:
                off[name] = o
                o += math.prod(shape)
            for name, g in zip(ACTOR_TENSORS, grads):
                n = g.numel()
                sl = slice(off[name], off[name] + n)
                g = (g * scale).reshape(-1)
                m, v = ppo.adam_m[sl], ppo.adam_v[sl]
                m.mul_(b1).add_(g, alpha=1 - b1)
                v.mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = (v.sqrt() / math.sqrt(bc2)).add_(cfg.adam_eps)
                ppo.policy.flat[sl].addcdiv_(m, denom, value=-cfg.learning_rate / bc1)
        self.actor_extra_steps += 1
        ppo.actor_extra_steps = self.actor_extra_steps
        if ppo._mlp is not None:
            ppo._mlp.pack(ppo.policy.flat)
            ppo._mlp.set_actor_extra_steps(self.actor_extra_steps)
        return float(loss.detach().item())

    def summary(self) -> dict[str, Any]:
        import dataclasses

        out = {f.name: getattr(self.config, f.name) for f in dataclasses.fields(self.config)}   # the config as the reference reports it
        out["enabled"] = True
        out["sample_count"] = int(self._actions.shape[0]) if self._actions is not None else 0
        return out
