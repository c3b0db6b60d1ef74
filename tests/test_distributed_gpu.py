"""Data-parallel hip path on ONE GPU: two ranks (gloo backend, both on cuda:0) run the production kernels.

The driver's multi-GPU runs use RCCL; the collectives are backend-agnostic torch.distributed calls, so what is checked here
is everything around them: env sharding by rank, per-epoch global advantage statistics, 1/(global count) loss scaling,
ent_coef / world, the flat gradient all-reduce, and that both ranks hold identical parameters after an update."""
from __future__ import annotations

import math
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, out_dir: str) -> None:
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import load_golden_config
        from rl_brain_trainer_amd import ppo as P
        from rl_brain_trainer_amd.vec_env import ArmKinematicVecEnv

        dev = torch.device("cuda", 0)
        cfg = load_golden_config("workspace_expansion_bigtrain")
        N, T = 128, 16
        env = ArmKinematicVecEnv(cfg, N, seed=806, first_env_id=rank * N)
        env.set_curriculum_stage(5)
        pcfg = P.PPOConfig(n_steps=T, batch_size=1024 * world, n_epochs=1, hidden=256, learning_rate=1e-3, seed=806, clip_range=0.1, ent_coef=3e-4)
        ppo = P.PPO(env, pcfg, dist=P.Dist(), backend="hip")
        assert ppo.dist.enabled and not ppo.use_graphs
        ppo.collect_rollouts()
        total = T * N
        obs = ppo.obs_buf[:T].view(total, ppo.obs_w)
        act, old_logp, adv, ret = ppo.act_buf.view(total, 7), ppo.logp_buf.view(total), ppo.adv_buf.view(total), ppo.ret_buf.view(total)
        # ---- one minibatch by hand: global statistics, scaled gradient, all-reduce
        g = torch.Generator(device=dev).manual_seed(100 + rank)
        perm = torch.randperm(total, device=dev, generator=g)
        local_bs = 1024
        stats = ppo._epoch_adv_stats(adv, perm, total, local_bs)
        idx = perm[:local_bs]
        ppo._mlp.loss_grad(obs, idx, local_bs, act, old_logp, adv, ret, clip_range=0.1, ent_coef=3e-4 / world, vf_coef=0.5, inv_count=1.0 / (local_bs * world),
                           grad_out=ppo.grad, stats_out=None, adv_stats=stats[0])
        ppo.dist.all_reduce_sum(ppo.grad)
        rows = {"obs": obs[idx, :56].cpu(), "act": act[idx].cpu(), "old": old_logp[idx].cpu(), "adv": adv[idx].cpu(), "ret": ret[idx].cpu()}
        gathered = [None] * world
        dist.all_gather_object(gathered, rows)
        if rank == 0:
            cat = {k: torch.cat([gr[k] for gr in gathered]).to(dev) for k in rows}
            flat = ppo.policy.flat.detach().clone().requires_grad_(True)
            Pv, off = {}, 0
            for name, shape in ppo.policy.spec:
                cnt = math.prod(shape)
                Pv[name] = flat[off:off + cnt].view(shape)
                off += cnt
            mean, value = P.mlp_forward(Pv, cat["obs"])
            logp = P.gaussian_log_prob(cat["act"], mean, Pv["log_std"])
            a = (cat["adv"] - cat["adv"].mean()) / (cat["adv"].std() + 1e-8)
            ratio = torch.exp(logp - cat["old"])
            pl = -torch.min(a * ratio, a * torch.clamp(ratio, 0.9, 1.1)).mean()
            vl = torch.nn.functional.mse_loss(cat["ret"], value)
            ent = (0.5 + 0.5 * math.log(2 * math.pi) + Pv["log_std"]).sum()
            (ref,) = torch.autograd.grad(pl + 0.5 * vl - 3e-4 * ent, flat)
            off, worst = 0, 0.0
            for name, shape in ppo.policy.spec:
                cnt = math.prod(shape)
                scale = ref[off:off + cnt].abs().max().item() + 1e-12
                worst = max(worst, (ppo.grad[off:off + cnt] - ref[off:off + cnt]).abs().max().item() / scale)
                off += cnt
            assert worst <= 5e-4, worst
        # ---- a whole update through the production loop: ranks stay in lock step
        before = ppo.policy.flat.clone()
        ppo.train()
        torch.cuda.synchronize()
        after = ppo.policy.flat.clone()
        assert torch.isfinite(after).all() and (after - before).abs().max() > 0
        both = [None] * world
        dist.all_gather_object(both, after.cpu().numpy())
        assert np.array_equal(both[0], both[1])
        # curriculum bytes are gathered in global env order
        done = torch.full((4,), rank + 1, dtype=torch.uint8, device=dev)
        assert ppo.dist.all_gather_bytes(done).cpu().tolist() == [1] * 4 + [2] * 4
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
        env.close()
    finally:
        dist.destroy_process_group()


def test_two_rank_hip_data_parallel_update(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
