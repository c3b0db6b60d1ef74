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
        assert ppo.dist.enabled and ppo.use_graphs and ppo.graph_mode == "segmented"     # gloo: graph segments, eager collectives
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


def _segmented_worker(rank: int, world: int, port: int, out_dir: str, backend: str = "gloo", mode: str = "segmented", direct: bool = True) -> None:
    """the same two-iteration training run twice in one process: hipGraph segments with eager collectives between them, then every launch
    eager.  Both must leave bit-identical parameters, Adam moments, rollout buffers and curriculum tracker state on every rank.
    backend "nccl" with world 1 (KP1_DIST_FORCE_SINGLE): the same run with the collectives going through RCCL on its own stream."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    if backend == "nccl":
        os.environ["KP1_DIST_FORCE_SINGLE"] = "1"
        os.environ["KP1_DIST_GRAPHS"] = "1" if mode == "captured" else "0"
        os.environ["KP1_RCCL_DIRECT"] = "1" if direct else "0"
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import load_golden_config
        from rl_brain_trainer_amd import ppo as P
        from rl_brain_trainer_amd.curriculum import PointCurriculum
        from rl_brain_trainer_amd.vec_env import ArmKinematicVecEnv

        cfg = load_golden_config("workspace_expansion_bigtrain")
        N, T = 192, 32

        def run(use_graphs: bool):
            env = ArmKinematicVecEnv(cfg, N, seed=806, first_env_id=rank * N)
            cur = PointCurriculum(success_rate_threshold=0.0, window_episodes=8, min_episodes_per_stage=16, max_stage_index=11, initial_stage_index=2)
            pcfg = P.PPOConfig(n_steps=T, batch_size=1024 * world, n_epochs=3, hidden=256, learning_rate=1e-3, seed=806, clip_range=0.1, ent_coef=3e-4)
            ppo = P.PPO(env, pcfg, curriculum=cur, dist=P.Dist(), backend="hip", use_graphs=use_graphs)
            for _ in range(3):          # iteration 1 captures (rollout at once, the epoch after one eager epoch), 2 and 3 replay
                ppo.collect_rollouts()
                ppo.train()
            torch.cuda.synchronize()
            st = cur.read()
            assert ppo.dist.collectives == ("rccl on the launch stream" if backend == "nccl" and direct else "torch.distributed")
            out = (ppo.policy.flat.clone(), ppo.adam_m.clone(), ppo.adam_v.clone(), ppo.adv_buf.clone(), ppo.obs_buf.clone(),
                   (st.stage_index, st.stage_episode_count, st.ring_len, st.ring_head, st.n_events, st.num_timesteps), ppo.graph_mode,
                   None if not use_graphs else (getattr(ppo._rollout_graph, "n_graphs", 1), getattr(ppo._epoch_graph, "n_graphs", 1)))
            cur.close()
            env.close()
            return out

        seg, eager = run(True), run(False)
        assert seg[6] == mode and eager[6] == "none", seg[6]
        # rollout: one segment per done-exchange chunk + the tail; epoch: advantage sums | stats + first tile ... | last Adam
        if mode == "segmented":
            assert seg[7][0] == T // 16 + 1 and seg[7][1] == (T * N) // 1024 + 2, seg[7]
        for a, b in zip(seg[:5], eager[:5]):
            assert torch.equal(a, b)
        assert seg[5] == eager[5] and seg[5][0] > 2 and seg[5][5] == 3 * T * N * world       # the tracker promoted, on global timesteps
        both = [None] * world
        dist.all_gather_object(both, seg[0].cpu().numpy())
        assert all(np.array_equal(both[0], b) for b in both)
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("direct", [True, False])
def test_one_rank_rccl_segmented_graphs_match_eager(tmp_path, direct):
    """the data-parallel path over the REAL backend, as far as one GPU can show it: a one-rank RCCL group that takes the multi-rank code path.
    direct: the all-reduces and byte all-gathers are enqueued on the launch stream itself (rccl.RcclComm, the default with nccl), between the
    graph segments; otherwise through torch.distributed, on its NCCL stream."""
    mp.spawn(_segmented_worker, args=(1, _free_port(), str(tmp_path), "nccl", "segmented", direct), nprocs=1, join=True)
    assert (tmp_path / "ok0").exists()


@pytest.mark.parametrize("direct", [True, False])
def test_one_rank_rccl_captured_graphs_match_eager(tmp_path, direct):
    """the opt-in mode (KP1_DIST_GRAPHS=1): RCCL collectives captured INSIDE the rollout and epoch graphs, one rank"""
    mp.spawn(_segmented_worker, args=(1, _free_port(), str(tmp_path), "nccl", "captured", direct), nprocs=1, join=True)
    assert (tmp_path / "ok0").exists()


def test_two_rank_segmented_graphs_match_eager(tmp_path):
    world = 2
    mp.spawn(_segmented_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_two_rank_hip_data_parallel_update(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
