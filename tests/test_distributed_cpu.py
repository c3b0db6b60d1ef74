"""world_size-2 gloo tests (CPU): the data-parallel path of the PPO update and the env sharding rules.

What N > 1 changes (bench.py / train.py under torch.distributed.run):
  * envs are sharded by rank: rank r owns global env ids [r*N_local, (r+1)*N_local) and seeds default_rng(seed + id);
  * each rank computes the loss gradient on its shard of the minibatch with 1/(global count) scaling and GLOBAL advantage
    statistics, then ONE flat all-reduce(SUM) gives the single-process gradient;
  * done bytes are all-gathered in rank order so the curriculum tracker sees envs in global id order.
"""
from __future__ import annotations

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden_config
from oracle import oracle as orc
from rl_brain_trainer_amd import ppo as P


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _batch(n=256, seed=0, hidden=64):
    g = torch.Generator().manual_seed(seed)
    pol = P.ActorCritic(hidden, torch.device("cpu"), seed=3)
    obs = torch.rand((n, 56), generator=g) * 2 - 1
    with torch.no_grad():
        mean, value = P.mlp_forward(pol.views, obs)
    act = mean + torch.randn((n, 7), generator=g)
    old = P.gaussian_log_prob(act, mean + 0.05 * torch.randn((n, 7), generator=g), pol.views["log_std"])
    adv = torch.randn(n, generator=g) * 2 + 0.3
    ret = value + torch.randn(n, generator=g)
    return pol, obs, act, old, adv, ret


def _worker(rank: int, world: int, port: int, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = P.Dist()
        assert d.enabled and d.world_size == world and d.rank == rank
        pol, obs, act, old, adv, ret = _batch()
        n = obs.shape[0] // world
        sl = slice(rank * n, (rank + 1) * n)
        mean, inv_std = P.global_advantage_stats(adv[sl], d)
        a = (adv[sl] - mean) * inv_std
        grad = P.ppo_loss_and_grad_torch(pol.flat, pol.spec, obs[sl], act[sl], old[sl], a, ret[sl], clip_range=0.1, ent_coef=3e-4, vf_coef=0.5,
                                         world_size=world)
        d.all_reduce_sum(grad)
        done = torch.full((4,), rank + 1, dtype=torch.uint8)
        gathered = d.all_gather_bytes(done)
        flat = pol.flat.clone()
        d.broadcast(flat, src=0)
        if rank == 0:
            out["grad"] = grad.clone()
            out["mean"] = float(mean)
            out["inv_std"] = float(inv_std)
            out["gathered"] = gathered.tolist()
    finally:
        dist.destroy_process_group()


def test_data_parallel_gradient_matches_single_process():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    pol, obs, act, old, adv, ret = _batch()
    a = (adv - adv.mean()) / (adv.std() + 1e-8)
    ref = P.ppo_loss_and_grad_torch(pol.flat, pol.spec, obs, act, old, a, ret, clip_range=0.1, ent_coef=3e-4, vf_coef=0.5, world_size=1)
    assert abs(out["mean"] - float(adv.mean())) < 1e-6
    assert abs(out["inv_std"] - float(1.0 / (adv.std() + 1e-8))) < 1e-5
    err = (out["grad"] - ref).abs().max().item()
    assert err <= 1e-6 * (ref.abs().max().item() + 1e-9) + 1e-9, err
    assert out["gathered"] == [1, 1, 1, 1, 2, 2, 2, 2]  # rank-major = global env order


def test_env_sharding_is_invariant_to_gpu_count():
    """rank r with first_env_id = r * N_local reproduces envs [r*N_local, ...) of the single-process run (oracle, CPU)."""
    cfg = load_golden_config("workspace_expansion_bigtrain")
    full = orc.OracleVecEnv(cfg, 8, seed0=806, first_env_id=0, stage=5)
    o_full = full.reset().copy()
    for r in range(2):
        shard = orc.OracleVecEnv(cfg, 4, seed0=806, first_env_id=4 * r, stage=5)
        assert np.array_equal(shard.reset(), o_full[4 * r:4 * r + 4])
    rng = np.random.default_rng(0)
    a = rng.uniform(-1, 1, size=(8, 7))
    shards = [orc.OracleVecEnv(cfg, 4, seed0=806, first_env_id=4 * r, stage=5) for r in range(2)]
    for s in shards:
        s.reset()
    for _ in range(100):
        of, rf, df = full.step(a)
        for r, s in enumerate(shards):
            o, rw, d = s.step(a[4 * r:4 * r + 4])
            assert np.array_equal(o, of[4 * r:4 * r + 4]) and np.array_equal(rw, rf[4 * r:4 * r + 4]) and np.array_equal(d, df[4 * r:4 * r + 4])


def test_gated_score_golden():
    import json
    from conftest import GOLDEN
    from rl_brain_trainer_amd import evaluate as ev

    for case in json.loads((GOLDEN / "gated_score.json").read_text())["cases"]:
        table = {int(k): v for k, v in case["stage_metrics"].items()}
        got = ev.gated_score(table, case["score_stage_index"], ev.gate_config_from_dict(case["gate"]))
        assert got.keys() == case["selection"].keys()
        for k, v in case["selection"].items():
            if isinstance(v, float):
                assert abs(got[k] - v) <= 1e-12, k
            else:
                assert got[k] == v, k
