"""`python bench.py --gpus N` must start N ranks by itself (the reference has nothing distributed: this is wholly the build's job).

No multi-GPU box is available to the builder, so the N-rank path is exercised on ONE GPU: two ranks over gloo, both on cuda:0
(KP1_BENCH_BACKEND / KP1_BENCH_SINGLE_DEVICE are test hooks read by bench.py).  What this covers: the launch itself (parent never touches
the GPU, children get RANK / WORLD_SIZE), env sharding by rank, global advantage statistics, the flat gradient all-reduce, the chunked
done-byte exchange feeding the device tracker, weak-scaling accounting (value = all ranks' env steps / max time) and that both ranks hold
bit-identical parameters after the timed updates.  RCCL itself (and the collectives-inside-hipGraph path) is unmeasured on hardware."""
from __future__ import annotations

import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run_bench(gpus: int, extra_env: dict[str, str]) -> dict:
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.update(extra_env)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1", "--envs", "256", "--n-steps", "32",
           "--batch", "1024", "--epochs", "2", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_gpus2_spawns_two_ranks():
    one = _run_bench(1, {})
    two = _run_bench(2, {"KP1_BENCH_BACKEND": "gloo", "KP1_BENCH_SINGLE_DEVICE": "1"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["scaling"] == "weak" and two["config"]["ranks_in_sync"] is True
    assert two["config"]["backend"] == "gloo" and two["config"]["done_exchange_steps"] == 16
    # gloo cannot be captured: the compute segments are hipGraphs, the collectives run eagerly between them
    assert two["config"]["collectives_in_graph"] is False and two["config"]["graph_mode"] == "segmented" and two["config"]["hip_graphs"] is True
    assert one["config"]["graph_mode"] == "captured" and one["config"]["hip_graphs"] is True
    # weak-scaling bookkeeping, as an identity over fields the line itself prints (no ratio of two wall clocks: two gloo ranks on one card
    # measure the box's host all-reduce latency, not the engine): value = steps * n_steps * envs/GPU * world / (max-over-ranks seconds)
    for rec in (one, two):
        c = rec["config"]
        env_steps = rec["steps"] * c["n_steps"] * c["envs_per_gpu"] * rec["n_gpus"]
        assert rec["value"] > 0 and rec["ms_per_step"] > 0
        assert abs(rec["value"] - env_steps / (rec["ms_per_step"] * rec["steps"] * 1e-3)) <= 1e-6 * rec["value"]
        assert rec["config"]["env_steps_timed"] == env_steps
    assert "finisher" in two and two["finisher"]["value"] > 0
    for rec in (one, two):
        assert rec["roofline"]["frac"] > 0 and rec["roofline"]["optimizer_step_kernels_us"]["mlp_train_tile"] > 0
    assert "config3_env_kernel" in one and one["config3_env_kernel"]["envs"] == 32768 and one["config3_env_kernel"]["frac"] > 0
    assert one["minibatch512"]["optimizer_step_us_eager"] > 0
    assert one["config4_eval_shard"]["pairs"] == 8192 and one["config4_eval_shard"]["env_steps"] >= 8192
    assert [b["envs"] for b in one["config3_env_kernel"]["larger_batches"]] == [131072, 524288]


def test_bench_gpus4_ranks_stay_in_sync():
    """four ranks (gloo, one GPU): rank-count-dependent pieces -- shard offsets, the 1 / (world * n) loss scaling, ent_coef / world, the chunked
    done exchange with four blocks per step -- leave all ranks with bit-identical parameters"""
    four = _run_bench(4, {"KP1_BENCH_BACKEND": "gloo", "KP1_BENCH_SINGLE_DEVICE": "1"})
    assert four["n_gpus"] == 4 and four["config"]["ranks_in_sync"] is True and four["config"]["parallelism"].startswith("dp4")
    assert four["value"] > 0 and four["finisher"]["value"] > 0


def test_chunked_done_exchange_matches_per_step_tracker():
    """kp1_curriculum_observe_chunk on an all-gathered [world, chunk, n_local] block == kp1_curriculum_observe called per env step on the
    rank-major concatenation (what the reference callback sees on one VecEnv of world * n_local envs)."""
    from rl_brain_trainer_amd.curriculum import PointCurriculum

    world, chunk, n_local = 3, 16, 640
    g = torch.Generator().manual_seed(5)
    done = (torch.rand((world, chunk, n_local), generator=g) < 0.02).to(torch.uint8) * 2          # truncated
    succ = (torch.rand((world, chunk, n_local), generator=g) < 0.8).to(torch.uint8) * 4
    block = (done | (succ * (done > 0))).cuda().contiguous()
    kw = dict(success_rate_threshold=0.75, window_episodes=64, min_episodes_per_stage=100, max_stage_index=11, initial_stage_index=2)
    a, b = PointCurriculum(**kw), PointCurriculum(**kw)
    for t in range(chunk):
        a.observe(block[:, t, :].contiguous().view(-1), world * n_local)
    b.observe_chunk(block, n_local, chunk, world)
    sa, sb = a.read(), b.read()
    assert sa.stage_index == sb.stage_index and sa.stage_index > 2
    assert (sa.stage_episode_count, sa.ring_len, sa.ring_head, sa.n_events, sa.num_timesteps) == \
           (sb.stage_episode_count, sb.ring_len, sb.ring_head, sb.n_events, sb.num_timesteps)
    assert list(sa.ring[:sa.ring_len]) == list(sb.ring[:sb.ring_len])
    for k in range(sa.n_events):
        ea, eb = sa.events[k], sb.events[k]
        assert (ea.total_timesteps, ea.from_stage, ea.to_stage, ea.trigger_success_rate) == (eb.total_timesteps, eb.from_stage, eb.to_stage, eb.trigger_success_rate)
    a.close()
    b.close()
