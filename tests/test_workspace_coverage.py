"""Random-start workspace coverage against the reference's outputs (tests/golden/coverage_maps.json,
written by tests/golden/make_golden_coverage.py).

CPU tests inject the oracle's FK (the product path uses the batched device FK and has no CPU fallback); the GPU tests run
the maps with the device FK and the whole evaluator end to end."""
from __future__ import annotations

import json

import numpy as np
import pytest

from conftest import GOLDEN, load_golden_config
from oracle import oracle as orc
from rl_brain_trainer_amd import workspace_coverage as wc


@pytest.fixture(scope="module")
def gold():
    return json.loads((GOLDEN / "coverage_maps.json").read_text())


def _same(a, b, tol=1e-12):
    if isinstance(b, dict):
        assert isinstance(a, dict) and set(a) == set(b), (sorted(a), sorted(b))
        for k in b:
            _same(a[k], b[k], tol)
    elif isinstance(b, (list, tuple)):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            _same(x, y, tol)
    elif isinstance(b, float) and not isinstance(b, bool):
        assert abs(float(a) - b) <= tol * max(1.0, abs(b)), (a, b)
    else:
        assert a == b, (a, b)


def _maps(gold, fk):
    cfg = load_golden_config(gold["config"])
    sz, seed = gold["sizes"], gold["seed"]
    targets, tsum = wc.generate_workspace_target_map(cfg, seed=seed + 1, stage_samples_per_stage=sz["target_stage_samples"],
                                                     random_samples=sz["target_random"], fk=fk)
    starts, ssum = wc.generate_workspace_start_state_map(cfg, seed=seed + 2, stage_samples_per_stage=sz["start_stage_samples"],
                                                         random_samples=sz["start_random"], fk=fk)
    return targets, tsum, starts, ssum


def _check_maps(gold, fk):
    targets, tsum, starts, ssum = _maps(gold, fk)
    _same(targets, gold["targets"])
    _same(tsum, gold["target_summary"])
    _same(starts, gold["starts"])
    _same(ssum, gold["start_summary"])
    pairs, psum = wc.build_pair_sampler_summary(starts=starts, targets=targets, seed=gold["seed"] + 3, pair_count=gold["sizes"]["pair_count"])
    _same(pairs, gold["pairs"])
    _same(psum, gold["pair_summary"])
    rng = np.random.default_rng(gold["seed"])
    for mode in ("known", "frontier", "stress"):
        sel = wc.select_pairs(pairs, mode=mode, limit=gold["sizes"]["limit"], rng=rng)
        assert [p["pair_id"] for p in sel] == gold["selected"][mode]


def test_maps_pairs_and_split_selection_match_reference_with_oracle_fk(gold):
    _check_maps(gold, orc.fk_pose6)


def test_classify_pair_cases(gold):
    for c in gold["classify_cases"]:
        assert wc.classify_pair(start=c["start"], target=c["target"], q_l2=c["q_l2"]) == c["class"], c


def test_summaries_bucket_metrics_and_priorities(gold):
    rows = gold["rows"]
    _same(wc.summarize(rows), gold["summary"])
    bm = wc.bucket_metrics(rows)
    _same(bm, gold["bucket_metrics"])
    _same(wc.update_bucket_priorities(bm), gold["priorities"])
    _same(wc.update_bucket_priorities(gold["bucket_metrics_prev"]), gold["priorities_prev"])
    with pytest.raises(ValueError):
        wc.select_pairs([], mode="nope", limit=1, rng=np.random.default_rng(0))


@pytest.mark.parametrize("heads_draws,tails_draws", [(True, True), (False, False), (True, False), (False, True)])
def test_draw_stream_reproduces_sequential_generator_calls(heads_draws, tails_draws):
    """the block-drawn stream of the map generators == the reference's call sequence on one Generator: a scalar `random()` coin, then a
    7-vector `uniform(-noise, noise)` only when the chosen branch has noise, then scalar-bound `uniform(-s, s, size=7)` draws"""
    seed, count = 1234, 57
    noise = np.linspace(0.01, 0.07, 7)
    rng = np.random.default_rng(seed)
    coins, vecs = [], []
    for _ in range(count):
        heads = rng.random() < 0.65
        coins.append(heads)
        vecs.append(rng.uniform(low=-noise, high=noise) if (heads_draws if heads else tails_draws) else np.zeros(7))
    tail = np.stack([rng.uniform(-0.03, 0.03, size=7) for _ in range(5)])
    stream = wc._DrawStream(seed, block=64)
    heads, u7 = wc._coin_and_noise(stream, count, heads_draws, tails_draws)
    assert heads.tolist() == coins
    used = np.where(heads, heads_draws, tails_draws)
    assert np.array_equal(np.where(used[:, None], wc._scaled(u7, -noise, noise), 0.0), np.stack(vecs))
    assert np.array_equal(wc._scaled(stream.take(5, 7), -0.03, 0.03), tail)


@pytest.mark.gpu
def test_maps_with_device_fk_match_reference(gold):
    _check_maps(gold, None)


@pytest.mark.gpu
def test_coverage_evaluator_end_to_end_matches_serial_oracle(tmp_path, gold):
    """A servo policy through evaluate_full_workspace_coverage: every row of the batched Approach->Finisher run equals the
    reference's serial per-pair loop restated on the CPU oracle (success flags, ready flags, final errors)."""
    import torch

    from rl_brain_trainer_amd import evaluate as ev
    from rl_brain_trainer_amd.vec_env import ArmKinematicVecEnv  # noqa: F401
    from test_eval_checkpoint_gpu import _serial_oracle_episode

    acfg = load_golden_config(gold["config"])
    fcfg = load_golden_config("dock_workspace_handoff_noop_ft_12env_raw")
    gain = 0.9
    holder: dict = {}

    class Servo:
        """model.predict stand-in that reads the vectorised env the evaluator is currently stepping"""
        def __call__(self, obs):
            env = holder["env"]
            dl = torch.tensor(env.config.c.joints.delta_limit[:], device="cuda", dtype=torch.float64)
            scale = env.config.c.env.dock_action_delta_scale or env.config.c.env.action_delta_scale
            info = env.info()
            a = gain * (info["goal_q"].double().t() - info["q"].double().t()) / (dl * scale)
            return a.clamp(-1, 1).to(env.dtype)

    # run_episodes creates the envs internally; capture them through a thin wrapper around reset
    orig_run = ev.run_episodes

    def run(env, policy, opts, **kw):
        holder["env"] = env
        return orig_run(env, policy, opts, **kw)

    ev.run_episodes = run
    try:
        cov = wc.evaluate_full_workspace_coverage(approach_policy=Servo(), approach_cfg=acfg, finisher_policy=Servo(), finisher_cfg=fcfg,
                                                  artifact_root=tmp_path, seed=gold["seed"], episodes_per_split=6, stage_samples_per_stage=3,
                                                  random_target_samples=6, random_start_samples=5, pair_count=64, include_home_stage_eval=False)
    finally:
        ev.run_episodes = orig_run
    for f in ("maps/target_map.jsonl", "maps/start_state_map.jsonl", "maps/start_target_pairs.jsonl", "known_random_start_eval_summary.json",
              "workspace_bucket_metrics.json", "full_workspace_coverage_summary.json", "workspace_failure_report.json"):
        assert (tmp_path / f).exists(), f
    assert cov["pair_sampler_summary"]["pair_count"] == 64
    targets = {json.loads(l)["target_id"]: json.loads(l) for l in (tmp_path / "maps/target_map.jsonl").read_text().splitlines()}
    starts = {json.loads(l)["start_id"]: json.loads(l) for l in (tmp_path / "maps/start_state_map.jsonl").read_text().splitlines()}
    rows = json.loads((tmp_path / "stress_random_start_eval_summary.json").read_text())["episode_rows"]
    r = acfg.c.reward
    for row in rows:
        s, t = starts[row["start_id"]], targets[row["target_id"]]
        opts = {"initial_q": np.array(s["q_start"]), "initial_dq": np.array(s["dq_start"]), "initial_prev_action": np.array(s["prev_action"]),
                "goal_q": np.array(t["q_target"]), "goal_pose6": np.array([*t["ee_target_position"], *t["ee_target_orientation"]]), "policy_mode": "approach"}
        a_res = _serial_oracle_episode(acfg, opts, gain, r, 2)
        # fp32 device env vs fp64 oracle over a whole closed-loop episode
        assert abs(row["approach_final_position_error"] - a_res["pos"]) <= 5e-5, (row["pair_id"], row["approach_final_position_error"], a_res["pos"])
        assert abs(row["approach_final_orientation_error"] - a_res["ori"]) <= 5e-4
        if a_res["max_streak"] >= 2:
            assert row["finisher_ready_dwell"] and row["finisher_ready_hit"]


# --------------------------------------------------------------------------------------------- data parallel (BASELINE configs[3])
def _servo_run(dist_ctx, out_path):
    """the coverage evaluation with a servo policy (approach + finisher), optionally sharded over the ranks of `dist_ctx`"""
    import torch

    from rl_brain_trainer_amd import evaluate as ev

    g = json.loads((GOLDEN / "coverage_maps.json").read_text())
    acfg = load_golden_config(g["config"])
    fcfg = load_golden_config("dock_workspace_handoff_noop_ft_12env_raw")
    holder: dict = {}

    def servo(obs):
        env = holder["env"]
        dl = torch.tensor(env.config.c.joints.delta_limit[:], device="cuda", dtype=torch.float64)
        scale = env.config.c.env.dock_action_delta_scale or env.config.c.env.action_delta_scale
        info = env.info()
        return (0.9 * (info["goal_q"].double().t() - info["q"].double().t()) / (dl * scale)).clamp(-1, 1).to(env.dtype)

    orig_run = ev.run_episodes

    def run(env, policy, opts, **kw):
        holder["env"] = env
        return orig_run(env, policy, opts, **kw)

    ev.run_episodes = run
    try:
        cov = wc.evaluate_full_workspace_coverage(approach_policy=servo, approach_cfg=acfg, finisher_policy=servo, finisher_cfg=fcfg, artifact_root=out_path,
                                                  seed=g["seed"], episodes_per_split=37, stage_samples_per_stage=4, random_target_samples=12,
                                                  random_start_samples=9, pair_count=160, include_home_stage_eval=False, dist=dist_ctx)
    finally:
        ev.run_episodes = orig_run
    return cov


def _coverage_rank(rank: int, world: int, port: int, out_dir: str) -> None:
    import os

    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rl_brain_trainer_amd.ppo import Dist

        cov = _servo_run(Dist(), os.path.join(out_dir, "dp"))
        with open(os.path.join(out_dir, f"cov{rank}.json"), "w") as f:
            json.dump(cov, f)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_coverage_evaluator_sharded_over_two_ranks_matches_single_process(tmp_path):
    """BASELINE configs[3] shards the random-start evaluation over the GPUs of a node: the pair list is cut into rank blocks and ONE
    all-gather of the f64 result columns gives every rank the whole table.  Two gloo ranks on one GPU (37 pairs per split: ragged
    blocks of 19 + 18) must reproduce the single-process run exactly -- every episode row, every summary -- on both ranks, and only
    rank 0 writes the artefact files."""
    import socket

    import torch.multiprocessing as mp

    single = _servo_run(None, tmp_path / "single")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_coverage_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    ranks = [json.loads((tmp_path / f"cov{r}.json").read_text()) for r in range(2)]
    ref = json.loads(json.dumps(single))
    assert ranks[0] == ref and ranks[1] == ref
    for split in ("known", "frontier", "stress"):
        a = json.loads((tmp_path / "single" / f"{split}_random_start_eval_summary.json").read_text())
        b = json.loads((tmp_path / "dp" / f"{split}_random_start_eval_summary.json").read_text())
        assert a == b and len(a["episode_rows"]) > 0


@pytest.mark.gpu
def test_config4_full_size_rank_shards_reproduce_the_unsharded_run():
    """BASELINE configs[3] at its full size: 65536 random-start pairs of workspace_full_coverage_randomstart_overnight (160-step episodes),
    pair list drawn as the reference's sampler draws it.  Size-independent property: the result columns of each of the eight 8192-pair rank
    blocks (global env ids kept) are bit-identical to the corresponding rows of ONE 65536-env run -- what a pair's episode produces does not
    depend on how the evaluation is cut over GPUs.  Deterministic elementwise policy of the observation (row-independent by construction)."""
    import torch

    from rl_brain_trainer_amd import config as kcfg

    cfg = kcfg.to_env_config(kcfg.load_workspace_expansion_config(kcfg.builtin_config_dir() / "workspace_full_coverage_randomstart_overnight.yaml"))
    fk = wc._device_fk(0)
    targets, _ = wc.generate_workspace_target_map(cfg, seed=940002, stage_samples_per_stage=8, random_samples=64, fk=fk)
    starts, _ = wc.generate_workspace_start_state_map(cfg, seed=940003, stage_samples_per_stage=4, random_samples=64, fk=fk)
    n, shard = 65536, 8192
    pairs, summary = wc.build_pair_sampler_summary(starts=starts, targets=targets, seed=940004, pair_count=n)
    assert summary["pair_count"] == n and len(summary["difficulty_class_counts"]) >= 3

    def policy(obs):
        return torch.tanh(3.0 * obs[:, :7] - obs[:, 7:14] + 0.25 * obs[:, 14:21])

    common = dict(starts_by_id={r["start_id"]: r for r in starts}, targets_by_id={r["target_id"]: r for r in targets}, approach_policy=policy,
                  approach_cfg=cfg, finisher_policy=None, finisher_cfg=None, handoff_confirm_steps=2, device=0, obs_stride=56, seed=760001)
    full = wc._run_pairs_columns(pairs=pairs, first_env_id=0, **common)
    assert full.shape == (n, len(wc._COLS)) and torch.isfinite(full).all()
    steps = full[:, wc._COLS.index("approach_steps")]
    assert steps.min() >= 1 and steps.max() <= cfg.c.termination.max_episode_steps + 1
    for r in range(n // shard):
        part = wc._run_pairs_columns(pairs=pairs[r * shard:(r + 1) * shard], first_env_id=r * shard, **common)
        assert torch.equal(part, full[r * shard:(r + 1) * shard]), r
