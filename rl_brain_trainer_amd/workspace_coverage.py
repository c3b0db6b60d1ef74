"""Random-start full-workspace coverage evaluation on the device engine.

Mirror of the reference's

  workspace/workspace_target_map.py:76-157        generate_workspace_target_map
  workspace/workspace_start_state_map.py:62-134   generate_workspace_start_state_map
  workspace/start_target_pair_sampler.py:31-115   classify_pair, build_pair_sampler_summary
  workspace/adaptive_frontier_sampler.py:22-80    classify_bucket, update_bucket_priorities
  eval/eval_full_workspace_coverage.py:58-308     _select_pairs, _run_pairs, _summarize, _bucket_metrics,
                                                  evaluate_full_workspace_coverage

with the same function names, argument meaning, seeds (seed+1 targets, seed+2 starts, seed+3 pairs, seed for the split
selection, seed+4 home-stage eval) and row / summary keys.  The maps are host logic (a few thousand numpy Generator
draws in the reference's order); forward kinematics of the sampled joint vectors runs batched on the GPU in fp64
(``vec_env.fk_pose6``; tests inject the oracle's FK to check the maps without a GPU), and all episodes of a split run
in lock step in ONE vectorised env (``evaluate.run_episodes``) instead of one ``ArmKinematicEnv`` per pair.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Any, Callable, Sequence

import numpy as np

from . import config as kcfg

FkFn = Callable[[np.ndarray], np.ndarray]  # q[n,7] float64 -> pose6[n,6] float64


def _device_fk(device: int = 0) -> FkFn:
    import torch

    from .vec_env import fk_pose6

    def fk(q: np.ndarray) -> np.ndarray:
        if q.shape[0] == 0:
            return np.zeros((0, 6))
        return fk_pose6(torch.tensor(np.ascontiguousarray(q), dtype=torch.float64, device=torch.device("cuda", device))).cpu().numpy()

    return fk


# --------------------------------------------------------------------------------------------- joint helpers
def _limits(env_cfg: kcfg.EnvConfig) -> tuple[np.ndarray, np.ndarray]:
    c = env_cfg.c
    return np.array(c.joints.lower[:], dtype=float), np.array(c.joints.upper[:], dtype=float)


def joint_limit_margin(q: np.ndarray, lo: np.ndarray, hi: np.ndarray) -> np.ndarray:
    """kinematics/joint_limits.py:166-174"""
    span = np.maximum(hi - lo, 1e-9)
    q = np.asarray(q, dtype=float)
    return np.clip(2.0 * np.minimum((q - lo) / span, (hi - q) / span), 0.0, 1.0)


def sample_joint_configuration(rng: np.random.Generator, lo: np.ndarray, hi: np.ndarray, margin_fraction: float = 0.1) -> np.ndarray:
    """kinematics/joint_limits.py:138-150: one vector draw in [lo + m, hi - m], m = max(span * frac, 1e-6)"""
    margin = np.maximum((hi - lo) * margin_fraction, 1e-6)
    return rng.uniform(low=lo + margin, high=hi - margin, size=(lo.shape[0],)).astype(float)


def sample_stage_joint_target(rng: np.random.Generator, base_q: Sequence[float], noise_q: Sequence[float], lo: np.ndarray, hi: np.ndarray) -> np.ndarray:
    """envs/curriculum.py:90-101: the uniform draw happens only if some noise component is positive"""
    base = np.asarray(base_q, dtype=float)
    noise = np.asarray(noise_q, dtype=float)
    if np.any(noise > 0.0):
        base = base + rng.uniform(low=-noise, high=noise)
    return np.clip(base, lo, hi)


def _stage_arrays(env_cfg: kcfg.EnvConfig, stage_id: int) -> tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    st = env_cfg.c.stages[stage_id]
    return (np.array(st.start_q[:], dtype=float), np.array(st.start_noise[:], dtype=float), np.array(st.goal_q[:], dtype=float),
            np.array(st.goal_noise[:], dtype=float))


def _selected_stages(env_cfg: kcfg.EnvConfig, stage_indices: Sequence[int] | None) -> list[int]:
    n = env_cfg.n_stages
    sel = list(stage_indices) if stage_indices is not None else list(range(n))
    return [int(np.clip(i, 0, n - 1)) for i in sel]


# --------------------------------------------------------------------------------------------- the two maps, column-wise
# Both maps are a few thousand uniform draws from ONE numpy Generator followed by per-sample arithmetic.  The reference makes the draws one
# sample at a time; what has to be reproduced is only the ORDER in which doubles leave the generator (the golden maps of
# tests/golden/coverage_maps.json pin it).  `Generator.random` and `Generator.uniform` both consume exactly one next_double per output element,
# in C order, and uniform is `low + (high - low) * u` with the product rounded before the add -- so the whole draw sequence of a map is a
# slicing of one stream of doubles, taken here in blocks (`_DrawStream`), and everything after it is arithmetic on columns.
class _DrawStream:
    """next_double stream of `np.random.default_rng(seed)`, drawn in blocks; take / peek / skip address it as one flat sequence"""

    def __init__(self, seed: int, block: int = 8192):
        self._rng = np.random.default_rng(seed)
        self._buf = np.empty(0, dtype=float)
        self._block = block

    def peek(self, count: int) -> np.ndarray:
        if count > self._buf.shape[0]:
            self._buf = np.concatenate([self._buf, self._rng.random(max(count - self._buf.shape[0], self._block))])
        return self._buf[:count]

    def skip(self, count: int) -> None:
        self.peek(count)
        self._buf = self._buf[count:]

    def take(self, *shape: int) -> np.ndarray:
        count = int(np.prod(shape, dtype=np.int64))
        out = self.peek(count).copy().reshape(shape)
        self.skip(count)
        return out


def _scaled(u: np.ndarray, low, high) -> np.ndarray:
    """numpy's uniform on already-drawn doubles: low + (high - low) * u, product first (distributions.c random_uniform)"""
    low, high = np.asarray(low, dtype=float), np.asarray(high, dtype=float)
    return low + (high - low) * u


def _stage_targets(stream: _DrawStream, base_q: np.ndarray, noise_q: np.ndarray, count: int, lo: np.ndarray, hi: np.ndarray) -> np.ndarray:
    """`count` consecutive sample_stage_joint_target draws (envs/curriculum.py:90-101) as one [count, 7] block; a stage without noise draws nothing"""
    rows = np.broadcast_to(base_q, (count, kcfg.NJ))
    if np.any(noise_q > 0.0):
        rows = base_q + _scaled(stream.take(count, kcfg.NJ), -noise_q, noise_q)
    return np.clip(rows, lo, hi)


def _valid_q_block(stream: _DrawStream, count: int, lo: np.ndarray, hi: np.ndarray, margin_fraction: float) -> np.ndarray:
    """`count` consecutive sample_joint_configuration draws (kinematics/joint_limits.py:138-150)"""
    margin = np.maximum((hi - lo) * margin_fraction, 1e-6)
    return _scaled(stream.take(count, kcfg.NJ), lo + margin, hi - margin)


def _row_norm(m: np.ndarray) -> np.ndarray:
    return np.sqrt(np.einsum("ij,ij->i", m, m))


def _bin(values: np.ndarray, bins: int) -> np.ndarray:
    """floor(values * bins) clamped to [0, bins - 1], as integers"""
    return np.clip(np.floor(values * bins), 0, bins - 1).astype(int)


def _span_stats(pos: np.ndarray) -> tuple[list[float], list[float], list[float]]:
    if not len(pos):
        zero = [0.0, 0.0, 0.0]
        return zero, zero, zero
    mn, mx = pos.min(axis=0), pos.max(axis=0)
    return (mx - mn).tolist(), mn.tolist(), mx.tolist()


def generate_workspace_target_map(env_cfg: kcfg.EnvConfig, *, seed: int, stage_samples_per_stage: int, random_samples: int,
                                  stage_indices: Sequence[int] | None = None, xyz_bins: int = 8, ori_bins: int = 6, q_l2_bins: int = 6,
                                  fk: FkFn | None = None) -> tuple[list[dict[str, Any]], dict[str, Any]]:
    """workspace/workspace_target_map.py:76-157.  Draw order: for every selected stage its `stage_samples_per_stage` goal draws, then the
    random valid-q draws (margin 0.08)."""
    fk = fk or _device_fk()
    lo, hi = _limits(env_cfg)
    stream = _DrawStream(seed)
    stages = _selected_stages(env_cfg, stage_indices)
    per_stage, n_random = max(stage_samples_per_stage, 0), max(random_samples, 0)
    blocks = [_stage_targets(stream, *_stage_arrays(env_cfg, sid)[2:], per_stage, lo, hi) for sid in stages]
    blocks.append(_valid_q_block(stream, n_random, lo, hi, 0.08))
    q = np.concatenate(blocks, axis=0) if blocks else np.zeros((0, kcfg.NJ))
    n = q.shape[0]
    stage_col: list[int | None] = [sid for sid in stages for _ in range(per_stage)] + [None] * n_random
    kind_col = ["stage_distribution"] * (n - n_random) + ["random_valid_q"] * n_random

    pose = fk(q)
    pos, rpy = pose[:, :3], pose[:, 3:]
    if n:
        box_lo, box_hi = pos.min(axis=0) - 1e-6, pos.max(axis=0) + 1e-6
    else:
        box_lo, box_hi = np.array([-1.0, -1.0, 0.0]), np.array([1.0, 1.0, 2.0])
    margin = joint_limit_margin(q, lo, hi).min(axis=1) if n else np.zeros(0)
    q_l2, ori_l2 = _row_norm(q), _row_norm(rpy)
    cell = _bin((pos - box_lo) / np.maximum(box_hi - box_lo, 1e-9), xyz_bins)
    ori_cell, q_cell = _bin(ori_l2 / np.pi, ori_bins), _bin(q_l2 / 4.5, q_l2_bins)
    # difficulty: 0.45 joint distance from home + 0.35 orientation magnitude + 0.20 closeness to a joint limit, each in [0, 1]
    difficulty = 0.45 * np.minimum(q_l2 / 4.5, 1.0) + 0.35 * np.minimum(ori_l2 / np.pi, 1.0) + 0.20 * (1.0 - np.clip(margin, 0.0, 1.0))
    labels = [f"x{cx}_y{cy}_z{cz}_o{co}_q{cq}" for (cx, cy, cz), co, cq in zip(cell.tolist(), ori_cell.tolist(), q_cell.tolist())]

    q_l, pos_l, rpy_l, cell_l = q.tolist(), pos.tolist(), rpy.tolist(), cell.tolist()
    samples = [{
        "target_id": f"target_{k:06d}", "q_target": q_l[k], "ee_target_position": pos_l[k], "ee_target_orientation": rpy_l[k],
        "stage_id": stage_col[k], "source_type": kind_col[k], "bucket_id": labels[k], "xyz_bucket": cell_l[k],
        "orientation_bucket": int(ori_cell[k]), "joint_l2_bucket": int(q_cell[k]), "joint_limit_margin_min": float(margin[k]),
        "reachability_flag": bool(margin[k] > 0.0), "difficulty_score": float(difficulty[k]),
        "previous_eval_success_rate": None, "previous_failure_reason_counts": None,
    } for k in range(n)]
    span, box_min, box_max = _span_stats(pos)
    summary = {
        "seed": int(seed), "total_target_count": n, "valid_target_count": int(np.count_nonzero(margin > 0.0)), "rejected_target_count": 0,
        "stage_indices": stages, "xyz_span": span, "xyz_min": box_min, "xyz_max": box_max,
        "q_l2_range": [float(q_l2.min()), float(q_l2.max())] if n else [0.0, 0.0],
        "joint_limit_margin_min": float(margin.min()) if n else 0.0, "joint_limit_margin_mean": float(margin.mean()) if n else 0.0,
        "bucket_count": len(set(labels)),
        "stage_is_workspace_note": "Stage IDs are difficulty shells, not the full continuous workspace.",
    }
    return samples, summary


def _coin_and_noise(stream: _DrawStream, count: int, heads_draws: bool, tails_draws: bool, p_heads: float = 0.65) -> tuple[np.ndarray, np.ndarray]:
    """`count` consecutive [one coin double, then a 7-vector of doubles if the branch the coin selects has noise].  Returns (heads[count],
    doubles[count, 7]; rows of a branch without noise are unused).  With both branches alike the records have a fixed length and the block is
    one reshape; otherwise a record's position depends on the coins before it and the offsets are walked over the already-drawn doubles."""
    nj = kcfg.NJ
    if heads_draws == tails_draws:
        rec = stream.take(count, 1 + nj * int(heads_draws))
        return rec[:, 0] < p_heads, (rec[:, 1:] if heads_draws else np.zeros((count, nj)))
    flat = stream.peek(count * (1 + nj))
    starts = np.empty(count, dtype=np.int64)
    at = 0
    for k in range(count):
        starts[k] = at
        at += 1 + nj * int(heads_draws if flat[at] < p_heads else tails_draws)
    heads = flat[starts] < p_heads
    has = np.where(heads, heads_draws, tails_draws)
    doubles = flat[np.minimum(starts[:, None] + 1 + np.arange(nj), flat.shape[0] - 1)] * has[:, None]
    stream.skip(at)
    return heads, doubles


def generate_workspace_start_state_map(env_cfg: kcfg.EnvConfig, *, seed: int, stage_samples_per_stage: int, random_samples: int,
                                       stage_indices: Sequence[int] | None = None, dq_noise: float = 0.001, prev_action_noise: float = 0.03,
                                       fk: FkFn | None = None) -> tuple[list[dict[str, Any]], dict[str, Any]]:
    """workspace/workspace_start_state_map.py:62-134.  Row 0 is the home pose.  Draw order: per selected stage and sample a coin (< 0.65: the
    stage's goal distribution, a "successful_rollout"; else its start distribution, "near_target" from stage 6 on) followed by that
    distribution's draw; then the random valid-q draws (margin 0.10); then, for EVERY row in row order (home included, whose values are then
    discarded), a dq vector and a prev_action vector."""
    fk = fk or _device_fk()
    lo, hi = _limits(env_cfg)
    nj = kcfg.NJ
    stream = _DrawStream(seed)
    stages = _selected_stages(env_cfg, stage_indices)
    per_stage, n_random = max(stage_samples_per_stage, 0), max(random_samples, 0)
    q_blocks, kind_col, stage_col, run_col = [np.zeros((1, nj))], ["home"], [0], [None]
    for sid in stages:
        start_q, start_noise, goal_q, goal_noise = _stage_arrays(env_cfg, sid)
        goal_draws, start_draws = bool(np.any(goal_noise > 0.0)), bool(np.any(start_noise > 0.0))
        heads, u7 = _coin_and_noise(stream, per_stage, goal_draws, start_draws)
        from_goal = goal_q + (_scaled(u7, -goal_noise, goal_noise) if goal_draws else 0.0)
        from_start = start_q + (_scaled(u7, -start_noise, start_noise) if start_draws else 0.0)
        q_blocks.append(np.clip(np.where(heads[:, None], from_goal, from_start), lo, hi))
        tails_kind = "near_target" if sid >= 6 else "successful_rollout"
        kind_col += ["successful_rollout" if h else tails_kind for h in heads.tolist()]
        stage_col += [sid] * per_stage
        run_col += [f"stage{sid:02d}_synthetic_{k:04d}" for k in range(per_stage)]
    q_blocks.append(_valid_q_block(stream, n_random, lo, hi, 0.10))
    kind_col += ["random_valid_q"] * n_random
    stage_col += [None] * n_random
    run_col += [f"random_{k:04d}" for k in range(n_random)]
    q = np.concatenate(q_blocks, axis=0)
    n = q.shape[0]

    motion = stream.take(n, 2 * nj)
    dq, prev_action = _scaled(motion[:, :nj], -dq_noise, dq_noise), _scaled(motion[:, nj:], -prev_action_noise, prev_action_noise)
    dq[0], prev_action[0] = 0.0, 0.0                                  # the home row is at rest

    pose = fk(q)
    pos, rpy = pose[:, :3], pose[:, 3:]
    margin = joint_limit_margin(q, lo, hi).min(axis=1)
    # stability: 0.7 joint-limit margin + 0.3 stillness (|dq| + |prev_action| saturating at 1)
    stability = 0.7 * np.clip(margin, 0.0, 1.0) + 0.3 * (1.0 - np.minimum(_row_norm(dq) + _row_norm(prev_action), 1.0))
    cells = np.stack([_bin((pos[:, 0] + 1.0) / 2.0, 8), _bin((pos[:, 1] + 1.0) / 2.0, 8), _bin(pos[:, 2] / 2.0, 6), _bin(_row_norm(q) / 4.5, 6),
                      _bin(margin, 5)], axis=1).tolist()
    labels = [f"x{c[0]}_y{c[1]}_z{c[2]}_q{c[3]}_m{c[4]}" for c in cells]

    q_l, dq_l, pa_l, pos_l, rpy_l = q.tolist(), dq.tolist(), prev_action.tolist(), pos.tolist(), rpy.tolist()
    samples = [{
        "start_id": f"start_{k:06d}", "q_start": q_l[k], "dq_start": dq_l[k], "prev_action": pa_l[k], "ee_position": pos_l[k],
        "ee_orientation": rpy_l[k], "source_type": kind_col[k], "source_stage": stage_col[k], "source_rollout_id": run_col[k],
        "stability_score": float(stability[k]), "joint_limit_margin_min": float(margin[k]), "bucket_id": labels[k],
    } for k in range(n)]
    kinds, kind_counts = np.unique(np.array(kind_col), return_counts=True)
    first_seen = sorted(range(len(kinds)), key=lambda j: kind_col.index(str(kinds[j])))      # dict order of the reference: first occurrence
    summary = {
        "seed": int(seed), "total_start_count": n, "source_counts": {str(kinds[j]): int(kind_counts[j]) for j in first_seen},
        "bucket_count": len(set(labels)), "xyz_span": _span_stats(pos)[0],
        "joint_limit_margin_min": float(margin.min()), "joint_limit_margin_mean": float(margin.mean()),
        "random_start_note": "Start states intentionally include non-home q states; this is the core distinction from prior home-start stage sweeps.",
    }
    return samples, summary


# --------------------------------------------------------------------------------------------- pairs
# Difficulty classes of a (start, target) pair for the layered pair curriculum (workspace/start_target_pair_sampler.py:13-50).  The rule set
# is evaluated for ALL pairs at once on columns; the first rule that holds gives the class:
#   retention  start is the home pose or the end of a successful rollout, and the target belongs to a stage <= 7
#   local      joint distance <= 0.28
#   frontier / stress   by the target's previous eval success rate when one is known: [0.35, 0.80] / below 0.35
#   medium     joint distance <= 0.70
#   frontier if the target stage is <= 10, else stress
LOCAL_Q_L2, MEDIUM_Q_L2, FRONTIER_SUCCESS_LOW, FRONTIER_SUCCESS_HIGH = 0.28, 0.70, 0.35, 0.80
_RETENTION_SOURCES = ("home", "successful_rollout")


def _classify_columns(*, start_source: np.ndarray, stage_id: np.ndarray, stage_known: np.ndarray, prev_success: np.ndarray, q_l2: np.ndarray) -> np.ndarray:
    """stage_id: int column (0 where unknown, stage_known says which); prev_success: float column, NaN where the target has no eval history"""
    has_rate = ~np.isnan(prev_success)
    rate = np.where(has_rate, prev_success, 1.0)
    rules = [
        (np.isin(start_source, _RETENTION_SOURCES) & stage_known & (stage_id <= 7), "retention"),
        (q_l2 <= LOCAL_Q_L2, "local"),
        (has_rate & (rate >= FRONTIER_SUCCESS_LOW) & (rate <= FRONTIER_SUCCESS_HIGH), "frontier"),
        (has_rate & (rate < FRONTIER_SUCCESS_LOW), "stress"),
        (q_l2 <= MEDIUM_Q_L2, "medium"),
        (stage_id <= 10, "frontier"),
    ]
    return np.select([m for m, _ in rules], [c for _, c in rules], default="stress")


def _stage_columns(stage_values: Sequence[Any]) -> tuple[np.ndarray, np.ndarray]:
    known = np.fromiter((v is not None for v in stage_values), dtype=bool, count=len(stage_values))
    ids = np.fromiter((int(v) if v is not None else 0 for v in stage_values), dtype=np.int64, count=len(stage_values))
    return ids, known


def classify_pair(*, start: dict[str, Any], target: dict[str, Any], q_l2: float) -> str:
    ids, known = _stage_columns([target.get("stage_id")])
    prev = target.get("previous_eval_success_rate")
    return str(_classify_columns(start_source=np.array([start.get("source_type")], dtype=object), stage_id=ids, stage_known=known,
                                 prev_success=np.array([np.nan if prev is None else float(prev)]), q_l2=np.array([float(q_l2)]))[0])


def _table(items: list[dict[str, Any]], key: str, width: int) -> np.ndarray:
    return np.asarray([it[key] for it in items], dtype=np.float64).reshape(len(items), width)


def build_pair_sampler_summary(*, starts: list[dict[str, Any]], targets: list[dict[str, Any]], seed: int, pair_count: int
                               ) -> tuple[list[dict[str, Any]], dict[str, Any]]:
    """start_target_pair_sampler.py:53-115.  The (start, target) picks are two scalar ``rng.integers`` draws per pair in the reference's order
    (that order IS the sampler); distances, margins and difficulty classes are then computed for all pairs at once."""
    if not starts or not targets:
        return [], {"pair_count": 0, "reason": "empty start or target map"}
    rng = np.random.default_rng(seed)
    P = max(int(pair_count), 0)
    picks = np.array([(int(rng.integers(0, len(starts))), int(rng.integers(0, len(targets)))) for _ in range(P)], dtype=np.int64).reshape(P, 2)
    si, ti = picks[:, 0], picks[:, 1]
    start_pos, target_pos = _table(starts, "ee_position", 3)[si], _table(targets, "ee_target_position", 3)[ti]
    q_l2 = np.linalg.norm(_table(targets, "q_target", 7)[ti] - _table(starts, "q_start", 7)[si], axis=1)
    ee_dist = np.linalg.norm(target_pos - start_pos, axis=1)
    ori_dist = np.linalg.norm(_table(targets, "ee_target_orientation", 3)[ti] - _table(starts, "ee_orientation", 3)[si], axis=1)
    z_disp = np.abs(target_pos[:, 2] - start_pos[:, 2])
    stage_all, known_all = _stage_columns([t.get("stage_id") for t in targets])
    prev_all = np.array([np.nan if t.get("previous_eval_success_rate") is None else float(t["previous_eval_success_rate"]) for t in targets])
    source_all = np.array([s_.get("source_type") for s_ in starts], dtype=object)
    classes = _classify_columns(start_source=source_all[si], stage_id=stage_all[ti], stage_known=known_all[ti], prev_success=prev_all[ti], q_l2=q_l2)
    margin_s = np.array([float(s_.get("joint_limit_margin_min", 0.0)) for s_ in starts])[si]
    margin_t = np.array([float(t.get("joint_limit_margin_min", 0.0)) for t in targets])[ti]
    pairs = []
    for k in range(P):
        st, tg = starts[int(si[k])], targets[int(ti[k])]
        pairs.append({"pair_id": f"pair_{k:06d}", "start_id": st["start_id"], "target_id": tg["target_id"], "start_source_type": st.get("source_type"),
                      "target_source_type": tg.get("source_type"), "target_stage_id": tg.get("stage_id"), "start_bucket_id": st.get("bucket_id"),
                      "target_bucket_id": tg.get("bucket_id"), "joint_distance_l2": float(q_l2[k]), "ee_position_distance": float(ee_dist[k]),
                      "orientation_distance": float(ori_dist[k]), "z_displacement": float(z_disp[k]), "start_joint_limit_margin": float(margin_s[k]),
                      "target_joint_limit_margin": float(margin_t[k]), "difficulty_class": str(classes[k])})
    names, first, counts = np.unique(classes, return_index=True, return_counts=True) if P else (np.array([]), np.array([]), np.array([]))
    order = np.argsort(first)
    summary = {
        "seed": int(seed), "pair_count": P, "start_count": len(starts), "target_count": len(targets),
        "difficulty_class_counts": {str(names[j]): int(counts[j]) for j in order},
        "mean_joint_distance_l2": float(q_l2.mean()) if P else 0.0, "mean_ee_position_distance": float(ee_dist.mean()) if P else 0.0,
        "max_joint_distance_l2": float(q_l2.max()) if P else 0.0,
        "pair_curriculum_note": "Pairs are classified for layered curriculum; full-random stress pairs should remain a minority during training.",
    }
    return pairs, summary


_SPLIT_RULES = {   # eval split -> (lowest target stage, highest target stage, admitted difficulty classes); eval_full_workspace_coverage.py:58-72
    "known": (None, 8, ("retention", "local", "medium")),
    "frontier": (8, 11, ("medium", "frontier", "stress")),
}


def select_pairs(pairs: list[dict[str, Any]], *, mode: str, limit: int, rng: np.random.Generator) -> list[dict[str, Any]]:
    """the pairs an eval split runs: a stage / class filter (the whole list for the stress split, or when the filter leaves nothing), cut to
    ``limit`` by ONE ``rng.choice`` without replacement"""
    if mode == "stress":
        keep = np.ones(len(pairs), dtype=bool)
    elif mode in _SPLIT_RULES:
        lo, hi, classes = _SPLIT_RULES[mode]
        stage, _ = _stage_columns([p.get("target_stage_id") for p in pairs])
        cls = np.array([p.get("difficulty_class") for p in pairs], dtype=object)
        keep = (stage <= hi) & np.isin(cls, classes)
        if lo is not None:
            keep &= stage >= lo
    else:
        raise ValueError(f"Unknown pair eval mode: {mode}")
    pool = np.flatnonzero(keep) if keep.any() else np.arange(len(pairs))
    if pool.size > limit:
        pool = pool[rng.choice(pool.size, size=limit, replace=False)]
    return [pairs[int(i)] for i in pool]


# --------------------------------------------------------------------------------------------- bucket priorities
def classify_bucket(*, success_rate: float, mean_min_error: float, mean_final_error: float, previous_success_rate: float | None = None) -> str:
    if previous_success_rate is not None and previous_success_rate >= 0.75 and success_rate < previous_success_rate - 0.20:
        return "forgetting_risk"
    if success_rate >= 0.85:
        return "mastered"
    if 0.35 <= success_rate < 0.85:
        return "frontier"
    if success_rate < 0.20 and mean_min_error > 0.025:
        return "too_hard"
    if mean_min_error <= 0.012 and mean_final_error > mean_min_error + 0.006:
        return "hard_but_promising"
    return "stress"


def priority_for_category(category: str) -> float:
    return {"mastered": 0.15, "frontier": 1.00, "hard_but_promising": 0.95, "forgetting_risk": 1.10, "stress": 0.25, "too_hard": 0.05}.get(category, 0.20)


def update_bucket_priorities(bucket_metrics: dict[str, dict[str, Any]]) -> list[dict[str, Any]]:
    out: list[dict[str, Any]] = []
    for bucket_id, m in bucket_metrics.items():
        success_rate = float(m.get("success_rate", 0.0))
        mean_min_error = float(m.get("mean_min_position_error", m.get("mean_min_error", 999.0)))
        mean_final_error = float(m.get("mean_final_position_error", m.get("mean_final_error", 999.0)))
        previous = m.get("previous_success_rate")
        previous_success_rate = float(previous) if previous is not None else None
        category = classify_bucket(success_rate=success_rate, mean_min_error=mean_min_error, mean_final_error=mean_final_error,
                                   previous_success_rate=previous_success_rate)
        failure_count = int(m.get("failure_count", 0))
        out.append({"bucket_id": bucket_id, "success_rate": success_rate, "mean_min_error": mean_min_error, "mean_final_error": mean_final_error,
                    "previous_success_rate": previous_success_rate, "failure_count": failure_count, "category": category,
                    "sampling_priority": float(priority_for_category(category) * (1.0 + min(failure_count, 20) / 40.0))})
    return sorted(out, key=lambda item: item["sampling_priority"], reverse=True)


# --------------------------------------------------------------------------------------------- summaries
_SUMMARY_MEANS = (("success_rate", "success"), ("ready_rate", "finisher_ready_hit"), ("dwell_success_rate", "finisher_ready_dwell"),
                  ("mean_final_position_error", "final_position_error"), ("mean_final_orientation_error", "final_orientation_error"),
                  ("mean_final_action_magnitude", "final_action_magnitude"), ("mean_final_dq_norm", "final_dq_norm"),
                  ("average_start_target_joint_distance", "joint_distance_l2"), ("average_start_target_ee_distance", "ee_position_distance"))


def _column(rows: list[dict[str, Any]], key: str) -> np.ndarray:
    return np.fromiter((float(r[key]) for r in rows), dtype=np.float64, count=len(rows))


def _groups(labels: list[str]) -> list[tuple[str, np.ndarray]]:
    """(label, row indices) per distinct label, in order of first appearance"""
    arr = np.array(labels, dtype=object)
    names, first, inverse = np.unique(arr, return_index=True, return_inverse=True) if len(labels) else (np.array([]), np.array([]), np.array([]))
    return [(str(names[j]), np.flatnonzero(inverse == j)) for j in np.argsort(first)]


def summarize(rows: list[dict[str, Any]]) -> dict[str, Any]:
    """per-split summary of the pair evaluation (eval_full_workspace_coverage.py:74-107): rates and means over the episode rows, failure
    reasons counted, success broken down by where the start state came from"""
    n = len(rows)
    succ = _column(rows, "success")
    out: dict[str, Any] = {"episode_count": n}
    out.update({name: float(_column(rows, key).mean()) if n else 0.0 for name, key in _SUMMARY_MEANS})
    dist = _column(rows, "joint_distance_l2")
    out["max_successful_joint_l2"] = float(dist[succ > 0].max()) if (succ > 0).any() else 0.0
    out["failure_reason_counts"] = {label: int(idx.size) for label, idx in _groups([r["failure_reason"] for r in rows])}
    out["success_by_start_source"] = {label: {"episode_count": int(idx.size), "success_rate": float(succ[idx].mean())}
                                      for label, idx in _groups([str(r.get("start_source_type", "unknown")) for r in rows])}
    return out


def bucket_metrics(rows: list[dict[str, Any]]) -> dict[str, dict[str, Any]]:
    """success / error statistics per target bucket (what update_bucket_priorities turns into sampling priorities)"""
    succ, final_err, min_err = _column(rows, "success"), _column(rows, "final_position_error"), _column(rows, "min_position_error")
    return {label: {"episode_count": int(idx.size), "success_rate": float(succ[idx].mean()), "failure_count": int((succ[idx] == 0).sum()),
                    "mean_final_position_error": float(final_err[idx].mean()), "mean_min_position_error": float(min_err[idx].mean())}
            for label, idx in _groups([str(r["target_bucket_id"]) for r in rows])}


def failure_reason(approach: dict[str, float], r, success: bool, dwell: bool) -> str:
    """eval_full_workspace_coverage.py:43-55 (r = the approach env's reward config)"""
    if success:
        return "success"
    if approach["final_position_error"] > r.finisher_ready_pos_threshold_m:
        return "position"
    if approach["final_orientation_error"] > r.finisher_ready_ori_threshold_rad:
        return "orientation"
    if approach["final_action_magnitude"] > r.finisher_ready_action_threshold:
        return "motion_action"
    if approach["final_dq_norm"] > r.finisher_ready_dq_threshold:
        return "motion_dq"
    if not dwell:
        return "dwell"
    return "timeout_or_regression"


# --------------------------------------------------------------------------------------------- batched pair runner
# per-pair results as one f64 row: 4 flags + 10 measurements (the columns every summary of this module is computed from)
_COLS = ("success", "ready_hit", "ready_dwell", "coarse_dwell",
         "a_final_position_error", "a_final_orientation_error", "a_final_action_magnitude", "a_final_dq_norm", "min_position_error", "min_orientation_error",
         "final_position_error", "final_orientation_error", "final_action_magnitude", "final_dq_norm",
         "approach_steps")   # env steps the Approach episode ran (throughput accounting; not part of the reference's episode rows)


def _run_pairs_columns(*, pairs, starts_by_id, targets_by_id, approach_policy, approach_cfg, finisher_policy, finisher_cfg, handoff_confirm_steps: int,
                       device: int, obs_stride: int, seed: int, first_env_id: int = 0):
    """one vectorised Approach run over `pairs` (explicit initial state and goal per env), then the handed-off ones continue in a
    vectorised Finisher run -> f64 [len(pairs), len(_COLS)] on the device.  Env k of this call is global env first_env_id + k."""
    import torch

    from . import evaluate as ev
    from .vec_env import ArmKinematicVecEnv

    E = len(pairs)
    dev = torch.device("cuda", device)
    if E == 0:
        return torch.zeros((0, len(_COLS)), dtype=torch.float64, device=dev)
    nj = kcfg.NJ
    # the maps are a few hundred rows, the pair list tens of thousands: one table per map, one index vector per pair list
    s_ids, t_ids = list(starts_by_id), list(targets_by_id)
    s_row, t_row = {k: i for i, k in enumerate(s_ids)}, {k: i for i, k in enumerate(t_ids)}
    S = [starts_by_id[k] for k in s_ids]
    T = [targets_by_id[k] for k in t_ids]
    s_q = np.array([s["q_start"] for s in S], dtype=float).reshape(len(S), nj)
    s_dq = np.array([s.get("dq_start", [0.0] * nj) for s in S], dtype=float).reshape(len(S), nj)
    s_pa = np.array([s.get("prev_action", [0.0] * nj) for s in S], dtype=float).reshape(len(S), nj)
    t_q = np.array([t["q_target"] for t in T], dtype=float).reshape(len(T), nj)
    t_pose = np.array([[*t["ee_target_position"], *t["ee_target_orientation"]] for t in T], dtype=float).reshape(len(T), 6)
    si = np.fromiter((s_row[p["start_id"]] for p in pairs), dtype=np.int64, count=E)
    ti = np.fromiter((t_row[p["target_id"]] for p in pairs), dtype=np.int64, count=E)
    opts = {"initial_q": s_q[si], "initial_dq": s_dq[si], "initial_prev_action": s_pa[si], "goal_q": t_q[ti], "goal_pose6": t_pose[ti], "policy_mode": "approach"}
    r = approach_cfg.c.reward
    env = ArmKinematicVecEnv(approach_cfg, E, device=device, seed=seed, first_env_id=first_env_id)
    if obs_stride != 56:
        env.set_obs_stride(obs_stride)
    a_res, hand = ev.run_episodes(env, approach_policy, opts, ready_cfg=r, handoff_confirm_steps=handoff_confirm_steps)
    env.close()
    final_ready = ev.finisher_ready(a_res["final_position_error"], a_res["final_orientation_error"], a_res["final_action_magnitude"], a_res["final_dq_norm"], r)
    has_hand = final_ready | hand["valid"]
    src = {}
    for k in ("state_q", "state_dq", "state_prev_action", "state_goal_q", "state_goal_pose6"):
        src[k] = torch.where(final_ready[:, None], a_res[k], hand.get(k, torch.zeros_like(a_res[k])))
    finals = ("final_position_error", "final_orientation_error", "final_action_magnitude", "final_dq_norm")
    final = {k: a_res[k].clone() for k in finals}
    success = a_res["success"].clone()
    if finisher_policy is not None and finisher_cfg is not None and bool(has_hand.any()):
        fenv = ArmKinematicVecEnv(finisher_cfg, E, device=device, seed=seed, first_env_id=first_env_id)
        if obs_stride != 56:
            fenv.set_obs_stride(obs_stride)
        safe = {k: torch.where(has_hand[:, None], v, a_res[k]) for k, v in src.items()}
        f_res, _ = ev.run_episodes(fenv, finisher_policy, ev._handoff_options(safe, "dock"), active=has_hand)
        fenv.close()
        for k in final:
            final[k] = torch.where(has_hand, f_res[k], final[k])
        success = torch.where(has_hand, f_res["success"], success)
    coarse_dwell = a_res["max_ready_streak"] >= handoff_confirm_steps
    cols = [success, a_res["ready_hit"] | final_ready, coarse_dwell | final_ready, coarse_dwell,
            *[a_res[k] for k in finals], a_res["min_position_error"], a_res["min_orientation_error"], *[final[k] for k in finals],
            a_res["step_count"]]
    return torch.stack([c.to(torch.float64) for c in cols], dim=1).contiguous()


def _rows_from_columns(pairs: list[dict[str, Any]], cols: np.ndarray, r) -> list[dict[str, Any]]:
    """the per-episode rows of eval_full_workspace_coverage.py:160-199 from the result columns"""
    c = {name: cols[:, k] for k, name in enumerate(_COLS)}
    rows = []
    for idx, pair in enumerate(pairs):
        approach = {k: float(c["a_" + k][idx]) for k in ("final_position_error", "final_orientation_error", "final_action_magnitude", "final_dq_norm")}
        succ = bool(c["success"][idx])
        rows.append({
            "episode_id": idx, "pair_id": pair["pair_id"], "start_id": pair["start_id"], "target_id": pair["target_id"],
            "start_source_type": pair.get("start_source_type"), "target_stage_id": pair.get("target_stage_id"), "target_bucket_id": pair.get("target_bucket_id"),
            "difficulty_class": pair.get("difficulty_class"), "joint_distance_l2": float(pair.get("joint_distance_l2", 0.0)),
            "ee_position_distance": float(pair.get("ee_position_distance", 0.0)), "success": succ,
            "finisher_ready_hit": bool(c["ready_hit"][idx]), "finisher_ready_dwell": bool(c["ready_dwell"][idx]),
            "failure_reason": failure_reason(approach, r, succ, bool(c["coarse_dwell"][idx])),
            "final_position_error": float(c["final_position_error"][idx]), "final_orientation_error": float(c["final_orientation_error"][idx]),
            "approach_final_position_error": approach["final_position_error"], "approach_final_orientation_error": approach["final_orientation_error"],
            "min_position_error": float(c["min_position_error"][idx]), "min_orientation_error": float(c["min_orientation_error"][idx]),
            "final_action_magnitude": float(c["final_action_magnitude"][idx]), "final_dq_norm": float(c["final_dq_norm"][idx]),
        })
    return rows


def run_pairs(*, pairs: list[dict[str, Any]], starts_by_id: dict[str, dict[str, Any]], targets_by_id: dict[str, dict[str, Any]], approach_policy,
              approach_cfg: kcfg.EnvConfig, finisher_policy=None, finisher_cfg: kcfg.EnvConfig | None = None, handoff_confirm_steps: int = 2,
              device: int = 0, obs_stride: int = 56, seed: int = 0, dist=None) -> list[dict[str, Any]]:
    """_run_pairs: every pair is one env of a vectorised Approach run, then the handed-off ones continue in a vectorised Finisher run.

    Data parallel (``dist`` = ppo.Dist of an initialised process group; BASELINE configs[3]: 65536 envs = 8 x 8192): the pair list --
    identical on every rank, it is a function of the seed -- is cut into contiguous rank blocks, each rank runs its block on its own GPU
    (global env ids kept, so a pair's episode does not depend on the GPU count) and ONE all-gather of the f64 result columns gives every
    rank the whole table; rows and summaries are then computed from it as in the single-process run."""
    import torch

    E = len(pairs)
    if E == 0:
        return []
    common = dict(starts_by_id=starts_by_id, targets_by_id=targets_by_id, approach_policy=approach_policy, approach_cfg=approach_cfg,
                  finisher_policy=finisher_policy, finisher_cfg=finisher_cfg, handoff_confirm_steps=handoff_confirm_steps, device=device,
                  obs_stride=obs_stride, seed=seed)
    if dist is not None and dist.enabled:
        world, per = dist.world_size, (E + dist.world_size - 1) // dist.world_size
        lo = min(dist.rank * per, E)
        hi = min(lo + per, E)
        mine = _run_pairs_columns(pairs=pairs[lo:hi], first_env_id=lo, **common)
        block = torch.zeros((per, len(_COLS)), dtype=torch.float64, device=mine.device)
        block[:hi - lo] = mine
        table = torch.empty((world, per, len(_COLS)), dtype=torch.float64, device=mine.device)
        dist.all_gather_into(table, block)
        cols = table.view(world * per, len(_COLS))[:E]      # rank-major blocks = pair order
    else:
        cols = _run_pairs_columns(pairs=pairs, first_env_id=0, **common)
    return _rows_from_columns(pairs, cols.cpu().numpy(), approach_cfg.c.reward)


def _write_json(path: Path, payload: Any) -> None:
    path.parent.mkdir(parents=True, exist_ok=True)
    path.write_text(json.dumps(payload, indent=2))


def _write_jsonl(path: Path, rows: list[dict[str, Any]]) -> None:
    path.parent.mkdir(parents=True, exist_ok=True)
    path.write_text("\n".join(json.dumps(r) for r in rows) + ("\n" if rows else ""))


def evaluate_full_workspace_coverage(*, approach_policy, approach_cfg: kcfg.EnvConfig, artifact_root: str | Path | None = None, finisher_policy=None,
                                     finisher_cfg: kcfg.EnvConfig | None = None, seed: int = 940001, episodes_per_split: int = 96,
                                     stage_samples_per_stage: int = 96, random_target_samples: int = 384, random_start_samples: int = 384,
                                     pair_count: int = 2048, handoff_confirm_steps: int = 2, include_home_stage_eval: bool = True, device: int = 0,
                                     obs_stride: int = 56, dist=None) -> dict[str, Any]:
    """evaluate_full_workspace_coverage with policies passed as callables (checkpoint loading is the caller's); same artefact
    files under ``artifact_root`` (maps/, *_random_start_eval_summary.json, workspace_bucket_metrics.json,
    full_workspace_coverage_summary.json, workspace_failure_report.json, home_start_stage_eval/)."""
    from . import evaluate as ev

    rng = np.random.default_rng(seed)
    # data parallel: maps, pairs and split selection are functions of the seed (identical on every rank); the episodes of each split are
    # sharded over the ranks by run_pairs; rank 0 writes the artefacts
    root = Path(artifact_root) if artifact_root is not None and (dist is None or dist.rank == 0) else None
    fk = _device_fk(device)
    target_samples, target_summary = generate_workspace_target_map(approach_cfg, seed=seed + 1, stage_samples_per_stage=stage_samples_per_stage,
                                                                   random_samples=random_target_samples, fk=fk)
    start_samples, start_summary = generate_workspace_start_state_map(approach_cfg, seed=seed + 2, stage_samples_per_stage=max(stage_samples_per_stage // 2, 1),
                                                                      random_samples=random_start_samples, fk=fk)
    pairs, pair_summary = build_pair_sampler_summary(starts=start_samples, targets=target_samples, seed=seed + 3, pair_count=pair_count)
    if root is not None:
        _write_jsonl(root / "maps" / "target_map.jsonl", target_samples)
        _write_json(root / "maps" / "target_map_summary.json", target_summary)
        _write_jsonl(root / "maps" / "start_state_map.jsonl", start_samples)
        _write_json(root / "maps" / "start_state_map_summary.json", start_summary)
        _write_jsonl(root / "maps" / "start_target_pairs.jsonl", pairs)
        _write_json(root / "maps" / "pair_sampler_summary.json", pair_summary)
    starts_by_id = {row["start_id"]: row for row in start_samples}
    targets_by_id = {row["target_id"]: row for row in target_samples}

    split_rows: dict[str, list[dict[str, Any]]] = {}
    for split in ("known", "frontier", "stress"):
        selected = select_pairs(pairs, mode=split, limit=episodes_per_split, rng=rng)
        rows = run_pairs(pairs=selected, starts_by_id=starts_by_id, targets_by_id=targets_by_id, approach_policy=approach_policy, approach_cfg=approach_cfg,
                         finisher_policy=finisher_policy, finisher_cfg=finisher_cfg, handoff_confirm_steps=handoff_confirm_steps, device=device,
                         obs_stride=obs_stride, seed=seed, dist=dist)
        split_rows[split] = rows
        if root is not None:
            _write_json(root / f"{split}_random_start_eval_summary.json", {"summary": summarize(rows), "episode_rows": rows})

    all_rows = [row for rows in split_rows.values() for row in rows]
    bm = bucket_metrics(all_rows)
    priorities = update_bucket_priorities(bm)
    stable = sum(1 for d in bm.values() if float(d["success_rate"]) >= 0.85)
    partial = sum(1 for d in bm.values() if 0.35 <= float(d["success_rate"]) < 0.85)
    stress = sum(1 for d in bm.values() if float(d["success_rate"]) < 0.35)
    coverage = {
        "target_map_summary": target_summary, "start_state_map_summary": start_summary, "pair_sampler_summary": pair_summary,
        "random_start_known_workspace": summarize(split_rows["known"]), "random_start_frontier": summarize(split_rows["frontier"]),
        "full_reachable_stress": summarize(split_rows["stress"]),
        "covered_bucket_fraction": float((stable + partial) / max(len(bm), 1)), "stable_bucket_fraction": float(stable / max(len(bm), 1)),
        "partial_bucket_fraction": float(partial / max(len(bm), 1)), "stress_bucket_fraction": float(stress / max(len(bm), 1)),
        "covered_bucket_count": int(stable + partial), "total_eval_bucket_count": len(bm), "top_sampling_priorities": priorities[:30],
    }
    if root is not None:
        _write_json(root / "workspace_bucket_metrics.json", bm)
        _write_json(root / "workspace_failure_report.json", {
            split: {"failure_reason_counts": summarize(rows)["failure_reason_counts"],
                    "worst_rows": sorted(rows, key=lambda row: row["final_position_error"] + 0.02 * row["final_orientation_error"], reverse=True)[:20]}
            for split, rows in split_rows.items()})
    if include_home_stage_eval:
        home = ev.evaluate_workspace_expansion(approach_policy=approach_policy, finisher_policy=finisher_policy, approach_cfg=approach_cfg,
                                               finisher_cfg=finisher_cfg, episodes=max(8, min(episodes_per_split // 4, 32)), seed=seed + 4,
                                               stage_indices=list(range(approach_cfg.n_stages)), handoff_confirm_steps=handoff_confirm_steps,
                                               artifact_root=(root / "home_start_stage_eval") if root is not None else None, device=device, obs_stride=obs_stride)
        coverage["home_start_stage_metrics"] = home["stage_metrics"]
    if root is not None:
        _write_json(root / "full_workspace_coverage_summary.json", coverage)
    return coverage
