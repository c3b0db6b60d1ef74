"""Route prefix curriculum and the sequential route evaluator on the device engine.

What the reference does in kinematic_phase1/route/route_curriculum.py:17-136 (``RoutePrefixCurriculumCallback``: four sliding windows over
finished episodes, promotion to the next route prefix when all four rates pass) runs here as a one-wave device tracker
(include/kp1_route.h, kp1_route_curriculum_*) so the PPO rollout stays one hipGraph replay; this module holds its host handle.
kinematic_phase1/eval/eval_route_curriculum.py:57-248 (sequential evaluation: the final (q, dq, prev_action) of waypoint k start
waypoint k+1) and eval/eval_route_gate.py:17-99 (accept / reject of a checkpoint from per-prefix evaluations) keep their JSON schemas --
status scripts read them -- with the bookkeeping done on numeric columns.
"""
from __future__ import annotations

import ctypes as C

import json
from dataclasses import dataclass
from pathlib import Path
from typing import Any, Callable, Sequence

import numpy as np
import torch

from . import config as kcfg
from . import route_config as rcfg
from .route_env import RouteVecEnv


@dataclass(frozen=True)
class RouteCurriculumStage:
    name: str
    prefix_end_index: int


def build_prefix_stages(prefixes: Sequence[int]) -> list[RouteCurriculumStage]:
    return [RouteCurriculumStage(name=f"prefix_{int(p)}", prefix_end_index=int(p)) for p in prefixes]


# --------------------------------------------------------------------------------------------- the same callback on the device
MAX_STAGES, MAX_WINDOW, MAX_HISTORY = 16, 1024, 32


class _CurriculumEvent(C.Structure):
    _fields_ = [("total_timesteps", C.c_int64), ("from_stage", C.c_int32), ("to_stage", C.c_int32), ("from_prefix_end_index", C.c_int32),
                ("to_prefix_end_index", C.c_int32), ("recent_success_rate", C.c_double), ("recent_route_ready_hit_rate", C.c_double),
                ("recent_orientation_hit_rate", C.c_double), ("recent_regression_rate", C.c_double)]


class _CurriculumState(C.Structure):
    _fields_ = [("stage_index", C.c_int32), ("stage_episode_count", C.c_int32), ("ring_len", C.c_int32), ("ring_head", C.c_int32),
                ("window_episodes", C.c_int32), ("min_episodes_per_stage", C.c_int32), ("n_stages", C.c_int32), ("n_events", C.c_int32),
                ("prefix_end_index", C.c_int32 * MAX_STAGES), ("ring_sums", C.c_int32 * 4), ("promotion_success_rate", C.c_double), ("promotion_route_ready_hit_rate", C.c_double),
                ("promotion_orientation_hit_rate", C.c_double), ("promotion_max_regression_rate", C.c_double), ("num_timesteps", C.c_int64),
                ("ring", (C.c_uint8 * MAX_WINDOW) * 4), ("events", _CurriculumEvent * MAX_HISTORY)]


class RoutePrefixCurriculumDevice:
    """RoutePrefixCurriculum with the per-step scan on the device (include/kp1_route.h, kp1_route_curriculum_*): same promotion rule and
    history, no host synchronisation per step, so the PPO rollout stays one hipGraph replay.  Plugs into ``PPO(curriculum=...)``."""

    def __init__(self, *, stages: list[RouteCurriculumStage], promotion_success_rate: float, promotion_route_ready_hit_rate: float,
                 promotion_orientation_hit_rate: float, promotion_max_regression_rate: float, window_episodes: int, min_episodes_per_stage: int = 128) -> None:
        if not stages:
            raise ValueError("RoutePrefixCurriculumCallback requires at least one stage")
        if len(stages) > MAX_STAGES or int(window_episodes) > MAX_WINDOW:
            raise ValueError(f"the device tracker holds at most {MAX_STAGES} stages and a window of {MAX_WINDOW} episodes")
        self.stages = list(stages)
        self._args = (float(promotion_success_rate), float(promotion_route_ready_hit_rate), float(promotion_orientation_hit_rate),
                      float(promotion_max_regression_rate), max(int(window_episodes), 1), max(int(min_episodes_per_stage), 1))
        self.env: RouteVecEnv | None = None
        self._st = C.c_void_p()

    @classmethod
    def from_config(cls, cfg: dict[str, Any], n_waypoints: int) -> "RoutePrefixCurriculumDevice":
        """``route.curriculum`` block of the YAML with the trainer's defaults (train_route_curriculum.py:86-88, 150-160)"""
        block = dict((cfg.get("route", {}) or {}).get("curriculum", {}) or {})
        defaults = {"promotion_success_rate": 0.80, "promotion_route_ready_hit_rate": 0.80, "promotion_orientation_hit_rate": 0.90,
                    "promotion_max_regression_rate": 0.20}
        rates = {k: float(block.get(k, v)) for k, v in defaults.items()}
        return cls(stages=build_prefix_stages(rcfg.prefix_stages(cfg, n_waypoints)), window_episodes=int(block.get("promotion_window_episodes", 256)),
                   min_episodes_per_stage=int(block.get("min_episodes_per_stage", 128)), **rates)

    @property
    def window_episodes(self) -> int:
        return int(self._args[4])

    @property
    def min_episodes_per_stage(self) -> int:
        return int(self._args[5])

    @property
    def promotion_max_regression_rate(self) -> float:
        return float(self._args[3])

    def attach(self, env: RouteVecEnv) -> None:
        """_on_training_start: allocate the tracker next to the env and apply the first prefix."""
        from . import native

        self.env = env
        L = env.L
        vp, i32, f64 = C.c_void_p, C.c_int32, C.c_double
        L.kp1_route_curriculum_create.argtypes = [vp, C.POINTER(i32), i32, f64, f64, f64, f64, i32, i32, C.POINTER(vp)]
        L.kp1_route_curriculum_destroy.argtypes = [vp, vp]
        L.kp1_route_curriculum_observe.argtypes = [vp, vp, vp, i32, vp]
        L.kp1_route_curriculum_read.argtypes = [vp, vp, C.POINTER(_CurriculumState), vp]
        prefixes = (i32 * len(self.stages))(*[int(s.prefix_end_index) for s in self.stages])
        with torch.cuda.device(env.device):
            native.check(L.kp1_route_curriculum_create(env._handle, prefixes, len(self.stages), *self._args, C.byref(self._st)))
        env.route_cfg.reset.min_route_index, env.route_cfg.reset.max_route_index = 1, int(self.stages[0].prefix_end_index)

    def observe(self, dones: torch.Tensor, steps_per_call: int) -> None:
        from . import native

        if dones.numel() != self.env.n_envs:
            raise ValueError("the device route curriculum tracks one process's envs (no data-parallel gather of the route flags)")
        stream = torch.cuda.current_stream(self.env.device).cuda_stream
        native.check(self.env.L.kp1_route_curriculum_observe(self.env._handle, self._st, C.c_void_p(dones.data_ptr()), int(steps_per_call), C.c_void_p(stream)))

    def read(self) -> _CurriculumState:
        from . import native

        out = _CurriculumState()
        stream = torch.cuda.current_stream(self.env.device).cuda_stream
        native.check(self.env.L.kp1_route_curriculum_read(self.env._handle, self._st, C.byref(out), C.c_void_p(stream)))
        self.env.route_cfg.reset.min_route_index, self.env.route_cfg.reset.max_route_index = 1, int(out.prefix_end_index[out.stage_index])
        return out

    def summary(self) -> dict[str, object]:
        st = self.read()
        n = int(st.ring_len)

        def mean(q: int) -> float:
            return float(sum(st.ring[q][k] for k in range(n))) / float(n) if n else 0.0

        history = []
        for k in range(min(int(st.n_events), MAX_HISTORY)):
            e = st.events[k]
            history.append({"from_stage": self.stages[e.from_stage].name, "to_stage": self.stages[e.to_stage].name,
                            "from_prefix_end_index": int(e.from_prefix_end_index), "to_prefix_end_index": int(e.to_prefix_end_index),
                            "total_timesteps": int(e.total_timesteps), "recent_success_rate": float(e.recent_success_rate),
                            "recent_route_ready_hit_rate": float(e.recent_route_ready_hit_rate),
                            "recent_orientation_hit_rate": float(e.recent_orientation_hit_rate), "recent_regression_rate": float(e.recent_regression_rate)})
        stage = self.stages[int(st.stage_index)]
        return {"stage_index": int(st.stage_index), "stage_name": stage.name, "prefix_end_index": int(stage.prefix_end_index),
                "stage_episode_count": int(st.stage_episode_count), "recent_success_rate": mean(0), "recent_route_ready_hit_rate": mean(1),
                "recent_orientation_hit_rate": mean(2), "recent_regression_rate": mean(3), "history": history}

    def close(self) -> None:
        if self.env is not None and self._st.value:
            self.env.L.kp1_route_curriculum_destroy(self.env._handle, self._st)
            self._st = C.c_void_p()


# --------------------------------------------------------------------------------------------- sequential evaluator
PolicyFn = Callable[[torch.Tensor], torch.Tensor]


def _roll_one(env: RouteVecEnv, policy: PolicyFn, *, initial_q: np.ndarray, goal_index: int, initial_dq: np.ndarray, initial_prev_action: np.ndarray,
              success_dwell_steps: int) -> dict[str, Any]:
    obs = env.reset(options={"route_index": int(goal_index), "start_route_index": 0, "initial_q": initial_q[None], "initial_dq": initial_dq[None],
                             "initial_prev_action": initial_prev_action[None], "evaluator_state": True})
    info = env.info()
    min_pos = float(info["position_error_norm"][0])
    min_ori = float(info["orientation_error_norm"][0])
    min_q = float(np.linalg.norm(env.route_q[min(max(goal_index, 0), env.n_waypoints - 1)] - env.get_state()["q"][0]))
    first_ready_step = None
    max_ready_streak = 0
    steps = 0
    done = 0
    # the episode is a serial dependency chain (one env, the next waypoint starts where this one ends): per step ONE device -> host copy of
    # the eight scalars the bookkeeping needs (f64, exact) instead of one synchronising read per scalar
    keys = ("position_error_norm", "orientation_error_norm", "route_q_error_norm", "route_ready", "route_ready_streak", "action_l2", "executed_delta_q_l2")
    pos = ori = qerr = act = dqn = 0.0
    while not (done & 3):
        action = policy(obs)
        obs, _, d = env.step(action, auto_reset=False)
        info = env.info()
        row = torch.stack([info[k][0].double() for k in keys] + [d[0].double()]).cpu().tolist()
        pos, ori, qerr, ready, streak, act, dqn = row[:7]
        done = int(row[7])
        steps += 1
        min_pos, min_ori, min_q = min(min_pos, pos), min(min_ori, ori), min(min_q, qerr)
        if ready != 0.0 and first_ready_step is None:
            first_ready_step = steps
        max_ready_streak = max(max_ready_streak, int(streak))
    if steps == 0:   # (cannot happen: a fresh episode is never done) keep the row well defined
        pos, ori, qerr = float(info["position_error_norm"][0]), float(info["orientation_error_norm"][0]), float(info["route_q_error_norm"][0])
        act, dqn = float(info["action_l2"][0]), float(info["executed_delta_q_l2"][0])
    st = env.get_state()
    return {
        "route_index": int(goal_index), "success": bool(done & 4), "route_ready_hit": bool(first_ready_step is not None),
        "route_ready_dwell": bool(max_ready_streak >= success_dwell_steps), "first_ready_step": first_ready_step, "max_ready_streak": int(max_ready_streak),
        "steps": int(steps), "final_position_error": float(pos), "final_orientation_error": float(ori), "final_q_error": float(qerr),
        "min_position_error": float(min_pos), "min_orientation_error": float(min_ori), "min_q_error": float(min_q),
        "final_action_magnitude": float(act), "final_dq_norm": float(dqn),
        "final_q": st["q"][0].copy(), "final_dq": st["dq"][0].copy(), "final_prev_action": st["prev_action"][0].copy(),
    }


# first matching rule names why a waypoint failed (eval_route_curriculum.py:113-124): (label, row key, limit); a value above its limit matches
_FAILURE_RULES = (("position", "final_position_error", 0.010), ("orientation", "final_orientation_error", 0.150),
                  ("motion_action", "final_action_magnitude", 1.20), ("motion_action", "final_dq_norm", 0.040), ("q_error", "final_q_error", 0.500))
_ROUTE_CHUNKS = np.array([1, 41, 81, 121, 181, 261, 361, 484])       # chunk c covers route indices [_ROUTE_CHUNKS[c], _ROUTE_CHUNKS[c + 1])
_ROW_MEANS = ("final_position_error", "final_orientation_error", "final_q_error")


def _failure_reason(row: dict[str, Any]) -> str:
    for label, key, limit in _FAILURE_RULES:
        if float(row.get(key, 0.0)) > limit:
            return label
    return "unknown" if row["route_ready_dwell"] else "dwell_or_motion"


def _row_columns(rows: list[dict[str, Any]]) -> dict[str, np.ndarray]:
    """the per-waypoint rows of a sequential evaluation as columns"""
    flags = {k: np.fromiter((bool(r[k]) for r in rows), dtype=bool, count=len(rows)) for k in ("success", "route_ready_hit", "route_ready_dwell")}
    reals = {k: np.fromiter((float(r[k]) for r in rows), dtype=np.float64, count=len(rows)) for k in _ROW_MEANS}
    return {**flags, **reals, "route_index": np.fromiter((int(r["route_index"]) for r in rows), dtype=np.int64, count=len(rows))}


def summarize_rows(rows: list[dict[str, Any]], route_progress_m: np.ndarray) -> dict[str, Any]:
    """route_eval_sequential_summary.json: rates over all targets, the unbroken success prefix from the start and the route length it covers,
    the first failed waypoint and why, mean / max final errors"""
    if not rows:
        return {"target_count": 0}
    col = _row_columns(rows)
    failed = np.flatnonzero(~col["success"])
    prefix = int(failed[0]) if failed.size else len(rows)
    reached = min(prefix, len(route_progress_m) - 1)
    out: dict[str, Any] = {"target_count": len(rows)}
    out.update({name: float(col[key].mean()) for name, key in (("success_rate", "success"), ("route_ready_hit_rate", "route_ready_hit"),
                                                               ("route_ready_dwell_rate", "route_ready_dwell"))})
    out["longest_success_prefix"] = prefix
    out["cumulative_successful_route_distance_m"] = float(route_progress_m[reached] - route_progress_m[0])
    out["first_failure_index"] = int(col["route_index"][failed[0]]) if failed.size else None
    out["first_failure_reason"] = _failure_reason(rows[int(failed[0])]) if failed.size else None
    out.update({f"mean_{k}": float(col[k].mean()) for k in _ROW_MEANS})
    out.update({f"max_{k}": float(col[k].max()) for k in _ROW_MEANS[:2]})
    return out


def chunk_metrics(rows: list[dict[str, Any]]) -> dict[str, Any]:
    """route_chunk_metrics.json: the same rates per fixed stretch of the route (chunks without a target are left out)"""
    if not rows:
        return {}
    col = _row_columns(rows)
    which = np.digitize(col["route_index"], _ROUTE_CHUNKS) - 1
    out: dict[str, Any] = {}
    for c in range(len(_ROUTE_CHUNKS) - 1):
        sel = which == c
        if not sel.any():
            continue
        entry: dict[str, Any] = {"target_count": int(sel.sum()), "success_rate": float(col["success"][sel].mean()),
                                 "route_ready_hit_rate": float(col["route_ready_hit"][sel].mean())}
        entry.update({f"mean_{k}": float(col[k][sel].mean()) for k in _ROW_MEANS})
        out[f"chunk_{c}_{int(_ROUTE_CHUNKS[c])}_{int(_ROUTE_CHUNKS[c + 1]) - 1}"] = entry
    return out


def evaluate_sequential_route(*, policy: PolicyFn | Callable[[RouteVecEnv], PolicyFn], cfg: dict[str, Any], route_q: np.ndarray,
                              artifact_root: str | Path | None = None, end_index: int | None = None, start_index: int = 1, device: int = 0,
                              real: str = "f32", policy_needs_env: bool = False) -> dict[str, Any]:
    """evaluate_sequential_route with the policy passed as a callable obs[1, obs_dim] -> action[1, 7] (checkpoint loading is the
    caller's).  Always the single-waypoint env (``_make_route_env``), whatever ``route.sequence`` says."""
    W = int(route_q.shape[0])
    final_end = min(int(end_index or (W - 1)), W - 1)
    seq_off = {**cfg, "route": {**(cfg.get("route", {}) or {}), "sequence": {**((cfg.get("route", {}) or {}).get("sequence", {}) or {}), "enabled": False}}}
    base = kcfg.to_env_config(cfg)
    env = RouteVecEnv(base, rcfg.route_config_from_dict(seq_off, max_route_index=end_index or (W - 1)), route_q, 1, device=device, seed=0, real=real)
    fn = policy(env) if policy_needs_env else policy
    rows: list[dict[str, Any]] = []
    cq = np.asarray(route_q[max(start_index - 1, 0)], dtype=float).copy()
    cdq = np.zeros_like(cq)
    cpa = np.zeros_like(cq)
    dwell = int(base.c.termination.success_dwell_steps)
    for idx in range(int(start_index), final_end + 1):
        row = _roll_one(env, fn, initial_q=cq, goal_index=idx, initial_dq=cdq, initial_prev_action=cpa, success_dwell_steps=dwell)
        rows.append({k: v for k, v in row.items() if k not in {"final_q", "final_dq", "final_prev_action"}})
        cq, cdq, cpa = row["final_q"], row["final_dq"], row["final_prev_action"]
    progress = env.route_progress_m.copy()
    env.close()
    summary = summarize_rows(rows, progress)
    summary.update({"schema_version": "v5.route_curriculum.sequential_eval.v1", "mode": "sequential_actual_final_q_to_next_dense_q_goal",
                    "start_index": int(start_index), "end_index": int(final_end)})
    if artifact_root is not None:
        root = Path(artifact_root)
        root.mkdir(parents=True, exist_ok=True)
        (root / "route_eval_sequential_summary.json").write_text(json.dumps(summary, indent=2))
        (root / "route_chunk_metrics.json").write_text(json.dumps(chunk_metrics(rows), indent=2))
        with (root / "route_eval_sequential_steps.jsonl").open("w", encoding="utf-8") as fh:
            for row in rows:
                fh.write(json.dumps(row, sort_keys=True) + "\n")
        failure = next((row for row in rows if not row["success"]), None)
        (root / "route_failure_report.json").write_text(json.dumps({"first_failure_index": summary["first_failure_index"],
                                                                    "first_failure_reason": summary["first_failure_reason"], "first_failure": failure}, indent=2))
    return {**summary, "rows": rows, "chunk_metrics": chunk_metrics(rows), "final_q": np.asarray(cq).tolist()}


# --------------------------------------------------------------------------------------------- sequential gate
def evaluate_route_gate(*, evaluate: Callable[..., dict[str, Any]], artifact_root: str | Path, prefixes: Sequence[int], full_end_index: int | None,
                        min_prefix120_success_rate: float, best_full_longest_prefix: int, full_prefix_tolerance: int, checkpoint: str = "",
                        config: str = "", route_path: str = "") -> dict[str, Any]:
    """Accept or reject a checkpoint from sequential evaluations at several prefixes (eval/eval_route_gate.py:17-99; summary schema
    v5.route_gate.v1).  ``evaluate(artifact_root=, start_index=, end_index=)`` runs one sequential evaluation -- the trainer binds
    evaluate_sequential_route to the policy under test.  A checkpoint passes when it (1) still chains all of prefix 120 at the required
    success rate, (2) gets beyond waypoint 120 on prefix 180, (3) has its first prefix-180 failure after 120, and (4) on the full route
    stays within the tolerance of the best chain length seen so far."""
    root = Path(artifact_root)
    root.mkdir(parents=True, exist_ok=True)
    runs = {f"prefix_{int(p)}": evaluate(artifact_root=root / f"prefix_{int(p)}", start_index=1, end_index=int(p)) for p in prefixes}
    full = None if full_end_index is None else evaluate(artifact_root=root / f"full_{full_end_index}", start_index=1, end_index=int(full_end_index))

    def chain(summary: dict[str, Any] | None) -> int:
        return int(summary.get("longest_success_prefix", 0)) if summary else 0

    at120, at180 = runs.get("prefix_120"), runs.get("prefix_180")
    first_fail_180 = at180.get("first_failure_index") if at180 else 0
    checks = (
        ("prefix120_retention_failed", bool(at120) and chain(at120) >= 120 and float(at120.get("success_rate", 0.0)) >= min_prefix120_success_rate),
        ("prefix180_did_not_expand_beyond_120", bool(at180) and chain(at180) > 120),
        ("prefix180_failed_before_or_at_120", bool(at180) and (first_fail_180 is None or int(first_fail_180) > 120)),
        ("full_route_prefix_regressed_too_much", full is None or chain(full) >= int(best_full_longest_prefix - full_prefix_tolerance)),
    )
    rejected = [reason for reason, ok in checks if not ok]
    summary = {
        "schema_version": "v5.route_gate.v1", "checkpoint": str(checkpoint), "config": str(config), "route_path": str(route_path),
        "accepted": not rejected, "rejection_reasons": rejected,
        "criteria": {"min_prefix120_success_rate": float(min_prefix120_success_rate), "best_full_longest_prefix": int(best_full_longest_prefix),
                     "full_prefix_tolerance": int(full_prefix_tolerance)},
        "prefix_results": runs, "full_result": full,
    }
    (root / "route_gate_summary.json").write_text(json.dumps(summary, indent=2))
    return summary
