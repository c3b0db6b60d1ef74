"""Route prefix curriculum and the sequential route evaluator on the device engine.

Mirror of kinematic_phase1/route/route_curriculum.py:17-136 (``RoutePrefixCurriculumCallback``, ``build_prefix_stages``) and
kinematic_phase1/eval/eval_route_curriculum.py:57-248 (``_roll_one``, ``_summarize_rows``, ``_failure_reason``, ``_chunk_metrics``,
``evaluate_sequential_route``).  The callback is a plain object fed with the vectorised env's (done, info) arrays in env order --
what SB3 hands ``_on_step`` -- and calls ``env.set_route_window`` on promotion.  The sequential evaluator chains the final
(q, dq, prev_action) of waypoint k into waypoint k+1, so it is serial by construction: one device env, one step per launch pair.
"""
from __future__ import annotations

import ctypes as C

import json
from collections import deque
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


class RoutePrefixCurriculum:
    def __init__(self, *, stages: list[RouteCurriculumStage], promotion_success_rate: float, promotion_route_ready_hit_rate: float,
                 promotion_orientation_hit_rate: float, promotion_max_regression_rate: float, window_episodes: int, min_episodes_per_stage: int = 128) -> None:
        if not stages:
            raise ValueError("RoutePrefixCurriculumCallback requires at least one stage")
        self.stages = list(stages)
        self.promotion_success_rate = float(promotion_success_rate)
        self.promotion_route_ready_hit_rate = float(promotion_route_ready_hit_rate)
        self.promotion_orientation_hit_rate = float(promotion_orientation_hit_rate)
        self.promotion_max_regression_rate = float(promotion_max_regression_rate)
        self.window_episodes = max(int(window_episodes), 1)
        self.min_episodes_per_stage = max(int(min_episodes_per_stage), 1)
        self.current_stage_index = 0
        self.stage_episode_count = 0
        self.successes: deque[int] = deque(maxlen=self.window_episodes)
        self.ready_hits: deque[int] = deque(maxlen=self.window_episodes)
        self.orientation_hits: deque[int] = deque(maxlen=self.window_episodes)
        self.regressions: deque[int] = deque(maxlen=self.window_episodes)
        self.history: list[dict[str, object]] = []
        self.num_timesteps = 0
        self.training_env: Any = None

    @classmethod
    def from_config(cls, cfg: dict[str, Any], n_waypoints: int) -> "RoutePrefixCurriculum":
        """train_route_curriculum.py:86-88, 150-160"""
        cur = (cfg.get("route", {}) or {}).get("curriculum", {}) or {}
        return cls(stages=build_prefix_stages(rcfg.prefix_stages(cfg, n_waypoints)), promotion_success_rate=float(cur.get("promotion_success_rate", 0.80)),
                   promotion_route_ready_hit_rate=float(cur.get("promotion_route_ready_hit_rate", 0.80)),
                   promotion_orientation_hit_rate=float(cur.get("promotion_orientation_hit_rate", 0.90)),
                   promotion_max_regression_rate=float(cur.get("promotion_max_regression_rate", 0.20)),
                   window_episodes=int(cur.get("promotion_window_episodes", 256)), min_episodes_per_stage=int(cur.get("min_episodes_per_stage", 128)))

    def _apply_stage(self) -> None:
        stage = self.stages[self.current_stage_index]
        self.training_env.env_method("set_route_window", max_route_index=int(stage.prefix_end_index), min_route_index=1)

    def on_training_start(self, env: Any) -> None:
        self.training_env = env
        self._apply_stage()

    def _metrics(self) -> dict[str, float]:
        def mean(xs: deque[int]) -> float:
            return float(sum(xs)) / float(len(xs)) if xs else 0.0

        return {"recent_success_rate": mean(self.successes), "recent_route_ready_hit_rate": mean(self.ready_hits),
                "recent_orientation_hit_rate": mean(self.orientation_hits), "recent_regression_rate": mean(self.regressions)}

    def _promote(self, metrics: dict[str, float]) -> None:
        if self.current_stage_index >= len(self.stages) - 1:
            return
        prev = self.stages[self.current_stage_index]
        self.current_stage_index += 1
        nxt = self.stages[self.current_stage_index]
        self.history.append({"from_stage": prev.name, "to_stage": nxt.name, "from_prefix_end_index": int(prev.prefix_end_index),
                             "to_prefix_end_index": int(nxt.prefix_end_index), "total_timesteps": int(self.num_timesteps), **metrics})
        self.stage_episode_count = 0
        for d in (self.successes, self.ready_hits, self.orientation_hits, self.regressions):
            d.clear()
        self._apply_stage()

    def on_step(self, dones: Sequence[Any], success: Sequence[Any], route_ready: Sequence[Any], orientation_hit: Sequence[Any],
                regression: Sequence[Any]) -> bool:
        """_on_step over one vectorised step (arrays in env order; ``dones`` = terminated | truncated)."""
        self.num_timesteps += len(dones)
        for i, done in enumerate(dones):
            if not done:
                continue
            self.stage_episode_count += 1
            self.successes.append(1 if bool(success[i]) else 0)
            self.ready_hits.append(1 if bool(route_ready[i]) else 0)
            self.orientation_hits.append(1 if bool(orientation_hit[i]) else 0)
            self.regressions.append(1 if bool(regression[i]) else 0)
            if self.stage_episode_count < self.min_episodes_per_stage or len(self.successes) < self.window_episodes:
                continue
            m = self._metrics()
            if (m["recent_success_rate"] >= self.promotion_success_rate and m["recent_route_ready_hit_rate"] >= self.promotion_route_ready_hit_rate
                    and m["recent_orientation_hit_rate"] >= self.promotion_orientation_hit_rate
                    and m["recent_regression_rate"] <= self.promotion_max_regression_rate):
                self._promote(m)
        return True

    def observe_env(self, env: RouteVecEnv) -> bool:
        """on_step with the device env's last step (done bits + info arrays copied to the host)."""
        done = env.done.cpu().numpy()
        info = env.info()
        return self.on_step((done & 3) != 0, (done & 4) != 0, info["route_ready"].cpu().numpy(), info["route_orientation_hit"].cpu().numpy(),
                            info["route_regression"].cpu().numpy())

    def observe_step(self, env: RouteVecEnv, done_bits: torch.Tensor) -> bool:
        """PPO.step_callback form: ``done_bits`` = the rollout buffer row of this step (step_into does not fill env.done)."""
        packed = torch.cat([done_bits.view(1, -1), env.episode_flags()]).cpu().numpy()     # one device->host copy per step
        done = packed[0]
        if not (done & 3).any():
            self.num_timesteps += len(done)
            return True
        return self.on_step((done & 3) != 0, (done & 4) != 0, packed[1], packed[2], packed[3])

    def summary(self) -> dict[str, object]:
        stage = self.stages[self.current_stage_index]
        return {"stage_index": int(self.current_stage_index), "stage_name": stage.name, "prefix_end_index": int(stage.prefix_end_index),
                "stage_episode_count": int(self.stage_episode_count), **self._metrics(), "history": list(self.history)}


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
        host = RoutePrefixCurriculum.from_config(cfg, n_waypoints)
        return cls(stages=host.stages, promotion_success_rate=host.promotion_success_rate, promotion_route_ready_hit_rate=host.promotion_route_ready_hit_rate,
                   promotion_orientation_hit_rate=host.promotion_orientation_hit_rate, promotion_max_regression_rate=host.promotion_max_regression_rate,
                   window_episodes=host.window_episodes, min_episodes_per_stage=host.min_episodes_per_stage)

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
              success_dwell_steps: int, record: list[tuple[np.ndarray, np.ndarray]] | None = None) -> dict[str, Any]:
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
    while not (done & 3):
        action = policy(obs)
        if record is not None:   # the observation the action was computed from (collect_route_teacher_rollout.py:68-73)
            record.append((obs[0, :env.obs_dim].cpu().numpy().astype(np.float32), action[0].detach().cpu().numpy().astype(np.float32)))
        obs, _, d = env.step(action, auto_reset=False)
        done = int(d[0])
        steps += 1
        info = env.info()
        min_pos = min(min_pos, float(info["position_error_norm"][0]))
        min_ori = min(min_ori, float(info["orientation_error_norm"][0]))
        min_q = min(min_q, float(info["route_q_error_norm"][0]))
        if bool(info["route_ready"][0]) and first_ready_step is None:
            first_ready_step = steps
        max_ready_streak = max(max_ready_streak, int(info["route_ready_streak"][0]))
    st = env.get_state()
    return {
        "route_index": int(goal_index), "success": bool(done & 4), "route_ready_hit": bool(first_ready_step is not None),
        "route_ready_dwell": bool(max_ready_streak >= success_dwell_steps), "first_ready_step": first_ready_step, "max_ready_streak": int(max_ready_streak),
        "steps": int(steps), "final_position_error": float(info["position_error_norm"][0]),
        "final_orientation_error": float(info["orientation_error_norm"][0]), "final_q_error": float(info["route_q_error_norm"][0]),
        "min_position_error": float(min_pos), "min_orientation_error": float(min_ori), "min_q_error": float(min_q),
        "final_action_magnitude": float(info["action_l2"][0]), "final_dq_norm": float(info["executed_delta_q_l2"][0]),
        "final_q": st["q"][0].copy(), "final_dq": st["dq"][0].copy(), "final_prev_action": st["prev_action"][0].copy(),
    }


def _failure_reason(row: dict[str, Any]) -> str:
    if row["final_position_error"] > 0.010:
        return "position"
    if row["final_orientation_error"] > 0.150:
        return "orientation"
    if row.get("final_action_magnitude", 0.0) > 1.20 or row.get("final_dq_norm", 0.0) > 0.040:
        return "motion_action"
    if row["final_q_error"] > 0.500:
        return "q_error"
    if not row["route_ready_dwell"]:
        return "dwell_or_motion"
    return "unknown"


def summarize_rows(rows: list[dict[str, Any]], route_progress_m: np.ndarray) -> dict[str, Any]:
    if not rows:
        return {"target_count": 0}
    first_failure = next((row for row in rows if not row["success"]), None)
    longest_prefix = 0
    for row in rows:
        if row["success"]:
            longest_prefix += 1
        else:
            break
    prefix_end = min(longest_prefix, len(route_progress_m) - 1)
    return {
        "target_count": len(rows), "success_rate": float(np.mean([row["success"] for row in rows])),
        "route_ready_hit_rate": float(np.mean([row["route_ready_hit"] for row in rows])),
        "route_ready_dwell_rate": float(np.mean([row["route_ready_dwell"] for row in rows])), "longest_success_prefix": int(longest_prefix),
        "cumulative_successful_route_distance_m": float(route_progress_m[prefix_end] - route_progress_m[0]),
        "first_failure_index": None if first_failure is None else int(first_failure["route_index"]),
        "first_failure_reason": None if first_failure is None else _failure_reason(first_failure),
        "mean_final_position_error": float(np.mean([row["final_position_error"] for row in rows])),
        "mean_final_orientation_error": float(np.mean([row["final_orientation_error"] for row in rows])),
        "mean_final_q_error": float(np.mean([row["final_q_error"] for row in rows])),
        "max_final_position_error": float(np.max([row["final_position_error"] for row in rows])),
        "max_final_orientation_error": float(np.max([row["final_orientation_error"] for row in rows])),
    }


def chunk_metrics(rows: list[dict[str, Any]]) -> dict[str, Any]:
    chunks = [(1, 40), (41, 80), (81, 120), (121, 180), (181, 260), (261, 360), (361, 483)]
    out: dict[str, Any] = {}
    for idx, (lo, hi) in enumerate(chunks):
        subset = [row for row in rows if lo <= row["route_index"] <= hi]
        if not subset:
            continue
        out[f"chunk_{idx}_{lo}_{hi}"] = {
            "target_count": len(subset), "success_rate": float(np.mean([row["success"] for row in subset])),
            "route_ready_hit_rate": float(np.mean([row["route_ready_hit"] for row in subset])),
            "mean_final_position_error": float(np.mean([row["final_position_error"] for row in subset])),
            "mean_final_orientation_error": float(np.mean([row["final_orientation_error"] for row in subset])),
            "mean_final_q_error": float(np.mean([row["final_q_error"] for row in subset])),
        }
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
def _passes_prefix120(summary: dict[str, Any], *, min_success_rate: float) -> bool:
    return int(summary.get("longest_success_prefix", 0)) >= 120 and float(summary.get("success_rate", 0.0)) >= min_success_rate


def evaluate_route_gate(*, evaluate: Callable[..., dict[str, Any]], artifact_root: str | Path, prefixes: Sequence[int], full_end_index: int | None,
                        min_prefix120_success_rate: float, best_full_longest_prefix: int, full_prefix_tolerance: int, checkpoint: str = "",
                        config: str = "", route_path: str = "") -> dict[str, Any]:
    """eval/eval_route_gate.py:17-99.  ``evaluate(artifact_root=, start_index=, end_index=)`` runs one sequential evaluation (the trainer
    binds evaluate_sequential_route to the policy under test); the accept / reject rules and the summary file are the reference's."""
    root = Path(artifact_root)
    root.mkdir(parents=True, exist_ok=True)
    prefix_results: dict[str, Any] = {}
    for prefix in prefixes:
        prefix_results[f"prefix_{prefix}"] = evaluate(artifact_root=root / f"prefix_{prefix}", start_index=1, end_index=int(prefix))
    full_summary = None
    if full_end_index is not None:
        full_summary = evaluate(artifact_root=root / f"full_{full_end_index}", start_index=1, end_index=int(full_end_index))
    p120 = prefix_results.get("prefix_120")
    p180 = prefix_results.get("prefix_180")
    prefix120_retained = bool(p120 and _passes_prefix120(p120, min_success_rate=min_prefix120_success_rate))
    expands_beyond_120 = bool(p180 and int(p180.get("longest_success_prefix", 0)) > 120)
    first_failure_not_before_120 = bool(p180 and (p180.get("first_failure_index") is None or int(p180.get("first_failure_index", 0)) > 120))
    full_not_too_regressed = True
    if full_summary is not None:
        full_not_too_regressed = int(full_summary.get("longest_success_prefix", 0)) >= int(best_full_longest_prefix - full_prefix_tolerance)
    accepted = bool(prefix120_retained and expands_beyond_120 and first_failure_not_before_120 and full_not_too_regressed)
    reasons: list[str] = []
    if not prefix120_retained:
        reasons.append("prefix120_retention_failed")
    if not expands_beyond_120:
        reasons.append("prefix180_did_not_expand_beyond_120")
    if not first_failure_not_before_120:
        reasons.append("prefix180_failed_before_or_at_120")
    if not full_not_too_regressed:
        reasons.append("full_route_prefix_regressed_too_much")
    summary = {
        "schema_version": "v5.route_gate.v1", "checkpoint": str(checkpoint), "config": str(config), "route_path": str(route_path), "accepted": accepted,
        "rejection_reasons": reasons,
        "criteria": {"min_prefix120_success_rate": float(min_prefix120_success_rate), "best_full_longest_prefix": int(best_full_longest_prefix),
                     "full_prefix_tolerance": int(full_prefix_tolerance)},
        "prefix_results": prefix_results, "full_result": full_summary,
    }
    (root / "route_gate_summary.json").write_text(json.dumps(summary, indent=2))
    return summary


# --------------------------------------------------------------------------------------------- teacher-anchor dataset
def collect_teacher_rollout(*, policy: PolicyFn, cfg: dict[str, Any], route_q: np.ndarray, artifact_root: str | Path, start_index: int = 1, end_index: int = 120,
                            device: int = 0, checkpoint: str = "", config: str = "", route_path: str = "") -> dict[str, Any]:
    """route/collect_route_teacher_rollout.py:22-123: the sequential evaluator's chained episodes with every (observation, action) pair
    recorded; the walk stops at the first failed waypoint and that episode's samples are dropped.  Writes
    ``teacher_route_anchor_dataset.npz`` (``obs__<key>``, ``actions``, ``route_index``, ``step``) + the summary JSON."""
    W = int(route_q.shape[0])
    seq_off = {**cfg, "route": {**(cfg.get("route", {}) or {}), "sequence": {**((cfg.get("route", {}) or {}).get("sequence", {}) or {}), "enabled": False}}}
    base = kcfg.to_env_config(cfg)
    env = RouteVecEnv(base, rcfg.route_config_from_dict(seq_off, max_route_index=end_index), route_q, 1, device=device, seed=0)
    layout = rcfg.ROUTE_OBS_LAYOUT if env.obs_dim == rcfg.ROUTE_OBS_DIM else kcfg.OBS_LAYOUT
    obs_rows: list[np.ndarray] = []
    action_rows: list[np.ndarray] = []
    meta: list[tuple[int, int]] = []
    ok: list[int] = []
    failed: list[int] = []
    cq = np.asarray(route_q[max(start_index - 1, 0)], dtype=float).copy()
    cdq, cpa = np.zeros_like(cq), np.zeros_like(cq)
    dwell = int(base.c.termination.success_dwell_steps)
    for idx in range(int(start_index), min(int(end_index), W - 1) + 1):
        rec: list[tuple[np.ndarray, np.ndarray]] = []
        row = _roll_one(env, policy, initial_q=cq, goal_index=idx, initial_dq=cdq, initial_prev_action=cpa, success_dwell_steps=dwell, record=rec)
        cq, cdq, cpa = row["final_q"], row["final_dq"], row["final_prev_action"]
        if not row["success"]:
            failed.append(idx)
            break
        ok.append(idx)
        for step, (o, a) in enumerate(rec):
            obs_rows.append(o)
            action_rows.append(a)
            meta.append((idx, step))
    env.close()
    root = Path(artifact_root)
    root.mkdir(parents=True, exist_ok=True)
    dataset_path = root / "teacher_route_anchor_dataset.npz"
    obs_mat = np.asarray(obs_rows, dtype=np.float32).reshape(len(obs_rows), sum(w for _, w in layout.values()))
    arrays: dict[str, np.ndarray] = {"actions": np.asarray(action_rows, dtype=np.float32).reshape(len(action_rows), kcfg.NJ),
                                     "route_index": np.asarray([m[0] for m in meta], dtype=np.int32), "step": np.asarray([m[1] for m in meta], dtype=np.int32)}
    for key, (off, width) in layout.items():
        arrays[f"obs__{key}"] = obs_mat[:, off:off + width]
    np.savez_compressed(dataset_path, **arrays)
    summary = {"schema_version": "v5.route_teacher_anchor_dataset.v1", "checkpoint": str(checkpoint), "config": str(config), "route_path": str(route_path),
               "dataset_path": str(dataset_path), "start_index": int(start_index), "requested_end_index": int(end_index), "successful_indices": ok,
               "failed_indices": failed, "sample_count": int(len(action_rows)), "obs_keys": sorted(layout.keys()),
               "action_dim": int(kcfg.NJ) if action_rows else 0}
    (root / "teacher_route_anchor_summary.json").write_text(json.dumps(summary, indent=2))
    return summary
