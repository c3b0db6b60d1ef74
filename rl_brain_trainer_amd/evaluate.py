"""Deterministic Approach -> Finisher evaluation, batched on the GPU.

Mirrors the reference's serial evaluators (one env + batch-1 ``model.predict`` per episode) as one batched state
machine over all episodes of all stages:

* ``build_curriculum_local_eval_suite``   kinematic_phase1/eval/fixed_eval_suite.py:78-105
* ``dock_coarse_ready`` / ``finisher_ready`` eval_three_stage.py:41-56, eval_approach_finisher.py:24-32
* ``run_approach_with_handoff``           eval_pipeline_ablation.py:60-147
* ``run_policy`` (finisher episode)       eval_three_stage.py:59-125, reset from ``_state_reset_options`` :30-38
* ``evaluate_workspace_expansion``        eval_workspace_expansion.py:86-211 (same stage_metrics / selection JSON)
* ``gated_score`` & friends               workspace/workspace_curriculum.py:10-95

Suites are drawn on the host with numpy's PCG64 exactly like the reference (``default_rng(seed + 1009 * stage)``); the
goal poses come from the fp64 HIP FK kernel.
"""
from __future__ import annotations

import ctypes as C
import json
from pathlib import Path
from typing import Any, Callable

import numpy as np
import torch

from . import config as kcfg
from . import native
from .vec_env import ArmKinematicVecEnv, fk_pose6

PolicyFn = Callable[[torch.Tensor], torch.Tensor]  # obs [E, stride] -> clipped deterministic action [E, 7]


# ----------------------------------------------------------------------------- gate scoring (host logic)
# The YAML ``gate:`` block and the ``best_model_selection`` JSON are schemas other tools read (check_workspace_expansion_status.sh,
# plot_workspace_expansion.py): key names and default values are the reference's (workspace/workspace_curriculum.py:20-32).  The
# evaluation itself works on one small numeric table per call instead of walking dicts stage by stage.
_GATE_DEFAULTS: dict[str, Any] = {
    "retention_stage0_4_success": 0.95, "retention_stage5_success": 0.85, "retention_stage_thresholds": (),
    "promotion_stage_success": 0.80, "promotion_ready_rate": 0.80, "max_mean_position_error_m": 0.020, "max_mean_orientation_error_rad": 0.15,
    "score_current_success_weight": 0.45, "score_current_ready_weight": 0.20, "score_retention_weight": 0.20, "score_error_weight": 0.15,
}
_METRIC_COLUMNS = ("success_rate", "finisher_ready_hit_rate", "mean_final_position_error", "mean_final_orientation_error")


class WorkspaceGate:
    """Thresholds and score weights of the workspace-expansion eval gate (attribute per YAML key)."""

    def __init__(self, **overrides: Any) -> None:
        for key, default in _GATE_DEFAULTS.items():
            value = overrides.get(key, default)
            setattr(self, key, tuple(float(v) for v in value) if key == "retention_stage_thresholds" else float(value))

    def as_dict(self) -> dict[str, Any]:
        return {k: getattr(self, k) for k in _GATE_DEFAULTS}


WorkspaceGateConfig = WorkspaceGate   # name the trainers import


def gate_config_from_dict(payload: dict[str, Any] | None) -> WorkspaceGate:
    return WorkspaceGate(**{k: v for k, v in dict(payload or {}).items() if k in _GATE_DEFAULTS})


class _StageTable:
    """stage_metrics {stage index -> summary dict} as columns: `idx` sorted ascending, one float column per gate metric."""

    def __init__(self, stage_metrics: dict[int, dict[str, Any]], missing: tuple[float, float, float, float]) -> None:
        self.idx = np.array(sorted(int(k) for k in stage_metrics), dtype=np.int64)
        rows = [[float(stage_metrics[int(i)].get(name, fill)) for name, fill in zip(_METRIC_COLUMNS, missing)] for i in self.idx]
        self.col = np.array(rows, dtype=np.float64).reshape(len(rows), len(_METRIC_COLUMNS))

    def lookup(self, stages: np.ndarray, column: int, fill: float) -> np.ndarray:
        """column values at the given stage indices; stages without a row read as `fill`"""
        out = np.full(len(stages), fill, dtype=np.float64)
        if self.idx.size:
            pos = np.searchsorted(self.idx, stages)
            hit = (pos < self.idx.size) & (self.idx[np.minimum(pos, self.idx.size - 1)] == stages)
            out[hit] = self.col[pos[hit], column]
        return out


def _passed_mask(tab: _StageTable, gate: WorkspaceGate) -> np.ndarray:
    """per evaluated stage: success, ready-hit rate and both mean errors inside the promotion thresholds"""
    c = tab.col
    return (c[:, 0] >= gate.promotion_stage_success) & (c[:, 1] >= gate.promotion_ready_rate) & \
           (c[:, 2] <= gate.max_mean_position_error_m) & (c[:, 3] <= gate.max_mean_orientation_error_rad)


def stage_passed(metrics: dict[str, Any], gate: WorkspaceGate) -> bool:
    return bool(_passed_mask(_StageTable({0: metrics}, (0.0, 0.0, 999.0, 999.0)), gate)[0])


def retention_ok(stage_metrics: dict[int, dict[str, Any]], gate: WorkspaceGate) -> bool:
    """old stages must keep their success rate: per-stage thresholds where the config lists them (only stages that were evaluated count),
    else stages 0-4 and stage 5 against the two fixed levels (a stage that was not evaluated counts as 0)"""
    tab = _StageTable(stage_metrics, (0.0, 0.0, 999.0, 999.0))
    if gate.retention_stage_thresholds:
        thr = np.asarray(gate.retention_stage_thresholds, dtype=np.float64)
        listed = tab.idx[(tab.idx >= 0) & (tab.idx < thr.size)]
        return bool(np.all(tab.lookup(listed, 0, 0.0) >= thr[listed]))
    succ = tab.lookup(np.arange(6), 0, 0.0)
    return bool(np.all(succ[:5] >= gate.retention_stage0_4_success) and succ[5] >= gate.retention_stage5_success)


def highest_passed_stage(stage_metrics: dict[int, dict[str, Any]], gate: WorkspaceGate) -> int:
    """largest passing stage index below the first FAILING expansion stage (index >= 6); -1 when none passes"""
    tab = _StageTable(stage_metrics, (0.0, 0.0, 999.0, 999.0))
    ok = _passed_mask(tab, gate)
    blockers = np.flatnonzero(~ok & (tab.idx >= 6))
    reach = int(blockers[0]) if blockers.size else tab.idx.size
    good = tab.idx[:reach][ok[:reach]]
    return int(good[-1]) if good.size else -1


def gated_score(stage_metrics: dict[int, dict[str, Any]], current_stage: int, gate: WorkspaceGate) -> dict[str, Any]:
    """best-checkpoint score of one evaluation: weighted current-stage success / ready-hit rate, retention over stages 0..min(5, current)
    and an error score; the current stage's missing errors count as 1.0 (m / rad)"""
    tab = _StageTable(stage_metrics, (0.0, 0.0, 1.0, 1.0))
    stage = np.array([int(current_stage)])
    cur_succ, cur_ready, cur_pos, cur_ori = (float(tab.lookup(stage, k, fill)[0]) for k, fill in enumerate((0.0, 0.0, 1.0, 1.0)))
    kept = tab.lookup(np.arange(0, min(6, int(current_stage) + 1)), 0, 0.0)
    retention = float(kept.sum() / kept.size) if kept.size else 0.0
    limits = np.maximum([gate.max_mean_position_error_m, gate.max_mean_orientation_error_rad], 1e-6)
    error_score = float(0.5 * np.maximum(0.0, 1.0 - np.array([cur_pos, cur_ori]) / limits).sum())
    score = cur_succ * gate.score_current_success_weight + cur_ready * gate.score_current_ready_weight \
        + retention * gate.score_retention_weight + error_score * gate.score_error_weight
    return {
        "score": float(score), "current_stage": int(current_stage), "retention_ok": retention_ok(stage_metrics, gate),
        "highest_passed_stage": highest_passed_stage(stage_metrics, gate), "current_stage_success_rate": cur_succ,
        "current_stage_ready_rate": cur_ready, "retention_mean_success_rate": retention, "error_score": error_score,
    }


# ----------------------------------------------------------------------------- suites
def build_curriculum_local_eval_suite(env_cfg: kcfg.EnvConfig, *, seed: int = 700001, stage_index: int = 0, n_episodes: int = 10,
                                      device: torch.device | int = 0) -> dict[str, np.ndarray]:
    """Per episode: start = clip(start_q + U(-start_noise, start_noise)) (draws only if any noise > 0), goal likewise."""
    c = env_cfg.c
    if not c.curriculum_enabled:
        raise ValueError("Curriculum-local eval requires curriculum to be enabled")
    if c.n_stages <= 0:
        raise ValueError("No curriculum stages are defined")
    idx = int(np.clip(stage_index, 0, c.n_stages - 1))
    st = c.stages[idx]
    lo, hi = np.array(c.joints.lower[:]), np.array(c.joints.upper[:])
    rng = np.random.default_rng(seed)

    def sample(base, noise):
        base = np.asarray(base, dtype=float)
        noise = np.asarray(noise, dtype=float)
        if np.any(noise > 0.0):
            base = base + rng.uniform(low=-noise, high=noise)
        return np.clip(base, lo, hi)

    init = np.empty((n_episodes, 7))
    goal = np.empty((n_episodes, 7))
    for e in range(n_episodes):
        init[e] = sample(st.start_q[:], st.start_noise[:])
        goal[e] = sample(st.goal_q[:], st.goal_noise[:])
    dev = torch.device("cuda", device) if isinstance(device, int) else device
    goal_pose6 = fk_pose6(torch.tensor(goal, dtype=torch.float64, device=dev)).cpu().numpy()
    return {"initial_q": init, "goal_q": goal, "goal_pose6": goal_pose6, "stage_index": np.full(n_episodes, idx)}


# ----------------------------------------------------------------------------- readiness predicates
def _ready(pos, ori, action_norm, dq_norm, pos_thr, ori_thr, act_thr, dq_thr):
    if not (pos_thr > 0.0 and ori_thr > 0.0):
        return torch.zeros_like(pos, dtype=torch.bool)
    ok = (pos <= pos_thr) & (ori <= ori_thr)
    if act_thr > 0.0:
        ok &= action_norm <= act_thr
    if dq_thr > 0.0:
        ok &= dq_norm <= dq_thr
    return ok


def dock_coarse_ready(pos, ori, action_norm, dq_norm, r) -> torch.Tensor:
    return _ready(pos, ori, action_norm, dq_norm, r.dock_coarse_ready_pos_threshold_m, r.dock_coarse_ready_ori_threshold_rad,
                  r.dock_coarse_ready_action_threshold, r.dock_coarse_ready_dq_threshold)


def finisher_ready(pos, ori, action_norm, dq_norm, r) -> torch.Tensor:
    return _ready(pos, ori, action_norm, dq_norm, r.finisher_ready_pos_threshold_m, r.finisher_ready_ori_threshold_rad,
                  r.finisher_ready_action_threshold, r.finisher_ready_dq_threshold)


# ----------------------------------------------------------------------------- batched episode runners
def _snapshot(env: ArmKinematicVecEnv) -> dict[str, torch.Tensor]:
    info = env.info()
    return {k: info[k].t().clone().double() for k in ("q", "dq", "prev_action", "goal_q", "goal_pose6")}


_STATE_SLICES = (("q", 0, 7), ("dq", 7, 14), ("prev_action", 14, 21), ("goal_q", 21, 28), ("goal_pose6", 28, 34))
_ALIVE_CHECK_EVERY = 8      # env steps between two reads of the device's alive-episode counter


def run_episodes(env: ArmKinematicVecEnv, policy: PolicyFn, reset_options: dict[str, Any], *, ready_cfg=None, handoff_confirm_steps: int | None = None,
                 active: torch.Tensor | None = None, max_steps: int | None = None) -> tuple[dict[str, torch.Tensor], dict[str, torch.Tensor] | None]:
    """All episodes in lock step until each has terminated or truncated (_run_policy / _run_approach_with_handoff).

    Returns (final_result, handoff_result): tensors over episodes.  ``handoff_confirm_steps=None`` is the reference's _run_policy (no
    handoff bookkeeping, handoff_result None); an integer is _run_approach_with_handoff: handoff_result is the snapshot at the first
    step where ``ready_streak >= handoff_confirm_steps`` (eval_pipeline_ablation.py:103 -- so 0 or less hands over at step 1), with
    ``valid`` marking who has one, and carries the action / dq means up to that step like the reference's dict.

    Per env step: the policy, the norm of its action, kp1_step, and ONE kp1_eval_accumulate launch that does the whole per-episode
    bookkeeping on the device (finals / minima / sums, success, state snapshot, ready streak, first-confirmed handoff snapshot, alive
    mask); the host looks at the alive-episode counter every _ALIVE_CHECK_EVERY steps.  Results are bit-identical to the tensor-expression
    form (_run_episodes_reference; tests/test_eval_checkpoint_gpu.py)."""
    if not isinstance(env, ArmKinematicVecEnv):
        raise TypeError("run_episodes drives an ArmKinematicVecEnv (kp1_eval_accumulate reads its handle); the route wrappers have their own evaluator")
    E = env.n_envs
    dev = env.device
    L = native.load()
    env.use_current_stream()
    obs = env.reset(options=reset_options).clone()
    f64, i32, u8 = torch.float64, torch.int32, torch.uint8
    metrics = torch.empty((8, E), dtype=f64, device=dev)
    counters = torch.empty((4, E), dtype=i32, device=dev)
    flags = torch.empty((4, E), dtype=u8, device=dev)
    state = torch.empty((E, 34), dtype=f64, device=dev)
    n_alive = torch.zeros(1, dtype=i32, device=dev)
    want_hand = handoff_confirm_steps is not None and ready_cfg is not None
    confirm = int(handoff_confirm_steps or 0)
    hand_metrics = torch.empty((8, E), dtype=f64, device=dev) if want_hand else None
    hand_step = torch.empty(E, dtype=i32, device=dev) if want_hand else None
    hand_success = torch.empty(E, dtype=u8, device=dev) if want_hand else None
    hand_state = torch.empty((E, 34), dtype=f64, device=dev) if want_hand else None
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
    bufs = native.EvalBuffers(ptr(metrics), ptr(counters), ptr(flags), ptr(state), ptr(hand_metrics), ptr(hand_step), ptr(hand_success), ptr(hand_state),
                              ptr(n_alive))
    thr = None
    if ready_cfg is not None:
        thr = (C.c_double * 4)(ready_cfg.dock_coarse_ready_pos_threshold_m, ready_cfg.dock_coarse_ready_ori_threshold_rad,
                               ready_cfg.dock_coarse_ready_action_threshold, ready_cfg.dock_coarse_ready_dq_threshold)
    act_mask = None if active is None else active.to(dev).to(u8).contiguous()
    native.check(L.kp1_eval_accumulate(env._handle, C.byref(bufs), None, None, ptr(act_mask), 0, thr, confirm, None))
    limit = int(max_steps or (env.config.c.termination.max_episode_steps + 1))
    for step in range(1, limit + 1):
        if (step - 1) % _ALIVE_CHECK_EVERY == 0 and int(n_alive.item()) == 0:
            break
        action = policy(obs)
        a_norm = torch.linalg.vector_norm(action.double(), dim=1).contiguous()
        obs, _, done = env.step(action, auto_reset=False)
        obs = obs.clone()
        native.check(L.kp1_eval_accumulate(env._handle, C.byref(bufs), ptr(a_norm), ptr(done), None, step, thr, confirm, None))
    res = {
        "success": flags[1].bool(), "final_position_error": metrics[0], "final_orientation_error": metrics[1], "min_position_error": metrics[2],
        "min_orientation_error": metrics[3], "final_action_magnitude": metrics[4], "final_dq_norm": metrics[5], "sum_action": metrics[6],
        "sum_dq": metrics[7], "ready_hit": flags[2].bool(), "max_ready_streak": counters[1], "first_ready_step": counters[2], "step_count": counters[0],
    }
    steps = res["step_count"].clamp_min(1).double()
    res["mean_action_magnitude"] = res["sum_action"] / steps
    res["mean_dq_norm"] = res["sum_dq"] / steps
    for k, lo, hi in _STATE_SLICES:
        res["state_" + k] = state[:, lo:hi].contiguous()
    hand = None
    if want_hand:
        hand = {"valid": flags[3].bool(), "final_position_error": hand_metrics[0], "final_orientation_error": hand_metrics[1],
                "final_action_magnitude": hand_metrics[2], "final_dq_norm": hand_metrics[3], "min_position_error": hand_metrics[4],
                "min_orientation_error": hand_metrics[5], "step_count": hand_step, "success": hand_success.bool()}
        hsteps = hand_step.clamp_min(1).double()
        hand["mean_action_magnitude"] = hand_metrics[6] / hsteps
        hand["mean_dq_norm"] = hand_metrics[7] / hsteps
        for k, lo, hi in _STATE_SLICES:
            hand["state_" + k] = hand_state[:, lo:hi].contiguous()
    return res, hand


def _run_episodes_reference(env: ArmKinematicVecEnv, policy: PolicyFn, reset_options: dict[str, Any], *, ready_cfg=None, handoff_confirm_steps: int | None = None,
                 active: torch.Tensor | None = None, max_steps: int | None = None) -> tuple[dict[str, torch.Tensor], dict[str, torch.Tensor] | None]:
    """The per-step bookkeeping as tensor expressions (about 45 launches and two host synchronisations per env step): the form
    run_episodes had before kp1_eval_accumulate, kept as the test reference the device kernel is compared with bit for bit.

    All episodes in lock step until each has terminated or truncated (_run_policy / _run_approach_with_handoff).

    Returns (final_result, handoff_result) as run_episodes does."""
    E = env.n_envs
    dev = env.device
    obs = env.reset(options=reset_options).clone()
    info = env.info()
    f64 = torch.float64
    alive = torch.ones(E, dtype=torch.bool, device=dev) if active is None else active.clone().to(dev)
    pos0 = info["position_error_norm"].double().clone()
    ori0 = info["orientation_error_norm"].double().clone()
    res = {
        "success": torch.zeros(E, dtype=torch.bool, device=dev), "final_position_error": pos0.clone(), "final_orientation_error": ori0.clone(),
        "min_position_error": pos0.clone(), "min_orientation_error": ori0.clone(),
        "final_action_magnitude": torch.zeros(E, dtype=f64, device=dev), "final_dq_norm": torch.zeros(E, dtype=f64, device=dev),
        "sum_action": torch.zeros(E, dtype=f64, device=dev), "sum_dq": torch.zeros(E, dtype=f64, device=dev),
        "ready_hit": torch.zeros(E, dtype=torch.bool, device=dev), "max_ready_streak": torch.zeros(E, dtype=torch.int32, device=dev),
        "first_ready_step": torch.full((E,), -1, dtype=torch.int32, device=dev), "step_count": torch.zeros(E, dtype=torch.int32, device=dev),
    }
    streak = torch.zeros(E, dtype=torch.int32, device=dev)
    state = _snapshot(env)
    hand = None
    if handoff_confirm_steps is not None and ready_cfg is not None:
        hand = {"valid": torch.zeros(E, dtype=torch.bool, device=dev)}
        hsum = {"a": torch.zeros(E, dtype=f64, device=dev), "d": torch.zeros(E, dtype=f64, device=dev)}
    limit = int(max_steps or (env.config.c.termination.max_episode_steps + 1))
    for step in range(1, limit + 1):
        if not bool(alive.any()):
            break
        action = policy(obs)
        a_norm = torch.linalg.vector_norm(action.double(), dim=1)
        obs, _, done = env.step(action, auto_reset=False)
        obs = obs.clone()
        info = env.info()
        pos = info["position_error_norm"].double()
        ori = info["orientation_error_norm"].double()
        dqn = info["executed_delta_q_l2"].double()
        upd = alive
        res["step_count"] = torch.where(upd, torch.full_like(res["step_count"], step), res["step_count"])
        res["sum_action"] += torch.where(upd, a_norm, torch.zeros_like(a_norm))
        res["sum_dq"] += torch.where(upd, dqn, torch.zeros_like(dqn))
        for key, val in (("final_position_error", pos), ("final_orientation_error", ori), ("final_action_magnitude", a_norm), ("final_dq_norm", dqn)):
            res[key] = torch.where(upd, val, res[key])
        res["min_position_error"] = torch.where(upd, torch.minimum(res["min_position_error"], pos), res["min_position_error"])
        res["min_orientation_error"] = torch.where(upd, torch.minimum(res["min_orientation_error"], ori), res["min_orientation_error"])
        res["success"] = torch.where(upd, (done & 4) != 0, res["success"])
        snap = _snapshot(env)
        for k in state:
            state[k] = torch.where(upd[:, None], snap[k], state[k])
        if ready_cfg is not None:
            rdy = dock_coarse_ready(pos, ori, a_norm, dqn, ready_cfg) & upd
            res["ready_hit"] |= rdy
            res["first_ready_step"] = torch.where(rdy & (res["first_ready_step"] < 0), torch.full_like(res["first_ready_step"], step), res["first_ready_step"])
            streak = torch.where(upd, torch.where(rdy, streak + 1, torch.zeros_like(streak)), streak)
            res["max_ready_streak"] = torch.maximum(res["max_ready_streak"], streak)
            if hand is not None:
                take = upd & (~hand["valid"]) & (streak >= handoff_confirm_steps)
                hsum["a"] = torch.where(take, res["sum_action"], hsum["a"])
                hsum["d"] = torch.where(take, res["sum_dq"], hsum["d"])
                if bool(take.any()):
                    cur = {"final_position_error": pos, "final_orientation_error": ori, "final_action_magnitude": a_norm, "final_dq_norm": dqn,
                           "min_position_error": res["min_position_error"], "min_orientation_error": res["min_orientation_error"],
                           "step_count": torch.full_like(res["step_count"], step), "success": (done & 4) != 0, **{"state_" + k: v for k, v in snap.items()}}
                    for k, v in cur.items():
                        if k not in hand:
                            hand[k] = torch.zeros_like(v)
                        m = take if v.ndim == 1 else take[:, None]
                        hand[k] = torch.where(m, v, hand[k])
                    hand["valid"] |= take
        alive = alive & ((done & 3) == 0)
    steps = res["step_count"].clamp_min(1).double()
    res["mean_action_magnitude"] = res["sum_action"] / steps
    res["mean_dq_norm"] = res["sum_dq"] / steps
    for k, v in state.items():
        res["state_" + k] = v
    if hand is not None and "step_count" in hand:
        hsteps = hand["step_count"].clamp_min(1).double()
        hand["mean_action_magnitude"] = hsum["a"] / hsteps
        hand["mean_dq_norm"] = hsum["d"] / hsteps
    return res, hand


def _handoff_options(src: dict[str, torch.Tensor], mode: str) -> dict[str, Any]:
    """_state_reset_options; eval_three_stage.py:30-38"""
    return {"initial_q": src["state_q"].cpu().numpy(), "initial_dq": src["state_dq"].cpu().numpy(),
            "initial_prev_action": src["state_prev_action"].cpu().numpy(), "goal_q": src["state_goal_q"].cpu().numpy(),
            "goal_pose6": src["state_goal_pose6"].cpu().numpy(), "policy_mode": mode}


def _mean(x) -> float:
    x = np.asarray(x, dtype=float)
    return float(x.mean()) if x.size else 0.0


def evaluate_workspace_expansion(*, approach_policy: PolicyFn, finisher_policy: PolicyFn | None, approach_cfg: kcfg.EnvConfig,
                                 finisher_cfg: kcfg.EnvConfig | None, episodes: int = 50, seed: int = 700001,
                                 stage_indices: list[int] | None = None, handoff_confirm_steps: int = 2, gate_config: dict[str, Any] | None = None,
                                 artifact_root: str | Path | None = None, device: int = 0, obs_stride: int = 56) -> dict[str, Any]:
    """evaluate_workspace_expansion_checkpoint with policies passed as callables (checkpoint loading is the caller's)."""
    n_stages = approach_cfg.n_stages
    stages = stage_indices if stage_indices is not None else list(range(n_stages))
    stages = [int(np.clip(s, 0, n_stages - 1)) for s in stages]
    suites = [build_curriculum_local_eval_suite(approach_cfg, seed=seed + s * 1009, stage_index=s, n_episodes=episodes, device=device) for s in stages]
    cat = {k: np.concatenate([su[k] for su in suites]) for k in suites[0]}
    E = cat["initial_q"].shape[0]
    r = approach_cfg.c.reward

    env = ArmKinematicVecEnv(approach_cfg, E, device=device, seed=seed)
    if obs_stride != 56:
        env.set_obs_stride(obs_stride)
    a_res, hand = run_episodes(env, approach_policy, {"initial_q": cat["initial_q"], "goal_q": cat["goal_q"], "goal_pose6": cat["goal_pose6"],
                                                      "policy_mode": "approach"}, ready_cfg=r, handoff_confirm_steps=handoff_confirm_steps)
    env.close()
    final_ready = finisher_ready(a_res["final_position_error"], a_res["final_orientation_error"], a_res["final_action_magnitude"], a_res["final_dq_norm"], r)
    # handoff source: the final state if it is finisher-ready, else the first confirmed-ready snapshot (eval_workspace_expansion.py:138-139)
    has_hand = final_ready | hand["valid"]
    src = {}
    for k in ("state_q", "state_dq", "state_prev_action", "state_goal_q", "state_goal_pose6"):
        hk = hand.get(k, torch.zeros_like(a_res[k]))
        src[k] = torch.where(final_ready[:, None], a_res[k], hk)
    final = {k: a_res[k].clone() for k in ("final_position_error", "final_orientation_error", "final_action_magnitude", "final_dq_norm")}
    success = a_res["success"].clone()
    if finisher_policy is not None and finisher_cfg is not None and bool(has_hand.any()):
        fenv = ArmKinematicVecEnv(finisher_cfg, E, device=device, seed=seed)
        if obs_stride != 56:
            fenv.set_obs_stride(obs_stride)
        # rows without a handoff still need finite reset inputs; they are masked out of every result
        safe = {k: torch.where(has_hand[:, None], v, a_res[k]) for k, v in src.items()}
        f_res, _ = run_episodes(fenv, finisher_policy, _handoff_options(safe, "dock"), active=has_hand)
        fenv.close()
        for k in final:
            final[k] = torch.where(has_hand, f_res[k], final[k])
        success = torch.where(has_hand, f_res["success"], success)
    ready_hit = a_res["ready_hit"] | final_ready
    ready_dwell = (a_res["max_ready_streak"] >= handoff_confirm_steps) | final_ready
    pos_reg = a_res["final_position_error"] > a_res["min_position_error"] + 0.002
    ori_reg = a_res["final_orientation_error"] > a_res["min_orientation_error"] + 0.01

    def cpu(t):
        return t.detach().cpu().numpy()

    A = {k: cpu(v) for k, v in a_res.items() if v.ndim == 1}
    F = {k: cpu(v) for k, v in final.items()}
    succ, rh, rd, pr, orr = cpu(success), cpu(ready_hit), cpu(ready_dwell), cpu(pos_reg), cpu(ori_reg)
    rows: list[dict[str, Any]] = []
    stage_summaries: dict[int, dict[str, Any]] = {}
    for si, s in enumerate(stages):
        sl = slice(si * episodes, (si + 1) * episodes)
        metrics = []
        for e in range(sl.start, sl.stop):
            if succ[e]:
                reason = "success"
            elif A["final_position_error"][e] > r.finisher_ready_pos_threshold_m:
                reason = "position"
            elif A["final_orientation_error"][e] > r.finisher_ready_ori_threshold_rad:
                reason = "orientation"
            elif A["final_action_magnitude"][e] > r.finisher_ready_action_threshold:
                reason = "motion_action"
            elif A["final_dq_norm"][e] > r.finisher_ready_dq_threshold:
                reason = "motion_dq"
            elif not bool(A["max_ready_streak"][e] >= handoff_confirm_steps):
                reason = "dwell"
            else:
                reason = "timeout_or_regression"
            metrics.append({
                "episode_id": e - sl.start, "stage_index": int(s), "stage_name": approach_cfg.stage_names[s] if approach_cfg.stage_names else str(s),
                "success": bool(succ[e]), "finisher_ready_hit": bool(rh[e]), "finisher_ready_dwell": bool(rd[e]), "failure_reason": reason,
                "final_position_error": float(F["final_position_error"][e]), "final_orientation_error": float(F["final_orientation_error"][e]),
                "approach_final_position_error": float(A["final_position_error"][e]), "approach_final_orientation_error": float(A["final_orientation_error"][e]),
                "final_action_magnitude": float(F["final_action_magnitude"][e]), "final_dq_norm": float(F["final_dq_norm"][e]),
                "min_position_error": float(A["min_position_error"][e]), "min_orientation_error": float(A["min_orientation_error"][e]),
                "position_regression": bool(pr[e]), "orientation_regression": bool(orr[e]),
                "goal_position": cat["goal_pose6"][e][:3].tolist(), "goal_orientation": cat["goal_pose6"][e][3:].tolist(),
            })
        reasons: dict[str, int] = {}
        for mrow in metrics:
            reasons[mrow["failure_reason"]] = reasons.get(mrow["failure_reason"], 0) + 1
        stage_summaries[s] = {
            "episode_count": len(metrics), "success_rate": _mean([m["success"] for m in metrics]),
            "finisher_ready_hit_rate": _mean([m["finisher_ready_hit"] for m in metrics]), "dwell_success_rate": _mean([m["finisher_ready_dwell"] for m in metrics]),
            "mean_final_position_error": _mean([m["final_position_error"] for m in metrics]),
            "mean_final_orientation_error": _mean([m["final_orientation_error"] for m in metrics]),
            "mean_final_action_magnitude": _mean([m["final_action_magnitude"] for m in metrics]), "mean_final_dq_norm": _mean([m["final_dq_norm"] for m in metrics]),
            "regression_rate": _mean([m["position_regression"] or m["orientation_regression"] for m in metrics]), "failure_reason_counts": reasons,
        }
        rows.extend(metrics)
    gcfg = gate_config_from_dict(gate_config)
    score_stage = int(np.clip(int((gate_config or {}).get("score_stage_index", max(stages))), min(stages), max(stages)))
    selection = gated_score(stage_summaries, score_stage, gcfg)
    payload = {"episodes_per_stage": int(episodes), "seed": int(seed), "stage_metrics": {str(k): v for k, v in stage_summaries.items()},
               "best_model_selection": selection, "target_rows": rows}
    if artifact_root is not None:
        root = Path(artifact_root)
        root.mkdir(parents=True, exist_ok=True)
        (root / "stage_metrics.json").write_text(json.dumps(payload["stage_metrics"], indent=2))
        (root / "best_model_selection_summary.json").write_text(json.dumps(selection, indent=2))
        (root / "workspace_eval_summary.json").write_text(json.dumps(payload, indent=2))
    return payload
