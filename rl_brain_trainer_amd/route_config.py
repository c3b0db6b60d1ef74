"""Route-curriculum configuration: ctypes mirror of include/kp1_route.h and the YAML -> struct mapping.

Mirrors the reference's config plumbing for the route trainers (kinematic_phase1/train_route_curriculum.py:45-57, 86-101):
``route.reward`` -> RouteRewardConfig(**...), ``route.reset`` -> RouteResetSamplerConfig(**..., max_route_index defaulted to the
first prefix), ``route.observation`` -> RouteObservationConfig, ``route.sequence`` -> RouteSequenceConfig; unknown keys raise
TypeError like the dataclass constructors do.  ``load_route_q`` reads the dense route JSON (route_dataset.py:43-50, 75-79).
"""
from __future__ import annotations

import ctypes as C
import json
import re
from pathlib import Path
from typing import Any

import numpy as np

from . import config as kcfg

_HEADER = (kcfg.repo_root() / "include" / "kp1_route.h").read_text()
_m = re.search(r"#define KP1_ROUTE_REWARD_FIELDS\(X\)(.*?)\ntypedef", _HEADER, re.S)
REWARD_FIELDS: list[tuple[str, float]] = [(n, float(d)) for n, d in re.findall(r"X\((\w+),\s*([-0-9.e]+)\)", _m.group(1))]
assert len(REWARD_FIELDS) == 19, REWARD_FIELDS

RESET_MODES = {"prefix_start_reset": 1, "random_prefix_reset": 2, "segment_reset": 3, "replay_reset": 4, "recovery_reset": 5}
MODE_NAMES = ["prefix_start", "random_prefix", "segment", "replay", "recovery", "explicit"]
ROUTE_OBS_DIM = 80
COMPONENT_NAMES = ["q_goal_progress", "ee_position_progress", "ee_orientation_progress", "route_tangent_progress_bonus", "same_step_route_ready_bonus",
                   "route_ready_dwell_bonus", "low_motion_near_waypoint_bonus", "orientation_regression_penalty", "q_route_regression_penalty",
                   "off_route_penalty", "action_smoothness_penalty", "dq_penalty", "no_progress_penalty", "curr_q_error", "curr_pos_error",
                   "curr_ori_error", "route_ready"]
# the 17 Dict keys in SB3's sorted order -> (offset, width) in the 80-float row
ROUTE_OBS_LAYOUT: dict[str, tuple[int, int]] = {**{k: v for k, v in kcfg.OBS_LAYOUT.items() if v[0] < 47},
                                                "route_q_error": (47, 7), "route_q_goal": (54, 7), "route_scalar": (61, 3), "route_tangent": (64, 7),
                                                "task_type": (71, 3), "wp_ori_err": (74, 3), "wp_pos_err": (77, 3)}


class RouteReward(C.Structure):
    _fields_ = [(n, C.c_double) for n, _ in REWARD_FIELDS]


class RouteResetCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("mode", "min_route_index", "max_route_index", "segment_start_index", "segment_end_index", "replay_start_index",
                                         "replay_end_index", "pad_")] + \
               [(n, C.c_double) for n in ("prefix_start_reset_ratio", "random_prefix_reset_ratio", "segment_reset_ratio", "replay_reset_ratio",
                                          "recovery_reset_ratio", "q_noise_std", "dq_noise_std", "prev_action_noise_std")]


class RouteConfig(C.Structure):
    _fields_ = [("reward", RouteReward), ("reset", RouteResetCfg), ("include_route_keys", C.c_int32), ("sequence_enabled", C.c_int32),
                ("sequence_length", C.c_int32), ("reset_ready_streak_on_advance", C.c_int32)]


_RESET_DEFAULTS: dict[str, Any] = {  # route_reset_samplers.py:14-30
    "mode": "mixed_prefix_segment", "min_route_index": 1, "max_route_index": 20, "segment_start_index": 1, "segment_end_index": 40,
    "replay_start_index": 1, "replay_end_index": 120, "prefix_start_reset_ratio": 0.10, "random_prefix_reset_ratio": 0.55, "segment_reset_ratio": 0.20,
    "replay_reset_ratio": 0.0, "recovery_reset_ratio": 0.15, "q_noise_std": 0.002, "dq_noise_std": 0.0005, "prev_action_noise_std": 0.02,
}


def default_route_config() -> RouteConfig:
    return route_config_from_dict({})


def route_config_from_dict(cfg: dict[str, Any], *, max_route_index: int | None = None) -> RouteConfig:
    """cfg = the merged YAML dict (its ``route`` block is read).  max_route_index = the setdefault of _route_reset_config."""
    route = cfg.get("route", {}) or {}
    out = RouteConfig()
    reward = dict(route.get("reward", {}) or {})
    known = {n for n, _ in REWARD_FIELDS}
    for k in reward:
        if k not in known:
            raise TypeError(f"RouteRewardConfig.__init__() got an unexpected keyword argument '{k}'")
    for n, d in REWARD_FIELDS:
        setattr(out.reward, n, float(reward.get(n, d)))
    reset = dict(route.get("reset", {}) or {})
    if max_route_index is not None:
        reset.setdefault("max_route_index", int(max_route_index))
    for k in reset:
        if k not in _RESET_DEFAULTS:
            raise TypeError(f"RouteResetSamplerConfig.__init__() got an unexpected keyword argument '{k}'")
    merged = {**_RESET_DEFAULTS, **reset}
    out.reset.mode = RESET_MODES.get(str(merged["mode"]), 0)
    for n in ("min_route_index", "max_route_index", "segment_start_index", "segment_end_index", "replay_start_index", "replay_end_index"):
        setattr(out.reset, n, int(merged[n]))
    for n in ("prefix_start_reset_ratio", "random_prefix_reset_ratio", "segment_reset_ratio", "replay_reset_ratio", "recovery_reset_ratio", "q_noise_std",
              "dq_noise_std", "prev_action_noise_std"):
        setattr(out.reset, n, float(merged[n]))
    obs = dict(route.get("observation", {}) or {})
    for k in obs:
        if k != "include_route_keys":
            raise TypeError(f"RouteObservationConfig.__init__() got an unexpected keyword argument '{k}'")
    out.include_route_keys = int(bool(obs.get("include_route_keys", False)))
    seq = dict(route.get("sequence", {}) or {})
    for k in seq:
        if k not in ("enabled", "sequence_length", "reset_ready_streak_on_advance"):
            raise TypeError(f"RouteSequenceConfig.__init__() got an unexpected keyword argument '{k}'")
    out.sequence_enabled = int(bool(seq.get("enabled", False)))
    out.sequence_length = int(seq.get("sequence_length", 5))
    out.reset_ready_streak_on_advance = int(bool(seq.get("reset_ready_streak_on_advance", True)))
    return out


def load_route_q(path: str | Path) -> np.ndarray:
    """Dense q-goal route as [W, 7] float64 ({"route_q": [q7 | {"q": q7} | {"q_goal": q7}, ...]} or a bare list)."""
    payload = json.loads(Path(path).read_text(encoding="utf-8"))
    entries = payload.get("route_q") if isinstance(payload, dict) else payload
    if not isinstance(entries, list) or not entries:
        raise ValueError(f"Route dataset must contain a non-empty list: {path}")

    def q_of(entry):
        if isinstance(entry, dict):
            if "q" in entry:
                return entry["q"]
            if "q_goal" in entry:
                return entry["q_goal"]
        return entry

    return np.asarray([q_of(e) for e in entries], dtype=float)


def prefix_stages(cfg: dict[str, Any], n_waypoints: int) -> list[int]:
    """train_route_curriculum.py:86-88"""
    route = cfg.get("route", {}) or {}
    stages = [int(x) for x in (route.get("curriculum", {}) or {}).get("prefix_stages", [20, 40, 80, 120, 180, 260, 360, n_waypoints - 1])]
    return [min(max(1, p), n_waypoints - 1) for p in stages]
