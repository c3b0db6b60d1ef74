"""YAML / dict config -> ``kp1_config`` (the C-ABI struct of include/kp1.h).

Host-side mirror of the reference's config plumbing:

* ``deep_merge`` / ``load_yaml_file``      -> kinematic_phase1/training/policy_config.py:72-83
* ``load_overlay_with_bases``              -> kinematic_phase1/train_workspace_expansion.py:34-44
* ``load_workspace_expansion_config``      -> kinematic_phase1/train_workspace_expansion.py:47-51
* ``load_dock_config``                     -> kinematic_phase1/training/train_dock_policy.py:32-36
* ``to_env_config``                        -> kinematic_phase1/training/policy_config.py:96-164

The ctypes mirror of the struct is generated from the X-macro field lists in
``include/kp1.h`` so the header stays the single source of truth for layout and
defaults.  Like the reference's dataclasses (built with ``**yaml_dict``), unknown
keys in the reward / termination / observation / dock_reset blocks raise ``TypeError``.
"""
from __future__ import annotations

import ctypes as C
import json
import math
import re
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any

import yaml

NJ = 7
OBS_DIM = 56
MAX_STAGES = 16
MAX_MILESTONES = 8
KP1_UNSET = -2147483648

MODE_NAMES = {"approach": 0, "dock": 1}
MODE_FROM_INDEX = {v: k for k, v in MODE_NAMES.items()}

# Observation slices in SB3 CombinedExtractor (alphabetical key) order; include/kp1.h KP1_OBS_*.
OBS_LAYOUT: dict[str, tuple[int, int]] = {
    "dq": (0, 7),
    "goal_ori_err": (7, 3),
    "goal_pos_err": (10, 3),
    "joint_limit_margin": (13, 7),
    "mode_flag": (20, 4),
    "next_wp_ori_err": (24, 3),
    "next_wp_pos_err": (27, 3),
    "prev_action": (30, 7),
    "progress": (37, 3),
    "q": (40, 7),
    "task_type": (47, 3),
    "wp_ori_err": (50, 3),
    "wp_pos_err": (53, 3),
}
# Key order of the reference's observation dict (observation_builder.py:77-93).
OBS_KEYS_REFERENCE_ORDER = [
    "q", "dq", "prev_action", "goal_pos_err", "goal_ori_err", "wp_pos_err", "wp_ori_err",
    "next_wp_pos_err", "next_wp_ori_err", "task_type", "mode_flag", "progress", "joint_limit_margin",
]


def repo_root() -> Path:
    return Path(__file__).resolve().parent.parent


def header_path() -> Path:
    return repo_root() / "include" / "kp1.h"


def builtin_config_dir() -> Path:
    return Path(__file__).resolve().parent / "configs"


# ----------------------------------------------------------------------------- header parsing
_FIELD_RE = re.compile(r"X\(\s*(f64|i32)\s*,\s*(\w+)\s*,\s*([^)]+?)\s*\)")


def _parse_field_lists(text: str) -> dict[str, list[tuple[str, str, float | int]]]:
    lists: dict[str, list[tuple[str, str, float | int]]] = {}
    for m in re.finditer(r"#define\s+(KP1_\w+_FIELDS)\(X\)((?:[^\n]*\\\n)*[^\n]*)", text):
        fields = []
        for t, name, dflt in _FIELD_RE.findall(m.group(2)):
            if dflt == "KP1_UNSET":
                val: float | int = KP1_UNSET
            elif t == "i32":
                val = int(dflt)
            else:
                val = float(dflt)
            fields.append((t, name, val))
        lists[m.group(1)] = fields
    return lists


_FIELD_LISTS = _parse_field_lists(header_path().read_text())
_CT = {"f64": C.c_double, "i32": C.c_int32}


def _make_struct(name: str, list_name: str, extra: list[tuple[str, Any]] | None = None) -> type[C.Structure]:
    fields = [(n, _CT[t]) for t, n, _ in _FIELD_LISTS[list_name]]
    fields += extra or []
    cls = type(name, (C.Structure,), {"_fields_": fields, "_kp1_list": list_name})
    return cls


def _apply_defaults(obj: C.Structure) -> None:
    for _, n, d in _FIELD_LISTS[obj._kp1_list]:  # type: ignore[attr-defined]
        setattr(obj, n, d)


F7 = C.c_double * NJ
F6 = C.c_double * 6
FM = C.c_double * MAX_MILESTONES

EnvScalars = _make_struct("EnvScalars", "KP1_ENV_FIELDS")
ApproachReward = _make_struct(
    "ApproachReward", "KP1_APPROACH_REWARD_FIELDS",
    [("orientation_milestone_thresholds_rad", FM), ("orientation_milestone_bonuses", FM)],
)
DockReward = _make_struct("DockReward", "KP1_DOCK_REWARD_FIELDS")
Termination = _make_struct("Termination", "KP1_TERMINATION_FIELDS")
Observation = _make_struct("Observation", "KP1_OBSERVATION_FIELDS")
StageSampling = _make_struct("StageSampling", "KP1_STAGE_SAMPLING_FIELDS")
RandomStart = _make_struct(
    "RandomStart", "KP1_RANDOM_START_FIELDS",
    [("failure_recovery_q_noise", F7), ("initial_dq_noise", F7), ("initial_prev_action_noise", F7)],
)
DockReset = _make_struct(
    "DockReset", "KP1_DOCK_RESET_FIELDS",
    [("goal_q", F7), ("goal_noise", F7), ("init_q_noise", F7), ("close_init_q_noise", F7)],
)


class JointSpecs(C.Structure):
    _fields_ = [("lower", F7), ("upper", F7), ("delta_limit", F7)]


class Stage(C.Structure):
    _fields_ = [("start_q", F7), ("goal_q", F7), ("start_noise", F7), ("goal_noise", F7)]


class Kp1Config(C.Structure):
    _fields_ = [
        ("env", EnvScalars),
        ("curriculum_enabled", C.c_int32),
        ("n_stages", C.c_int32),
        ("joints", JointSpecs),
        ("stages", Stage * MAX_STAGES),
        ("stage_sampling", StageSampling),
        ("random_start", RandomStart),
        ("reward", ApproachReward),
        ("dock_reward", DockReward),
        ("dock_reset", DockReset),
        ("termination", Termination),
        ("observation", Observation),
    ]


class HandoffState(C.Structure):
    _fields_ = [("initial_q", F7), ("goal_q", F7), ("goal_pose6", F6), ("initial_dq", F7), ("initial_prev_action", F7)]


class RngState(C.Structure):
    _fields_ = [
        ("state_hi", C.c_uint64), ("state_lo", C.c_uint64), ("inc_hi", C.c_uint64), ("inc_lo", C.c_uint64),
        ("has_uint32", C.c_uint32), ("uinteger", C.c_uint32),
    ]


class ResetOpts(C.Structure):
    _fields_ = [
        ("initial_q", C.c_void_p), ("initial_dq", C.c_void_p), ("initial_prev_action", C.c_void_p),
        ("goal_q", C.c_void_p), ("goal_pose6", C.c_void_p), ("policy_mode", C.c_int32),
    ]


class InfoView(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "position_error_norm", "orientation_error_norm", "min_position_error", "executed_delta_q_l2", "action_l2",
        "delta_q_change_l2", "q", "dq", "prev_action", "goal_q", "goal_pose6", "ee_pose6", "entry_metrics",
        "episode_step", "dwell_count", "near_goal_entry_count", "near_goal_drift_count", "flags", "stage_index",
    )] + [("n_envs", C.c_int32), ("real_type", C.c_int32)]


# Default joint table; kinematics/joint_limits.py:37-47.  The reference overrides limits from a URDF only if
# external/.../install/...urdf exists (it does not in the repository), so the defaults are the effective values.
DEFAULT_JOINT_LIMITS = (0.385, math.pi, math.pi, math.pi, math.pi, math.pi, math.pi)
DEFAULT_DELTA_LIMITS = (0.08, 0.30, 0.24, 0.24, 0.30, 0.40, 0.30)
JOINT_ORDER = (
    "Rack_joint", "robot_base_joint", "shoulder1_joint", "shoulder2_joint", "wr1_joint", "wr2_joint", "wr3_joint",
)

# envs/curriculum.py:36-78 default_point_curriculum_stages
_DEFAULT_STAGES = [
    ("region_small", 0.0, None, (0.01, 0.03, 0.04, 0.03, 0.02, 0.02, 0.01)),
    ("region_medium", 0.0, None, (0.02, 0.06, 0.08, 0.06, 0.04, 0.04, 0.03)),
    ("region_medium_wide", 0.0, None, (0.03, 0.09, 0.12, 0.09, 0.06, 0.05, 0.04)),
    ("region_large", 0.01, None, (0.04, 0.12, 0.16, 0.12, 0.08, 0.06, 0.05)),
    ("region_large_offset", 0.02, (0.03, -0.04, 0.05, -0.03, 0.02, -0.01, 0.01), (0.05, 0.14, 0.18, 0.14, 0.09, 0.07, 0.06)),
    ("region_wide_local_random", 0.03, None, (0.06, 0.18, 0.22, 0.16, 0.10, 0.08, 0.07)),
]


def default_stage_dicts() -> list[dict[str, Any]]:
    out = []
    for name, sn, gq, gn in _DEFAULT_STAGES:
        out.append({
            "name": name,
            "start_q": [0.0] * NJ,
            "goal_q": list(gq) if gq else [0.0] * NJ,
            "start_noise": [0.0] + [sn] * 6,
            "goal_noise": list(gn),
        })
    return out


# ----------------------------------------------------------------------------- yaml plumbing
def load_yaml_file(path: str | Path) -> dict[str, Any]:
    return yaml.safe_load(Path(path).read_text()) or {}


def deep_merge(base: dict[str, Any], overlay: dict[str, Any]) -> dict[str, Any]:
    merged = dict(base)
    for key, value in overlay.items():
        if isinstance(value, dict) and isinstance(merged.get(key), dict):
            merged[key] = deep_merge(merged[key], value)
        else:
            merged[key] = value
    return merged


def load_overlay_with_bases(path: str | Path, search_dirs: tuple[Path, ...] = ()) -> dict[str, Any]:
    """Overlay YAML with an optional ``base_config:`` chain (resolved next to the file, then in
    ``search_dirs``, then in the built-in config dir)."""
    path = Path(path)
    overlay = load_yaml_file(path)
    base_config = overlay.pop("base_config", None)
    if not base_config:
        return overlay
    base_path = Path(str(base_config))
    if not base_path.is_absolute():
        for d in (path.parent, *search_dirs, builtin_config_dir()):
            if (d / base_path).exists():
                base_path = d / base_path
                break
        else:
            raise FileNotFoundError(f"base_config '{base_config}' referenced by {path} not found")
    return deep_merge(load_overlay_with_bases(base_path, search_dirs), overlay)


def load_workspace_expansion_config(explicit_path: str | Path | None) -> dict[str, Any]:
    """approach_default <- ppo_default <- overlay(+base chain); train_workspace_expansion.py:47-51."""
    cfg = deep_merge(load_yaml_file(builtin_config_dir() / "approach_default.yaml"),
                     load_yaml_file(builtin_config_dir() / "ppo_default.yaml"))
    if explicit_path:
        cfg = deep_merge(cfg, load_overlay_with_bases(explicit_path))
    return cfg


def load_dock_config(explicit_path: str | Path | None) -> dict[str, Any]:
    """dock_default <- ppo_default <- overlay; training/train_dock_policy.py:32-36."""
    cfg = deep_merge(load_yaml_file(builtin_config_dir() / "dock_default.yaml"),
                     load_yaml_file(builtin_config_dir() / "ppo_default.yaml"))
    if explicit_path:
        cfg = deep_merge(cfg, load_yaml_file(explicit_path))
    return cfg


# ----------------------------------------------------------------------------- dict -> struct
def _fill_scalars(obj: C.Structure, data: dict[str, Any], what: str, skip: tuple[str, ...] = ()) -> None:
    names = {n: t for t, n, _ in _FIELD_LISTS[obj._kp1_list]}  # type: ignore[attr-defined]
    for key, value in data.items():
        if key in skip:
            continue
        if key not in names or key.startswith("reserved") or key == "n_orientation_milestones":
            raise TypeError(f"{what}.__init__() got an unexpected keyword argument '{key}'")
        setattr(obj, key, int(bool(value)) if isinstance(value, bool) else (int(value) if names[key] == "i32" else float(value)))


def _vec7(values: Any, what: str) -> list[float]:
    data = [float(v) for v in values]
    if len(data) != NJ:
        raise ValueError(f"{what} requires 7-joint vectors")
    return data


@dataclass
class EnvConfig:
    """Resolved env config: the C struct plus the host-only pieces (names, handoff buffer)."""

    c: Kp1Config
    mode_name: str = "approach"
    stage_names: list[str] = field(default_factory=list)
    handoff_states: list[dict[str, list[float]]] = field(default_factory=list)
    source: dict[str, Any] = field(default_factory=dict)

    @property
    def n_stages(self) -> int:
        return int(self.c.n_stages)

    def handoff_array(self):
        arr = (HandoffState * max(len(self.handoff_states), 1))()
        for i, s in enumerate(self.handoff_states):
            arr[i].initial_q[:] = s["initial_q"]
            arr[i].goal_q[:] = s["goal_q"]
            arr[i].goal_pose6[:] = s["goal_pose6"]
            arr[i].initial_dq[:] = s["initial_dq"]
            arr[i].initial_prev_action[:] = s["initial_prev_action"]
        return arr

    def clone(self) -> "EnvConfig":
        c = Kp1Config()
        C.memmove(C.byref(c), C.byref(self.c), C.sizeof(Kp1Config))
        return EnvConfig(c=c, mode_name=self.mode_name, stage_names=list(self.stage_names),
                         handoff_states=list(self.handoff_states), source=self.source)


def default_config() -> Kp1Config:
    """Phase1EnvConfig() defaults; arm_kinematic_env.py:32-66."""
    c = Kp1Config()
    for blk in ("env", "reward", "dock_reward", "termination", "observation", "stage_sampling", "random_start", "dock_reset"):
        _apply_defaults(getattr(c, blk))
    for i in range(NJ):
        c.joints.lower[i] = -DEFAULT_JOINT_LIMITS[i]
        c.joints.upper[i] = DEFAULT_JOINT_LIMITS[i]
        c.joints.delta_limit[i] = DEFAULT_DELTA_LIMITS[i]
        c.random_start.failure_recovery_q_noise[i] = 0.04
    c.dock_reset.goal_noise[:] = (0.01, 0.03, 0.04, 0.03, 0.02, 0.02, 0.01)
    c.dock_reset.init_q_noise[:] = (0.01, 0.02, 0.03, 0.02, 0.015, 0.015, 0.01)
    c.dock_reset.close_init_q_noise[:] = (0.006, 0.012, 0.018, 0.012, 0.009, 0.009, 0.006)
    c.curriculum_enabled = 1
    _set_stages(c, default_stage_dicts())
    return c


def _set_stages(c: Kp1Config, stages: list[dict[str, Any]]) -> list[str]:
    if len(stages) > MAX_STAGES:
        raise ValueError(f"at most {MAX_STAGES} curriculum stages are supported")
    c.n_stages = len(stages)
    names = []
    for k, st in enumerate(stages):
        names.append(str(st["name"]))
        c.stages[k].start_q[:] = _vec7(st["start_q"], "Phase 1 curriculum stages")
        c.stages[k].goal_q[:] = _vec7(st["goal_q"], "Phase 1 curriculum stages")
        c.stages[k].start_noise[:] = _vec7(st.get("start_noise", [0.0] * NJ), "Phase 1 curriculum stages")
        c.stages[k].goal_noise[:] = _vec7(st.get("goal_noise", [0.0] * NJ), "Phase 1 curriculum stages")
    return names


def load_handoff_states(path_str: str, *, max_position_error_m: float, max_orientation_error_rad: float,
                        max_action_l2: float, base_dirs: tuple[Path, ...] = ()) -> list[dict[str, list[float]]]:
    """Filtered finisher handoff buffer; envs/reset_samplers.py:131-165."""
    if not path_str:
        return []
    path = Path(path_str)
    if not path.is_absolute() and not path.exists():
        for d in base_dirs:
            if (d / path).exists():
                path = d / path
                break
    if not path.exists():
        raise FileNotFoundError(f"Handoff state buffer does not exist: {path}")
    payload = json.loads(path.read_text())
    raw_states = payload.get("states", []) if isinstance(payload, dict) else payload
    states = []
    for item in raw_states:
        if float(item.get("position_error_norm", 0.0)) > max_position_error_m:
            continue
        if float(item.get("orientation_error_norm", 0.0)) > max_orientation_error_rad:
            continue
        if float(item.get("action_l2", 0.0)) > max_action_l2:
            continue
        g6 = [float(v) for v in item["goal_pose6"]]
        if len(g6) != 6:
            raise ValueError("Phase 1B reset samplers require 6D pose vectors")
        states.append({
            "initial_q": _vec7(item["initial_q"], "Phase 1B reset samplers"),
            "goal_q": _vec7(item["goal_q"], "Phase 1B reset samplers"),
            "goal_pose6": g6,
            "initial_dq": _vec7(item.get("initial_dq", [0.0] * NJ), "Phase 1B reset samplers"),
            "initial_prev_action": _vec7(item.get("initial_prev_action", [0.0] * NJ), "Phase 1B reset samplers"),
        })
    return states


_RS_INT_KEYS = {
    "home_stage_index", "old_success_max_stage_index", "frontier_min_stage_index", "frontier_max_stage_index",
    "known_target_max_stage_index", "frontier_target_min_stage_index", "frontier_target_max_stage_index",
    "stress_target_min_stage_index", "stress_target_max_stage_index", "mixed_target_max_stage_index",
}
_RS_FLOAT_KEYS = {
    "home_start_ratio", "old_successful_start_ratio", "random_valid_q_start_ratio", "frontier_pair_ratio",
    "failure_recovery_start_ratio", "stress_start_ratio", "min_pair_joint_l2",
}


def to_env_config(config: dict[str, Any], *, handoff_base_dirs: tuple[Path, ...] = ()) -> EnvConfig:
    """Mirror of policy_config.to_env_config (policy_config.py:96-164) for modes approach / dock."""
    env_cfg = config.get("env", {}) or {}
    c = default_config()
    mode_name = str(env_cfg.get("mode", "approach"))
    if mode_name not in MODE_NAMES:
        raise ValueError(f"Unsupported policy mode '{mode_name}' (this engine builds 'approach' and 'dock')")
    if int(env_cfg.get("n_joints", 7)) != NJ:
        raise ValueError("joint_specs length must match n_joints")
    e = c.env
    e.mode = MODE_NAMES[mode_name]
    e.episode_length = int(env_cfg.get("episode_length", 75))
    termination_cfg = env_cfg.get("termination", {}) or {}
    e.dwell_steps_target = int(termination_cfg.get("success_dwell_steps", 3))  # policy_config.py:146 quirk
    e.dynamic_action_delta_scale_enabled = int(bool(env_cfg.get("dynamic_action_delta_scale_enabled", False)))
    for key, dflt in (
        ("goal_sample_margin_fraction", 0.10), ("start_sample_margin_fraction", 0.20), ("action_delta_scale", 1.0),
        ("dynamic_action_delta_scale_near_pos_threshold_m", 0.0), ("dynamic_action_delta_scale_far_pos_threshold_m", 0.0),
        ("dynamic_action_delta_scale_near_multiplier", 1.0), ("dynamic_action_delta_scale_far_multiplier", 1.0),
        ("dock_action_delta_scale", 0.0), ("dock_residual_action_limit", 1.0), ("dock_delta_q_change_limit_scale", 0.0),
        ("dock_dynamic_action_limit_near_pos_threshold_m", 0.0), ("dock_dynamic_action_limit_far_pos_threshold_m", 0.0),
    ):
        setattr(e, key, float(env_cfg.get(key, dflt)))
    ral = env_cfg.get("dock_residual_action_limit", 1.0)
    dqc = env_cfg.get("dock_delta_q_change_limit_scale", 0.0)
    e.dock_dynamic_residual_action_limit_near = float(env_cfg.get("dock_dynamic_residual_action_limit_near", ral))
    e.dock_dynamic_residual_action_limit_far = float(env_cfg.get("dock_dynamic_residual_action_limit_far", ral))
    e.dock_dynamic_delta_q_change_limit_scale_near = float(env_cfg.get("dock_dynamic_delta_q_change_limit_scale_near", dqc))
    e.dock_dynamic_delta_q_change_limit_scale_far = float(env_cfg.get("dock_dynamic_delta_q_change_limit_scale_far", dqc))

    # curriculum
    cur = env_cfg.get("curriculum", {}) or {}
    c.curriculum_enabled = int(bool(cur.get("enabled", True)))
    stage_dicts = cur.get("stages")
    stage_names = _set_stages(c, list(stage_dicts) if stage_dicts else default_stage_dicts())

    # workspace_stage_sampling (+ nested random_start_pair_sampling)
    wss = dict(env_cfg.get("workspace_stage_sampling", {}) or {})
    ss = c.stage_sampling
    ss.enabled = int(bool(wss.get("enabled", False)))
    ss.current_stage_ratio = float(wss.get("current_stage_ratio", 0.50))
    ss.previous_stage_ratio = float(wss.get("previous_stage_ratio", 0.25))
    ss.old_workspace_replay_ratio = float(wss.get("old_workspace_replay_ratio", 0.20))
    ss.failure_replay_ratio = float(wss.get("failure_replay_ratio", 0.05))
    ss.previous_stage_min_index = int(wss.get("previous_stage_min_index", 0))
    if "old_workspace_max_stage_index" in wss:
        ss.old_workspace_max_stage_index = int(wss["old_workspace_max_stage_index"])
    rs_cfg = dict(wss.get("random_start_pair_sampling", {}) or {})
    rs = c.random_start
    rs.enabled = int(bool(rs_cfg.get("enabled", False)))
    for key in _RS_INT_KEYS:
        if key in rs_cfg:
            setattr(rs, key, int(rs_cfg[key]))
    for key in _RS_FLOAT_KEYS:
        if key in rs_cfg:
            setattr(rs, key, float(rs_cfg[key]))
    for key in ("stress_start_margin_fraction", "random_valid_start_margin_fraction"):
        if key in rs_cfg:
            setattr(rs, key, float(rs_cfg[key]))
            setattr(rs, "has_" + key, 1)
    for key in ("failure_recovery_q_noise", "initial_dq_noise", "initial_prev_action_noise"):
        if key in rs_cfg:
            getattr(rs, key)[:] = _vec7(rs_cfg[key], key)

    # rewards
    reward_cfg = dict(env_cfg.get("reward", {}) or {})
    thr = tuple(reward_cfg.pop("orientation_milestone_thresholds_rad", ()))
    bon = tuple(reward_cfg.pop("orientation_milestone_bonuses", ()))
    n_ms = min(len(thr), len(bon))  # zip(..., strict=False); reward_approach.py:111
    if n_ms > MAX_MILESTONES:
        raise ValueError(f"at most {MAX_MILESTONES} orientation milestones are supported")
    _fill_scalars(c.reward, reward_cfg, "ApproachRewardConfig", skip=())
    c.reward.n_orientation_milestones = n_ms
    for i in range(n_ms):
        c.reward.orientation_milestone_thresholds_rad[i] = float(thr[i])
        c.reward.orientation_milestone_bonuses[i] = float(bon[i])
    _fill_scalars(c.dock_reward, dict(env_cfg.get("dock_reward", {}) or {}), "DockRewardConfig")
    _fill_scalars(c.termination, dict(termination_cfg), "TerminationConfig")
    _fill_scalars(c.observation, dict(env_cfg.get("observation", {}) or {}), "ObservationBuilderConfig")

    # dock reset (+ handoff buffer, host side)
    dr_cfg = dict(env_cfg.get("dock_reset", {}) or {})
    handoff_states: list[dict[str, list[float]]] = []
    if dr_cfg:
        vec_keys = ("goal_q", "goal_noise", "init_q_noise", "close_init_q_noise")
        host_keys = ("handoff_state_buffer_path", "handoff_state_max_position_error_m",
                     "handoff_state_max_orientation_error_rad", "handoff_state_max_action_l2")
        for key in vec_keys:
            if key in dr_cfg:
                getattr(c.dock_reset, key)[:] = _vec7(dr_cfg[key], "Phase 1B reset samplers")
        _fill_scalars(c.dock_reset, dr_cfg, "DockResetConfig", skip=vec_keys + host_keys)
        handoff_states = load_handoff_states(
            str(dr_cfg.get("handoff_state_buffer_path", "") or ""),
            max_position_error_m=float(dr_cfg.get("handoff_state_max_position_error_m", 1.0)),
            max_orientation_error_rad=float(dr_cfg.get("handoff_state_max_orientation_error_rad", 10.0)),
            max_action_l2=float(dr_cfg.get("handoff_state_max_action_l2", 10.0)),
            base_dirs=handoff_base_dirs,
        )
    rr = env_cfg.get("route_reset", {}) or {}
    if bool(rr.get("enabled", False)):
        raise NotImplementedError("route_reset sampling is handled by the route layer, not the base env")
    return EnvConfig(c=c, mode_name=mode_name, stage_names=stage_names, handoff_states=handoff_states, source=config)


def to_algorithm_kwargs(config: dict[str, Any], algorithm: str = "ppo") -> dict[str, Any]:
    """policy_config.py:176-177"""
    return dict(config.get("algorithms", {}).get(algorithm, {}))
