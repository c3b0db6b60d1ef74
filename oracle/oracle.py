"""ctypes wrapper of the CPU oracle (oracle/libkp1_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
product package never does.  The wrapper presents the reference's single-env API
(``reset(seed=, options=)`` / ``step(action)``) so the parity tests read like the reference's.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path
from typing import Any

import numpy as np

from rl_brain_trainer_amd import config as kcfg

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / "libkp1_oracle.so"
MAX_COMPONENTS = 64


def build(force: bool = False) -> Path:
    if force or not LIB_PATH.exists() or LIB_PATH.stat().st_mtime < max(
        (HERE / "kp1_oracle.c").stat().st_mtime, (HERE / "kp1_oracle.h").stat().st_mtime,
        (HERE / "kp1_route_oracle.c").stat().st_mtime, (HERE / "kp1_route_oracle.h").stat().st_mtime,
        (HERE.parent / "include" / "kp1.h").stat().st_mtime, (HERE.parent / "include" / "kp1_route.h").stat().st_mtime,
    ):
        subprocess.run(["make", "-C", str(HERE)], check=True, capture_output=True)
    return LIB_PATH


class _U128(C.Structure):
    _fields_ = [("lo", C.c_uint64), ("hi", C.c_uint64)]


class ORng(C.Structure):
    # unsigned __int128 is 16-byte aligned in C: pad the tail so sizeof == 48
    _fields_ = [("state", _U128), ("inc", _U128), ("has_uint32", C.c_int), ("uinteger", C.c_uint32), ("_pad", C.c_char * 8)]


class OEnv(C.Structure):
    _fields_ = [
        ("cfg", kcfg.Kp1Config),
        ("_pad0", C.c_char * ((16 - C.sizeof(kcfg.Kp1Config) % 16) % 16)),
        ("rng", ORng),
        ("handoff", C.c_void_p),
        ("n_handoff", C.c_int),
        ("episode_step", C.c_int), ("dwell_count", C.c_int), ("near_goal_entry_count", C.c_int),
        ("near_goal_drift_count", C.c_int), ("pre_near_goal_hit", C.c_int), ("near_goal_hit", C.c_int),
        ("min_pos_error", C.c_double),
        ("q", kcfg.F7), ("dq", kcfg.F7), ("prev_action", kcfg.F7),
        ("entry_position_error_norm", C.c_double), ("entry_orientation_error_norm", C.c_double),
        ("entry_action_l2", C.c_double), ("entry_dq_norm", C.c_double),
        ("goal_q", kcfg.F7), ("goal_pose6", kcfg.F6), ("ee_pose6", kcfg.F6),
        ("curriculum_stage_index", C.c_int), ("policy_mode", C.c_int), ("last_reset_stage", C.c_int),
    ]


class OResetOpts(C.Structure):
    _fields_ = [
        ("initial_q", C.POINTER(C.c_double)), ("initial_dq", C.POINTER(C.c_double)),
        ("initial_prev_action", C.POINTER(C.c_double)), ("goal_q", C.POINTER(C.c_double)),
        ("goal_pose6", C.POINTER(C.c_double)), ("policy_mode", C.c_int),
    ]


class OStepOut(C.Structure):
    _fields_ = [
        ("reward", C.c_double),
        ("terminated", C.c_int), ("truncated", C.c_int), ("success", C.c_int), ("invalid", C.c_int),
        ("position_error_norm", C.c_double), ("orientation_error_norm", C.c_double),
        ("executed_delta_q_l2", C.c_double), ("action_l2", C.c_double), ("delta_q_change_l2", C.c_double),
        ("dock_action_limit", C.c_double), ("dock_delta_q_change_limit_scale", C.c_double),
        ("joint_limit_margin_min", C.c_double),
        ("n_components", C.c_int),
        ("components", C.c_double * MAX_COMPONENTS),
    ]


class OTracker(C.Structure):
    _fields_ = [
        ("threshold", C.c_double), ("window", C.c_int), ("min_episodes", C.c_int), ("max_stage_index", C.c_int),
        ("stage_index", C.c_int), ("stage_episode_count", C.c_int), ("ring", C.c_int * 1024),
        ("ring_len", C.c_int), ("ring_head", C.c_int), ("last_trigger_rate", C.c_double),
    ]


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(LIB_PATH))
        dp = C.POINTER(C.c_double)
        L.kp1o_fk_pose6.argtypes = [dp, dp]
        L.kp1o_pose_error.argtypes = [dp, dp, dp, dp]
        L.kp1o_wrap_to_pi.argtypes = [C.c_double]
        L.kp1o_wrap_to_pi.restype = C.c_double
        for fn in ("kp1o_clip_q", "kp1o_joint_limit_margin", "kp1o_normalize_q", "kp1o_normalize_dq"):
            getattr(L, fn).argtypes = [C.POINTER(kcfg.JointSpecs), dp, dp]
        L.kp1o_rng_seed.argtypes = [C.POINTER(ORng), C.c_uint64]
        L.kp1o_rng_next64.argtypes = [C.POINTER(ORng)]
        L.kp1o_rng_next64.restype = C.c_uint64
        L.kp1o_rng_next32.argtypes = [C.POINTER(ORng)]
        L.kp1o_rng_next32.restype = C.c_uint32
        L.kp1o_rng_double.argtypes = [C.POINTER(ORng)]
        L.kp1o_rng_double.restype = C.c_double
        L.kp1o_rng_integers.argtypes = [C.POINTER(ORng), C.c_int64, C.c_int64]
        L.kp1o_rng_integers.restype = C.c_int64
        L.kp1o_rng_get.argtypes = [C.POINTER(ORng), C.POINTER(kcfg.RngState)]
        L.kp1o_rng_set.argtypes = [C.POINTER(ORng), C.POINTER(kcfg.RngState)]
        L.kp1o_config_default.argtypes = [C.POINTER(kcfg.Kp1Config)]
        L.kp1o_env_init.argtypes = [C.POINTER(OEnv), C.POINTER(kcfg.Kp1Config)]
        L.kp1o_env_set_handoff.argtypes = [C.POINTER(OEnv), C.c_void_p, C.c_int]
        L.kp1o_env_seed.argtypes = [C.POINTER(OEnv), C.c_uint64]
        L.kp1o_env_set_stage.argtypes = [C.POINTER(OEnv), C.c_int]
        L.kp1o_env_reset.argtypes = [C.POINTER(OEnv), C.POINTER(OResetOpts), C.POINTER(C.c_float)]
        L.kp1o_env_step.argtypes = [C.POINTER(OEnv), dp, C.POINTER(C.c_float), C.POINTER(OStepOut)]
        L.kp1o_env_observe.argtypes = [C.POINTER(OEnv), C.POINTER(C.c_float)]
        L.kp1o_env_capture_entry_metrics.argtypes = [C.POINTER(OEnv)]
        L.kp1o_component_name.argtypes = [C.c_int, C.c_int]
        L.kp1o_component_name.restype = C.c_char_p
        L.kp1o_num_components.argtypes = [C.c_int]
        L.kp1o_tracker_init.argtypes = [C.POINTER(OTracker), C.c_double, C.c_int, C.c_int, C.c_int, C.c_int]
        L.kp1o_tracker_record.argtypes = [C.POINTER(OTracker), C.c_int]
        L.kp1o_max_threads.restype = C.c_int
        L.kp1o_sizeof_env.restype = C.c_size_t
        L.kp1o_offsetof_env.argtypes = [C.c_int]
        L.kp1o_offsetof_env.restype = C.c_size_t
        assert L.kp1o_sizeof_env() == C.sizeof(OEnv), (L.kp1o_sizeof_env(), C.sizeof(OEnv))
        for k, name in enumerate(("rng", "handoff", "min_pos_error", "goal_pose6", "last_reset_stage")):
            assert L.kp1o_offsetof_env(k) == getattr(OEnv, name).offset, name
        L.kp1o_batch_step.argtypes = [C.POINTER(OEnv), C.c_int, dp, C.POINTER(C.c_float), dp, C.POINTER(C.c_uint8), C.c_int, C.c_int]
        L.kp1o_batch_step_components.argtypes = [C.POINTER(OEnv), C.c_int, dp, C.POINTER(C.c_float), dp, C.POINTER(C.c_uint8), C.c_int, C.c_int, dp]
        _lib = L
    return _lib


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def fk_pose6(q: np.ndarray) -> np.ndarray:
    q = np.ascontiguousarray(np.atleast_2d(q), dtype=np.float64)
    out = np.empty((q.shape[0], 6))
    L = lib()
    for i in range(q.shape[0]):
        L.kp1o_fk_pose6(_dp(q[i]), _dp(out[i]))
    return out


def pose_error(curr: np.ndarray, goal: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    curr = np.ascontiguousarray(curr, dtype=np.float64)
    goal = np.ascontiguousarray(goal, dtype=np.float64)
    pe, oe = np.empty(3), np.empty(3)
    lib().kp1o_pose_error(_dp(curr), _dp(goal), _dp(pe), _dp(oe))
    return pe, oe


def rng_words(r: ORng) -> np.ndarray:
    st = kcfg.RngState()
    lib().kp1o_rng_get(C.byref(r), C.byref(st))
    return np.array([st.state_hi, st.state_lo, st.inc_hi, st.inc_lo, st.has_uint32, st.uinteger], dtype=np.uint64)


def component_names(mode: int) -> list[str]:
    L = lib()
    return [L.kp1o_component_name(mode, i).decode() for i in range(L.kp1o_num_components(mode))]


class OracleEnv:
    """Single env with the reference's API surface (arm_kinematic_env.py:69-560)."""

    def __init__(self, cfg: kcfg.EnvConfig) -> None:
        self.cfg = cfg
        self.L = lib()
        self.e = OEnv()
        self.L.kp1o_env_init(C.byref(self.e), C.byref(cfg.c))
        self._handoff = cfg.handoff_array()
        self.L.kp1o_env_set_handoff(C.byref(self.e), C.cast(self._handoff, C.c_void_p), len(cfg.handoff_states))
        self._obs = np.zeros(kcfg.OBS_DIM, dtype=np.float32)
        self._out = OStepOut()

    def _obs_ptr(self):
        return self._obs.ctypes.data_as(C.POINTER(C.c_float))

    def set_curriculum_stage(self, stage: int) -> None:
        self.L.kp1o_env_set_stage(C.byref(self.e), int(stage))

    def set_policy_mode(self, mode: str) -> None:
        if mode not in kcfg.MODE_NAMES:
            raise ValueError(f"Unsupported policy mode '{mode}'")
        self.e.policy_mode = kcfg.MODE_NAMES[mode]

    def rng_words(self) -> np.ndarray:
        return rng_words(self.e.rng)

    def reset(self, *, seed: int | None = None, options: dict[str, Any] | None = None) -> np.ndarray:
        if seed is not None:
            self.L.kp1o_env_seed(C.byref(self.e), int(seed))
        opts = None
        keep = []
        if options:
            o = OResetOpts()
            o.policy_mode = -1
            for key in ("initial_q", "initial_dq", "initial_prev_action", "goal_q", "goal_pose6"):
                if options.get(key) is not None:
                    arr = np.ascontiguousarray(options[key], dtype=np.float64)
                    keep.append(arr)
                    setattr(o, key, _dp(arr))
            if options.get("policy_mode") is not None:
                o.policy_mode = kcfg.MODE_NAMES[options["policy_mode"]]
            opts = C.byref(o)
        self.L.kp1o_env_reset(C.byref(self.e), opts, self._obs_ptr())
        return self._obs.copy()

    def step(self, action: np.ndarray) -> tuple[np.ndarray, dict[str, Any]]:
        a = np.ascontiguousarray(action, dtype=np.float64)
        if a.shape != (kcfg.NJ,):
            raise ValueError(f"Expected action shape {(kcfg.NJ,)}, got {a.shape}")
        self.L.kp1o_env_step(C.byref(self.e), _dp(a), self._obs_ptr(), C.byref(self._out))
        o = self._out
        e = self.e
        info = {
            "reward": o.reward, "terminated": bool(o.terminated), "truncated": bool(o.truncated), "success": bool(o.success),
            "position_error_norm": o.position_error_norm, "orientation_error_norm": o.orientation_error_norm,
            "executed_delta_q_l2": o.executed_delta_q_l2, "action_l2": o.action_l2, "delta_q_change_l2": o.delta_q_change_l2,
            "dock_action_limit": o.dock_action_limit, "dock_delta_q_change_limit_scale": o.dock_delta_q_change_limit_scale,
            "joint_limit_margin_min": o.joint_limit_margin_min,
            "components": np.array(o.components[: o.n_components]),
            "dwell_count": e.dwell_count, "near_goal_entry_count": e.near_goal_entry_count,
            "near_goal_drift_count": e.near_goal_drift_count, "pre_near_goal_hit": bool(e.pre_near_goal_hit),
            "near_goal_hit": bool(e.near_goal_hit), "min_position_error": e.min_pos_error,
            "q": np.array(e.q[:]), "dq": np.array(e.dq[:]), "ee_pose6": np.array(e.ee_pose6[:]), "step_count": e.episode_step,
        }
        return self._obs.copy(), info

    def state(self) -> dict[str, np.ndarray]:
        e = self.e
        return {
            "q": np.array(e.q[:]), "dq": np.array(e.dq[:]), "prev_action": np.array(e.prev_action[:]),
            "goal_q": np.array(e.goal_q[:]), "goal_pose6": np.array(e.goal_pose6[:]), "ee_pose6": np.array(e.ee_pose6[:]),
            "entry_metrics": np.array([e.entry_position_error_norm, e.entry_orientation_error_norm, e.entry_action_l2, e.entry_dq_norm]),
        }


class OracleVecEnv:
    """N oracle envs stepped together (OpenMP) -- the CPU baseline of bench.py and the batched parity checker."""

    def __init__(self, cfg: kcfg.EnvConfig, n_envs: int, seed0: int, first_env_id: int = 0, stage: int = 0) -> None:
        self.L = lib()
        self.n = n_envs
        self.cfg = cfg
        self.envs = (OEnv * n_envs)()
        self._handoff = cfg.handoff_array()
        for i in range(n_envs):
            self.L.kp1o_env_init(C.byref(self.envs[i]), C.byref(cfg.c))
            self.L.kp1o_env_set_handoff(C.byref(self.envs[i]), C.cast(self._handoff, C.c_void_p), len(cfg.handoff_states))
            self.L.kp1o_env_seed(C.byref(self.envs[i]), seed0 + first_env_id + i)
            self.L.kp1o_env_set_stage(C.byref(self.envs[i]), stage)
        self.obs = np.zeros((n_envs, kcfg.OBS_DIM), dtype=np.float32)
        self.reward = np.zeros(n_envs)
        self.done = np.zeros(n_envs, dtype=np.uint8)

    def set_stage(self, stage: int) -> None:
        for i in range(self.n):
            self.L.kp1o_env_set_stage(C.byref(self.envs[i]), stage)

    def reset(self) -> np.ndarray:
        for i in range(self.n):
            self.L.kp1o_env_reset(C.byref(self.envs[i]), None, self.obs[i].ctypes.data_as(C.POINTER(C.c_float)))
        return self.obs

    def step(self, actions: np.ndarray, auto_reset: bool = True, n_threads: int = 0):
        a = np.ascontiguousarray(actions, dtype=np.float64)
        assert a.shape == (self.n, kcfg.NJ)
        if getattr(self, "_components", None) is None:
            self._components = np.zeros((self.n, MAX_COMPONENTS))
        self.L.kp1o_batch_step_components(self.envs, self.n, _dp(a), self.obs.ctypes.data_as(C.POINTER(C.c_float)), _dp(self.reward),
                                          self.done.ctypes.data_as(C.POINTER(C.c_uint8)), int(auto_reset), n_threads, _dp(self._components))
        return self.obs, self.reward, self.done

    def components(self) -> np.ndarray:
        """reward components of the last step, [n_envs, n_components of this mode]"""
        mode = int(self.cfg.c.env.mode) if hasattr(self.cfg.c.env, "mode") else 0
        return self._components[:, :self.L.kp1o_num_components(mode)]

    def field(self, name: str) -> np.ndarray:
        return np.array([np.array(getattr(self.envs[i], name)) for i in range(self.n)])
