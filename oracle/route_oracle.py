"""ctypes wrapper of the route-curriculum CPU oracle (oracle/kp1_route_oracle.c).  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
from typing import Any

import numpy as np

from rl_brain_trainer_amd import config as kcfg
from rl_brain_trainer_amd import route_config as rcfg

from . import oracle as orc

N_COMP = 17


class RouteSample(C.Structure):
    _fields_ = [("initial_q", kcfg.F7), ("initial_dq", kcfg.F7), ("initial_prev_action", kcfg.F7), ("goal_q", kcfg.F7), ("route_index", C.c_int),
                ("start_index", C.c_int), ("mode", C.c_int)]


class RouteStepOut(C.Structure):
    _fields_ = [("reward", C.c_double)] + [(n, C.c_int) for n in ("terminated", "truncated", "success", "route_ready", "ready_streak", "waypoint_success",
                                                                     "route_regression", "orientation_hit", "route_index", "completed_waypoints")] + \
               [("q_error_norm", C.c_double), ("nearest_route_q_distance", C.c_double), ("components", C.c_double * N_COMP)]


_bound = False


def lib() -> C.CDLL:
    global _bound
    L = orc.lib()
    if not _bound:
        dp = C.POINTER(C.c_double)
        L.kp1o_route_load.argtypes = [dp, C.c_int]
        L.kp1o_route_load.restype = C.c_void_p
        L.kp1o_route_free.argtypes = [C.c_void_p]
        L.kp1o_rng_standard_normal.argtypes = [C.POINTER(orc.ORng)]
        L.kp1o_rng_standard_normal.restype = C.c_double
        L.kp1o_rng_choice_p.argtypes = [C.POINTER(orc.ORng), dp, C.c_int]
        L.kp1o_route_sample_reset.argtypes = [C.POINTER(orc.ORng), C.c_void_p, C.POINTER(kcfg.JointSpecs), C.POINTER(rcfg.RouteResetCfg), C.POINTER(RouteSample)]
        L.kp1o_route_reward.argtypes = [C.POINTER(rcfg.RouteReward)] + [dp] * 11 + [C.c_int, C.c_double, dp]
        L.kp1o_route_reward.restype = C.c_double
        L.kp1o_route_env_init.argtypes = [C.c_void_p, C.POINTER(kcfg.Kp1Config), C.POINTER(rcfg.RouteConfig), C.c_void_p]
        L.kp1o_route_env_seed.argtypes = [C.c_void_p, C.c_uint64]
        L.kp1o_route_env_reset.argtypes = [C.c_void_p, C.c_int, C.c_int, dp, dp, dp, C.POINTER(C.c_float)]
        L.kp1o_route_env_step.argtypes = [C.c_void_p, dp, C.POINTER(C.c_float), C.POINTER(RouteStepOut)]
        L.kp1o_route_obs_dim.argtypes = [C.c_void_p]
        L.kp1o_sizeof_route_env.restype = C.c_size_t
        L.kp1o_route_env_rng.argtypes = [C.c_void_p]
        L.kp1o_route_env_rng.restype = C.POINTER(orc.ORng)
        L.kp1o_route_env_base.argtypes = [C.c_void_p]
        L.kp1o_route_env_base.restype = C.POINTER(orc.OEnv)
        L.kp1o_route_env_field.argtypes = [C.c_void_p, C.c_int]
        L.kp1o_route_env_set_window.argtypes = [C.c_void_p, C.c_int, C.c_int]
        _bound = True
    return L


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Route:
    """load_route_dataset on the CPU oracle"""

    def __init__(self, route_q: np.ndarray) -> None:
        self.q = np.ascontiguousarray(route_q, dtype=np.float64)
        self.n = self.q.shape[0]
        self._h = lib().kp1o_route_load(_dp(self.q), self.n)

        class _R(C.Structure):
            _fields_ = [("n", C.c_int), ("q", C.POINTER(C.c_double)), ("pose", C.POINTER(C.c_double)), ("next_dq", C.POINTER(C.c_double)),
                        ("progress", C.POINTER(C.c_double)), ("chunk", C.POINTER(C.c_int))]

        r = C.cast(self._h, C.POINTER(_R)).contents
        self.poses6 = np.ctypeslib.as_array(r.pose, shape=(self.n, 6)).copy()
        self.next_q_delta = np.ctypeslib.as_array(r.next_dq, shape=(self.n, 7)).copy()
        self.progress = np.ctypeslib.as_array(r.progress, shape=(self.n,)).copy()
        self.chunk_id = np.ctypeslib.as_array(r.chunk, shape=(self.n,)).copy()

    def __del__(self) -> None:  # pragma: no cover
        try:
            lib().kp1o_route_free(self._h)
        except Exception:
            pass


def route_reward(cfg: rcfg.RouteReward, *, prev_q, curr_q, goal_q, prev_pose6, curr_pose6, goal_pose6, tangent, action, prev_action, prev_dq, curr_dq,
                 ready_streak: int, nearest: float) -> tuple[float, np.ndarray]:
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (prev_q, curr_q, goal_q, prev_pose6, curr_pose6, goal_pose6, tangent, action, prev_action,
                                                                 prev_dq, curr_dq)]
    comps = np.zeros(N_COMP)
    r = lib().kp1o_route_reward(C.byref(cfg), *[_dp(a) for a in arrs], int(ready_streak), float(nearest), _dp(comps))
    return float(r), comps


class OracleRouteEnv:
    """RouteKinematicEnv / RouteSequenceKinematicEnv (by cfg.sequence_enabled) on the CPU oracle"""

    def __init__(self, base_cfg: kcfg.EnvConfig, route_cfg: rcfg.RouteConfig, route: Route) -> None:
        L = lib()
        self._buf = C.create_string_buffer(int(L.kp1o_sizeof_route_env()) + 16)
        self._p = C.c_void_p((C.addressof(self._buf) + 15) // 16 * 16)
        self.route = route
        self.base_cfg = base_cfg
        self.cfg = route_cfg
        L.kp1o_route_env_init(self._p, C.byref(base_cfg.c), C.byref(route_cfg), route._h)
        self.obs_dim = int(L.kp1o_route_obs_dim(self._p))

    def seed(self, seed: int) -> None:
        lib().kp1o_route_env_seed(self._p, seed)

    def rng_words(self) -> np.ndarray:
        return orc.rng_words(lib().kp1o_route_env_rng(self._p).contents)

    def set_route_window(self, *, max_route_index: int, min_route_index: int = 1) -> None:
        lib().kp1o_route_env_set_window(self._p, int(min_route_index), int(max_route_index))

    def field(self, name: str) -> int:
        return int(lib().kp1o_route_env_field(self._p, ["current_route_index", "start_route_index", "last_route_index", "ready_streak",
                                                        "completed_waypoints", "reset_mode"].index(name)))

    def base_state(self) -> dict[str, np.ndarray]:
        b = lib().kp1o_route_env_base(self._p).contents
        return {"q": np.array(b.q[:]), "dq": np.array(b.dq[:]), "prev_action": np.array(b.prev_action[:]), "goal_q": np.array(b.goal_q[:]),
                "goal_pose6": np.array(b.goal_pose6[:])}

    def reset(self, *, seed: int | None = None, options: dict[str, Any] | None = None) -> np.ndarray:
        if seed is not None:
            self.seed(seed)
        o = options or {}
        obs = np.zeros(self.obs_dim, dtype=np.float32)
        ri = int(o["route_index"]) if "route_index" in o else -1
        si = int(o.get("start_route_index", -1)) if "route_index" in o else -1
        arrs = [np.ascontiguousarray(o[k], dtype=np.float64) if k in o else None for k in ("initial_q", "initial_dq", "initial_prev_action")]
        lib().kp1o_route_env_reset(self._p, ri, si, *[_dp(a) if a is not None else None for a in arrs], obs.ctypes.data_as(C.POINTER(C.c_float)))
        return obs

    def step(self, action: np.ndarray) -> tuple[np.ndarray, RouteStepOut]:
        a = np.ascontiguousarray(action, dtype=np.float64)
        obs = np.zeros(self.obs_dim, dtype=np.float32)
        out = RouteStepOut()
        lib().kp1o_route_env_step(self._p, _dp(a), obs.ctypes.data_as(C.POINTER(C.c_float)), C.byref(out))
        return obs, out
