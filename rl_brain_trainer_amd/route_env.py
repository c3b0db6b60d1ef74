"""Route-curriculum environments on the device: N ``RouteKinematicEnv`` / ``RouteSequenceKinematicEnv`` at once.

Mirror of kinematic_phase1/route/route_env.py:29-212 and route_sequence_env.py:29-278 (which of the two is chosen by
``route.sequence.enabled``, like train_route_curriculum.py:104-107) behind the C ABI of include/kp1_route.h.  Same reset
options (``route_index``, ``start_route_index``, ``initial_q`` / ``initial_dq`` / ``initial_prev_action`` for the sequence
env), same ``set_route_window``, same info keys; observations are rows of 56 floats, or 80 with ``include_route_keys`` (the
17 Dict keys in SB3's sorted order: route_config.ROUTE_OBS_LAYOUT).  Env i owns ``default_rng(seed + first_env_id + i)`` for
the wrapper's reset stream, as ``make_vec_env(make_env, n_envs, seed)`` gives it.
"""
from __future__ import annotations

import ctypes as C
from typing import Any

import numpy as np
import torch

from . import config as kcfg
from . import native
from . import route_config as rcfg
from .vec_env import ArmKinematicVecEnv, _view


class _RouteResetOpts(C.Structure):
    _fields_ = [("route_index", C.c_void_p), ("start_route_index", C.c_void_p), ("initial_q", C.c_void_p), ("initial_dq", C.c_void_p),
                ("initial_prev_action", C.c_void_p), ("evaluator_state", C.c_int32), ("pad_", C.c_int32)]


class _RouteInfoView(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("route_index", "start_route_index", "last_route_index", "reset_mode", "ready_streak", "completed_waypoints",
                                          "route_ready", "waypoint_success", "route_regression", "orientation_hit", "q_error_norm",
                                          "nearest_route_q_distance")]


def _bind(L) -> None:
    if getattr(L, "_kp1_route_bound", False):
        return
    vp, i32, u64 = C.c_void_p, C.c_int32, C.c_uint64
    L.kp1_route_create.argtypes = [vp, C.POINTER(rcfg.RouteConfig), vp, i32, u64, u64, C.POINTER(vp)]
    L.kp1_route_destroy.argtypes = [vp]
    L.kp1_route_get_dataset.argtypes = [vp, vp, vp, vp, vp]
    L.kp1_route_set_window.argtypes = [vp, i32, i32]
    L.kp1_route_seed.argtypes = [vp, u64, u64]
    L.kp1_route_obs_dim.argtypes = [vp]
    L.kp1_route_set_obs_stride.argtypes = [vp, C.c_int32]
    L.kp1_route_reset.argtypes = [vp, vp, C.POINTER(_RouteResetOpts), vp]
    L.kp1_route_step.argtypes = [vp, vp, vp, vp, vp, vp, i32]
    L.kp1_route_get_info.argtypes = [vp, C.POINTER(_RouteInfoView)]
    L.kp1_route_enable_reward_components.argtypes = [vp, i32]
    L.kp1_route_get_reward_components.argtypes = [vp, C.POINTER(vp)]
    L.kp1_route_component_name.argtypes = [i32]
    L.kp1_route_component_name.restype = C.c_char_p
    L.kp1_route_rng_get.argtypes = [vp, vp]
    L.kp1_route_rng_set.argtypes = [vp, vp]
    L.kp1_route_config_default.argtypes = [C.POINTER(rcfg.RouteConfig)]
    L._kp1_route_bound = True


class RouteVecEnv:
    """``route_q``: [W, 7] joint goals of the dense route (route_config.load_route_q); ``base_config``: the env block of the route
    YAML (approach mode); ``route_cfg``: route_config.route_config_from_dict(cfg, max_route_index=first prefix)."""

    def __init__(self, base_config: kcfg.EnvConfig, route_cfg: rcfg.RouteConfig, route_q: np.ndarray, n_envs: int, *, device: int | torch.device = 0,
                 seed: int = 0, first_env_id: int = 0, real: str = "f32", reward_components: bool = False) -> None:
        self.base = ArmKinematicVecEnv(base_config, n_envs, device=device, seed=seed, first_env_id=first_env_id, real=real)
        self.L = self.base.L
        _bind(self.L)
        self.config = base_config
        self.route_cfg = route_cfg
        self.n_envs = int(n_envs)
        self.device = self.base.device
        self.dtype = self.base.dtype
        self.route_q = np.ascontiguousarray(route_q, dtype=np.float64)
        if self.route_q.ndim != 2 or self.route_q.shape[1] != kcfg.NJ:
            raise ValueError("route_q must have shape (n_waypoints, 7)")
        self.n_waypoints = int(self.route_q.shape[0])
        self._handle = C.c_void_p()
        with torch.cuda.device(self.device):
            native.check(self.L.kp1_route_create(self.base._handle, C.byref(route_cfg), C.c_void_p(self.route_q.ctypes.data), self.n_waypoints, int(seed),
                                                 int(first_env_id), C.byref(self._handle)))
        self.obs_dim = int(self.L.kp1_route_obs_dim(self._handle))
        self.obs_stride = self.obs_dim
        n = self.n_envs
        self.obs = torch.zeros((n, self.obs_dim), dtype=torch.float32, device=self.device)
        self.terminal_obs = torch.zeros_like(self.obs)
        self.reward = torch.zeros(n, dtype=self.dtype, device=self.device)
        self.done = torch.zeros(n, dtype=torch.uint8, device=self.device)
        W = self.n_waypoints
        self.poses6 = np.zeros((W, 6))
        self.route_progress_m = np.zeros(W)
        self.next_q_delta = np.zeros((W, kcfg.NJ))
        self.chunk_id = np.zeros(W, dtype=np.int32)
        native.check(self.L.kp1_route_get_dataset(self._handle, C.c_void_p(self.poses6.ctypes.data), C.c_void_p(self.route_progress_m.ctypes.data),
                                                  C.c_void_p(self.next_q_delta.ctypes.data), C.c_void_p(self.chunk_id.ctypes.data)))
        self._keep: list[Any] = []
        if reward_components:
            self.enable_reward_components(True)

    def close(self) -> None:
        if getattr(self, "_handle", None) is not None and self._handle.value:
            self.L.kp1_route_destroy(self._handle)
            self._handle = C.c_void_p()
        self.base.close()

    def __del__(self) -> None:  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ reference API
    def set_route_window(self, *, max_route_index: int, min_route_index: int = 1) -> None:
        native.check(self.L.kp1_route_set_window(self._handle, int(min_route_index), int(max_route_index)))
        self.route_cfg.reset.min_route_index = int(min_route_index)
        self.route_cfg.reset.max_route_index = int(max_route_index)

    def env_method(self, name: str, *args: Any, **kwargs: Any) -> list[Any]:
        result = getattr(self, name)(*args, **kwargs)
        return [result] * self.n_envs

    def seed(self, seed: int, first_env_id: int = 0) -> None:
        native.check(self.L.kp1_route_seed(self._handle, int(seed), int(first_env_id)))

    def reset(self, *, seed: int | None = None, options: dict[str, Any] | None = None, mask: torch.Tensor | None = None) -> torch.Tensor:
        """options: {"route_index": int | [N], "start_route_index", "initial_q" [N,7], "initial_dq", "initial_prev_action"} or None = sample.
        ``evaluator_state=True`` makes the single-waypoint env honour the initial state too (the sequential evaluator's chained resets)."""
        if seed is not None:
            self.seed(seed)
        opts = None
        self._keep = []
        if options and "route_index" in options:
            n = self.n_envs

            def ints(v):
                t = torch.as_tensor(np.broadcast_to(np.asarray(v, dtype=np.int32), (n,)).copy(), device=self.device)
                self._keep.append(t)
                return t.data_ptr()

            def mats(v):
                t = torch.as_tensor(np.broadcast_to(np.asarray(v, dtype=np.float64), (n, kcfg.NJ)).copy(), device=self.device)
                self._keep.append(t)
                return t.data_ptr()

            opts = _RouteResetOpts()
            opts.route_index = ints(options["route_index"])
            opts.start_route_index = ints(options["start_route_index"]) if "start_route_index" in options else None
            for key in ("initial_q", "initial_dq", "initial_prev_action"):
                setattr(opts, key, mats(options[key]) if key in options else None)
            opts.evaluator_state = int(bool(options.get("evaluator_state", False)))
        m = None
        if mask is not None:
            m = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            self._keep.append(m)
        native.check(self.L.kp1_route_reset(self._handle, C.c_void_p(m.data_ptr()) if m is not None else None, C.byref(opts) if opts is not None else None,
                                            C.c_void_p(self.obs.data_ptr())))
        return self.obs

    def use_current_stream(self) -> None:
        """Order this handle's launches on torch's current stream (the wrapper kernels run on the base handle's stream)."""
        self.base.use_current_stream()

    def set_obs_stride(self, stride: int) -> None:
        """Row pitch of every observation buffer handed to reset / step (obs_dim, or the MFMA kernels' padded 64 / 128)."""
        native.check(self.L.kp1_route_set_obs_stride(self._handle, int(stride)))
        self.obs_stride = int(stride)
        self.obs = torch.zeros((self.n_envs, self.obs_stride), dtype=torch.float32, device=self.device)
        self.terminal_obs = torch.zeros_like(self.obs)

    def step_into(self, actions: torch.Tensor, obs: torch.Tensor, reward: torch.Tensor, done: torch.Tensor, terminal_obs: torch.Tensor | None,
                  auto_reset: bool = True) -> None:
        """Zero-copy variant used by the PPO rollout loop (ArmKinematicVecEnv.step_into): outputs go into caller-owned buffers."""
        native.check(self.L.kp1_route_step(self._handle, C.c_void_p(actions.data_ptr()), C.c_void_p(obs.data_ptr()), C.c_void_p(reward.data_ptr()),
                                           C.c_void_p(done.data_ptr()), C.c_void_p(terminal_obs.data_ptr()) if terminal_obs is not None else None,
                                           int(auto_reset)))

    def step(self, actions: torch.Tensor, *, auto_reset: bool = True) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """actions [N, 7] in the env's real type; returns (obs [N, obs_dim], reward [N], done bits [N] uint8)."""
        if actions.shape != (self.n_envs, kcfg.NJ):
            raise ValueError(f"Expected action shape {(self.n_envs, kcfg.NJ)}, got {tuple(actions.shape)}")
        a = actions.to(device=self.device, dtype=self.dtype).contiguous()
        native.check(self.L.kp1_route_step(self._handle, C.c_void_p(a.data_ptr()), C.c_void_p(self.obs.data_ptr()), C.c_void_p(self.reward.data_ptr()),
                                           C.c_void_p(self.done.data_ptr()), C.c_void_p(self.terminal_obs.data_ptr()), int(auto_reset)))
        return self.obs, self.reward, self.done

    @staticmethod
    def obs_dict(obs: torch.Tensor) -> dict[str, torch.Tensor]:
        layout = rcfg.ROUTE_OBS_LAYOUT if obs.shape[-1] in (rcfg.ROUTE_OBS_DIM, 128) else kcfg.OBS_LAYOUT   # 128 / 64: padded PPO rows
        return {k: obs[..., o:o + w] for k, (o, w) in layout.items()}

    def info(self) -> dict[str, torch.Tensor]:
        """Base env info (ArmKinematicVecEnv.info) plus the wrapper's keys; tensors are views of device state."""
        v = _RouteInfoView()
        native.check(self.L.kp1_route_get_info(self._handle, C.byref(v)))
        n = self.n_envs
        ts = "<f4" if self.dtype == torch.float32 else "<f8"
        out = dict(self.base.info())
        for key, name in (("route_index", "route_index"), ("start_route_index", "start_route_index"), ("last_route_index", "last_route_index"),
                          ("route_reset_mode", "reset_mode"), ("route_ready_streak", "ready_streak"), ("route_completed_waypoints", "completed_waypoints")):
            out[key] = _view(getattr(v, name), (n,), "<i4", self.device)
        for key, name in (("route_ready", "route_ready"), ("route_waypoint_success", "waypoint_success"), ("route_regression", "route_regression"),
                          ("route_orientation_hit", "orientation_hit")):
            out[key] = _view(getattr(v, name), (n,), "|u1", self.device)
        out["route_q_error_norm"] = _view(v.q_error_norm, (n,), ts, self.device)
        out["nearest_route_q_distance"] = _view(v.nearest_route_q_distance, (n,), ts, self.device)
        idx = out["route_index"].long().clamp(0, self.n_waypoints - 1).cpu().numpy()
        out["route_progress_m"] = torch.as_tensor(self.route_progress_m[idx])
        out["route_chunk_id"] = torch.as_tensor(self.chunk_id[idx])
        return out

    def episode_flags(self) -> torch.Tensor:
        """[3, N] uint8 device tensor (route_ready, route_orientation_hit, route_regression) of the last step: what the prefix curriculum
        reads per finished episode, without building the whole info dict."""
        if getattr(self, "_flag_views", None) is None:
            v = _RouteInfoView()
            native.check(self.L.kp1_route_get_info(self._handle, C.byref(v)))
            self._flag_views = tuple(_view(getattr(v, name), (self.n_envs,), "|u1", self.device) for name in ("route_ready", "orientation_hit", "route_regression"))
        return torch.stack(self._flag_views)

    def enable_reward_components(self, enable: bool = True) -> None:
        native.check(self.L.kp1_route_enable_reward_components(self._handle, int(enable)))

    def reward_components(self) -> tuple[list[str], torch.Tensor]:
        p = C.c_void_p()
        native.check(self.L.kp1_route_get_reward_components(self._handle, C.byref(p)))
        ts = "<f4" if self.dtype == torch.float32 else "<f8"
        return list(rcfg.COMPONENT_NAMES), _view(p.value, (len(rcfg.COMPONENT_NAMES), self.n_envs), ts, self.device)

    def rng_state(self) -> np.ndarray:
        arr = (kcfg.RngState * self.n_envs)()
        native.check(self.L.kp1_route_rng_get(self._handle, C.cast(arr, C.c_void_p)))
        return np.array([[s.state_hi, s.state_lo, s.inc_hi, s.inc_lo, s.has_uint32, s.uinteger] for s in arr], dtype=np.uint64)

    def get_state(self) -> dict[str, np.ndarray]:
        return self.base.get_state()
