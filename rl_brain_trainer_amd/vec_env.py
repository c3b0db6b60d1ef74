"""Batched MI355X environment behind the reference's env API.

``ArmKinematicVecEnv`` is N ``ArmKinematicEnv`` instances (reference:
kinematic_phase1/envs/arm_kinematic_env.py:69-560) living on one GPU, stepped by one HIP
kernel launch through the C ABI of include/kp1.h.  Method names, argument meaning and error
behaviour follow the reference class and SB3's VecEnv where the trainers use it
(``env_method("set_curriculum_stage", k)``, auto-reset with ``terminal_observation``).

PyTorch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
from typing import Any

import numpy as np
import torch

from . import config as kcfg
from . import native


class _DevView:
    """Expose a raw device pointer to torch through __cuda_array_interface__ (zero copy)."""

    def __init__(self, ptr: int, shape: tuple[int, ...], typestr: str) -> None:
        self.__cuda_array_interface__ = {"data": (int(ptr), False), "shape": tuple(shape), "typestr": typestr, "version": 2, "strides": None}


def _view(ptr: int, shape: tuple[int, ...], typestr: str, device: torch.device) -> torch.Tensor:
    return torch.as_tensor(_DevView(ptr, shape, typestr), device=device)


class ArmKinematicVecEnv:
    """N kinematic_phase1 environments on one MI355X.

    Parameters mirror ``make_vec_env(ArmKinematicEnv, n_envs, seed)``: env ``i`` owns the numpy
    PCG64 stream ``default_rng(seed + first_env_id + i)``.
    """

    metadata = {"render_modes": []}

    def __init__(self, config: kcfg.EnvConfig, n_envs: int, *, device: int | torch.device = 0, seed: int = 0,
                 first_env_id: int = 0, real: str = "f32", reward_components: bool = False) -> None:
        if not torch.cuda.is_available():
            raise native.Kp1Error("ArmKinematicVecEnv needs a HIP device (torch.cuda.is_available() is False); there is no CPU fallback")
        self.L = native.load()
        self.config = config
        self.n_envs = int(n_envs)
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        self.real_type = {"f32": native.REAL_F32, "f64": native.REAL_F64}[real]
        self.dtype = torch.float32 if real == "f32" else torch.float64
        self._typestr = "<f4" if real == "f32" else "<f8"
        self._handle = C.c_void_p()
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            native.check(self.L.kp1_create(C.byref(config.c), self.n_envs, self.device.index or 0, self.real_type,
                                           int(seed), int(first_env_id), C.c_void_p(stream), C.byref(self._handle)))
        if config.handoff_states:
            arr = config.handoff_array()
            native.check(self.L.kp1_set_handoff_states(self._handle, C.cast(arr, C.c_void_p), len(config.handoff_states)))
        n = self.n_envs
        self.obs = torch.zeros((n, kcfg.OBS_DIM), dtype=torch.float32, device=self.device)
        self.terminal_obs = torch.zeros((n, kcfg.OBS_DIM), dtype=torch.float32, device=self.device)
        self.reward = torch.zeros(n, dtype=self.dtype, device=self.device)
        self.done = torch.zeros(n, dtype=torch.uint8, device=self.device)
        self._mode_name = config.mode_name
        self.obs_stride = kcfg.OBS_DIM
        self._components = False
        # bumped by every setter whose value a captured hipGraph of kp1_step would have frozen into kernel arguments (stage, mode, config,
        # reward-component switch, handoff buffer, observation pitch): ppo.PPO re-captures its rollout graph when it changes
        self.launch_args_version = 0
        if reward_components:
            self.enable_reward_components(True)

    # ------------------------------------------------------------------ lifecycle
    def close(self) -> None:
        if getattr(self, "_handle", None) is not None and self._handle.value:
            self.L.kp1_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self) -> None:  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ reference API
    def set_curriculum_stage(self, stage_index: int) -> None:
        native.check(self.L.kp1_set_stage(self._handle, int(stage_index)))
        self.launch_args_version += 1

    def get_curriculum_stage(self) -> int:
        out = C.c_int32()
        native.check(self.L.kp1_get_stage(self._handle, C.byref(out)))
        return int(out.value)

    def set_policy_mode(self, mode_name: str) -> None:
        if mode_name not in kcfg.MODE_NAMES:
            raise ValueError(f"Unsupported policy mode '{mode_name}'")
        native.check(self.L.kp1_set_mode(self._handle, kcfg.MODE_NAMES[mode_name]))
        self._mode_name = mode_name
        self.launch_args_version += 1

    def env_method(self, name: str, *args: Any) -> list[Any]:
        """VecEnv.env_method: the trainers only broadcast (callbacks.py:55,69,162)."""
        result = getattr(self, name)(*args)
        return [result] * self.n_envs

    def apply_dock_training_stage(self, stage_updates: dict[str, Any]) -> None:
        """arm_kinematic_env.py:459-487: env scalar + dock_reset overrides for the dock curriculum."""
        if self.config.mode_name != "dock":
            return
        c = self.config.c
        for key, value in dict(stage_updates.get("dock_reset", {})).items():
            if key in ("goal_q", "goal_noise", "init_q_noise", "close_init_q_noise"):
                getattr(c.dock_reset, key)[:] = [float(v) for v in value]
            elif key.startswith("handoff_state_") and key != "handoff_state_probability":
                raise NotImplementedError("changing the handoff buffer filter mid-run: rebuild the EnvConfig instead")
            else:
                if not hasattr(c.dock_reset, key):
                    raise TypeError(f"DockResetConfig.__init__() got an unexpected keyword argument '{key}'")
                setattr(c.dock_reset, key, type(getattr(c.dock_reset, key))(value))
        for key in (
            "action_delta_scale", "dock_action_delta_scale", "dock_residual_action_limit", "dock_delta_q_change_limit_scale",
            "dock_dynamic_action_limit_near_pos_threshold_m", "dock_dynamic_action_limit_far_pos_threshold_m",
            "dock_dynamic_residual_action_limit_near", "dock_dynamic_residual_action_limit_far",
            "dock_dynamic_delta_q_change_limit_scale_near", "dock_dynamic_delta_q_change_limit_scale_far",
        ):
            if key in stage_updates:
                setattr(c.env, key, float(stage_updates[key]))
        native.check(self.L.kp1_update_config(self._handle, C.byref(c)))
        self.launch_args_version += 1

    def set_handoff_states(self, states: list[dict[str, Any]]) -> None:
        """Replace the handoff-state buffer dock resets draw from (reset_samplers.py:131-165).  The old device buffer is freed: a
        rollout graph captured before this call must not be replayed again (PPO checks launch_args_version)."""
        self.config.handoff_states = list(states)
        arr = self.config.handoff_array()
        native.check(self.L.kp1_set_handoff_states(self._handle, C.cast(arr, C.c_void_p), len(self.config.handoff_states)))
        self.launch_args_version += 1

    def set_obs_stride(self, stride: int) -> None:
        """Row pitch of every observation buffer handed to reset/step (56, or 64 = zero-padded for the MFMA GEMMs)."""
        native.check(self.L.kp1_set_obs_stride(self._handle, int(stride)))
        self.obs_stride = int(stride)
        self.launch_args_version += 1
        self.obs = torch.zeros((self.n_envs, self.obs_stride), dtype=torch.float32, device=self.device)
        self.terminal_obs = torch.zeros_like(self.obs)

    def seed(self, seed: int, first_env_id: int = 0) -> None:
        native.check(self.L.kp1_seed(self._handle, int(seed), int(first_env_id)))

    def reset(self, *, seed: int | None = None, options: dict[str, Any] | None = None, mask: torch.Tensor | None = None) -> torch.Tensor:
        """reset(seed=, options=) for all envs (or those where mask != 0).  options values are
        [N,7] / [N,6] arrays (or a single row, broadcast), like the reference's per-env dict."""
        if seed is not None:
            self.seed(seed)
        opts_p = None
        keep: list[np.ndarray] = []
        if options:
            o = kcfg.ResetOpts()
            o.policy_mode = -1
            for key, width in (("initial_q", 7), ("initial_dq", 7), ("initial_prev_action", 7), ("goal_q", 7), ("goal_pose6", 6)):
                val = options.get(key)
                if val is None:
                    continue
                arr = np.asarray(val, dtype=np.float64)
                if arr.ndim == 1:
                    arr = np.broadcast_to(arr, (self.n_envs, arr.shape[0]))
                if arr.shape != (self.n_envs, width):
                    raise ValueError(f"options['{key}'] must have shape ({self.n_envs}, {width}), got {arr.shape}")
                arr = np.ascontiguousarray(arr)
                keep.append(arr)
                setattr(o, key, arr.ctypes.data)
            pm = options.get("policy_mode")
            if pm is not None:
                if pm not in kcfg.MODE_NAMES:
                    raise ValueError(f"Unsupported policy mode '{pm}'")
                o.policy_mode = kcfg.MODE_NAMES[pm]
                self._mode_name = pm
            opts_p = C.byref(o)
        else:
            self._mode_name = self.config.mode_name
        mask_ptr = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            mask_ptr = C.c_void_p(mask.data_ptr())
        native.check(self.L.kp1_reset(self._handle, mask_ptr, opts_p, C.c_void_p(self.obs.data_ptr())))
        return self.obs

    def step(self, actions: torch.Tensor, *, auto_reset: bool = True) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """step(actions[N,7]) -> (obs[N,56] f32, reward[N], done[N] u8 bitmask).  Finished episodes are reset in the
        same launch (VecEnv semantics); their last observation is in ``self.terminal_obs``."""
        if actions.shape != (self.n_envs, kcfg.NJ):
            raise ValueError(f"Expected action shape {(self.n_envs, kcfg.NJ)}, got {tuple(actions.shape)}")
        a = actions.to(device=self.device, dtype=self.dtype).contiguous()
        native.check(self.L.kp1_step(self._handle, C.c_void_p(a.data_ptr()), C.c_void_p(self.obs.data_ptr()),
                                     C.c_void_p(self.reward.data_ptr()), C.c_void_p(self.done.data_ptr()),
                                     C.c_void_p(self.terminal_obs.data_ptr()), int(auto_reset)))
        return self.obs, self.reward, self.done

    def use_current_stream(self) -> None:
        """Order this handle's launches on torch's current stream (call inside torch.cuda.graph capture / stream contexts)."""
        native.check(self.L.kp1_set_stream(self._handle, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def snapshot(self) -> None:
        """Device-side copy of the whole env state (all fields, counters and PCG64 streams), ordered on the handle's stream."""
        native.check(self.L.kp1_state_snapshot(self._handle))

    def restore(self) -> None:
        """Put the state of the last snapshot() back (episodes and random streams continue exactly from there)."""
        native.check(self.L.kp1_state_restore(self._handle))

    def step_into(self, actions: torch.Tensor, obs: torch.Tensor, reward: torch.Tensor, done: torch.Tensor,
                  terminal_obs: torch.Tensor | None, auto_reset: bool = True) -> None:
        """Zero-copy variant used by the rollout loop: outputs go straight into caller-owned (rollout) buffers."""
        native.check(self.L.kp1_step(self._handle, C.c_void_p(actions.data_ptr()), C.c_void_p(obs.data_ptr()),
                                     C.c_void_p(reward.data_ptr()), C.c_void_p(done.data_ptr()),
                                     C.c_void_p(terminal_obs.data_ptr()) if terminal_obs is not None else None, int(auto_reset)))

    def current_observation(self) -> torch.Tensor:
        out = torch.empty_like(self.obs)
        native.check(self.L.kp1_observe(self._handle, C.c_void_p(out.data_ptr())))
        return out

    @staticmethod
    def obs_dict(obs: torch.Tensor) -> dict[str, torch.Tensor]:
        """Split [N,56] into the reference's 13-key Dict observation (views)."""
        return {k: obs[..., s:s + n] for k, (s, n) in kcfg.OBS_LAYOUT.items()}

    # ------------------------------------------------------------------ info / state
    def info(self) -> dict[str, torch.Tensor]:
        """Device views of the info keys downstream code reads (arm_kinematic_env.py:384-423)."""
        v = kcfg.InfoView()
        native.check(self.L.kp1_get_info(self._handle, C.byref(v)))
        n, ts, dev = self.n_envs, self._typestr, self.device
        out = {
            "position_error_norm": _view(v.position_error_norm, (n,), ts, dev),
            "orientation_error_norm": _view(v.orientation_error_norm, (n,), ts, dev),
            "min_position_error": _view(v.min_position_error, (n,), ts, dev),
            "executed_delta_q_l2": _view(v.executed_delta_q_l2, (n,), ts, dev),
            "action_l2": _view(v.action_l2, (n,), ts, dev),
            "delta_q_change_l2": _view(v.delta_q_change_l2, (n,), ts, dev),
            "q": _view(v.q, (7, n), ts, dev), "dq": _view(v.dq, (7, n), ts, dev),
            "prev_action": _view(v.prev_action, (7, n), ts, dev), "goal_q": _view(v.goal_q, (7, n), ts, dev),
            "goal_pose6": _view(v.goal_pose6, (6, n), ts, dev), "ee_pose6": _view(v.ee_pose6, (6, n), ts, dev),
            "entry_metrics": _view(v.entry_metrics, (4, n), ts, dev),
            "step_count": _view(v.episode_step, (n,), "<i4", dev), "dwell_count": _view(v.dwell_count, (n,), "<i4", dev),
            "near_goal_entry_count": _view(v.near_goal_entry_count, (n,), "<i4", dev),
            "near_goal_drift_count": _view(v.near_goal_drift_count, (n,), "<i4", dev),
            "flags": _view(v.flags, (n,), "<i4", dev), "stage_index": _view(v.stage_index, (n,), "<i4", dev),
        }
        return out

    def enable_reward_components(self, enable: bool = True) -> None:
        native.check(self.L.kp1_enable_reward_components(self._handle, int(enable)))
        self._reward_components_on = bool(enable)
        self._components = bool(enable)
        self.launch_args_version += 1

    def reward_components(self) -> tuple[list[str], torch.Tensor]:
        ptr, n = C.c_void_p(), C.c_int32()
        native.check(self.L.kp1_get_reward_components(self._handle, C.byref(ptr), C.byref(n)))
        mode = kcfg.MODE_NAMES[self._mode_name]
        return native.component_names(mode), _view(ptr.value, (int(n.value), self.n_envs), self._typestr, self.device)

    def get_state(self) -> dict[str, np.ndarray]:
        n = self.n_envs
        out = {k: np.empty((n, 6 if k == "goal_pose6" else 7)) for k in ("q", "dq", "prev_action", "goal_q", "goal_pose6")}
        native.check(self.L.kp1_get_state(self._handle, *[out[k].ctypes.data for k in ("q", "dq", "prev_action", "goal_q", "goal_pose6")]))
        return out

    def set_state(self, *, q=None, dq=None, prev_action=None, goal_q=None, goal_pose6=None, capture_entry_metrics: bool = True) -> None:
        ptrs, keep = [], []
        for val, w in ((q, 7), (dq, 7), (prev_action, 7), (goal_q, 7), (goal_pose6, 6)):
            if val is None:
                ptrs.append(None)
                continue
            arr = np.ascontiguousarray(np.broadcast_to(np.asarray(val, dtype=np.float64), (self.n_envs, w)))
            keep.append(arr)
            ptrs.append(arr.ctypes.data)
        native.check(self.L.kp1_set_state(self._handle, *ptrs, int(capture_entry_metrics)))

    def rng_state(self) -> np.ndarray:
        """[N,6] uint64 words (state_hi, state_lo, inc_hi, inc_lo, has_uint32, uinteger) of each env's PCG64 stream."""
        arr = (kcfg.RngState * self.n_envs)()
        native.check(self.L.kp1_rng_get(self._handle, C.cast(arr, C.c_void_p)))
        return np.array([[s.state_hi, s.state_lo, s.inc_hi, s.inc_lo, s.has_uint32, s.uinteger] for s in arr], dtype=np.uint64)

    def set_rng_state(self, words: np.ndarray) -> None:
        arr = (kcfg.RngState * self.n_envs)()
        for i in range(self.n_envs):
            (arr[i].state_hi, arr[i].state_lo, arr[i].inc_hi, arr[i].inc_lo) = (int(w) for w in words[i][:4])
            arr[i].has_uint32, arr[i].uinteger = int(words[i][4]), int(words[i][5])
        native.check(self.L.kp1_rng_set(self._handle, C.cast(arr, C.c_void_p)))


def fk_pose6(q: torch.Tensor) -> torch.Tensor:
    """compute_ee_pose6 batched on the GPU (kinematics/fk_interface.py:21-22): q[n,7] -> pose6[n,6]."""
    if q.ndim != 2 or q.shape[1] != kcfg.NJ:
        raise ValueError("Expected q of shape (n, 7)")
    if not q.is_cuda:
        raise native.Kp1Error("fk_pose6 needs a device tensor; there is no CPU fallback")
    L = native.load()
    q = q.contiguous()
    rt = native.REAL_F64 if q.dtype == torch.float64 else native.REAL_F32
    if q.dtype not in (torch.float32, torch.float64):
        raise ValueError("q must be float32 or float64")
    out = torch.empty((q.shape[0], 6), dtype=q.dtype, device=q.device)
    stream = torch.cuda.current_stream(q.device).cuda_stream
    native.check(L.kp1_fk_pose6(q.device.index or 0, rt, C.c_void_p(q.data_ptr()), C.c_void_p(out.data_ptr()), q.shape[0], C.c_void_p(stream)))
    return out


def _real_type(t: torch.Tensor) -> int:
    if t.dtype not in (torch.float32, torch.float64):
        raise ValueError("expected a float32 or float64 tensor")
    if not t.is_cuda:
        raise native.Kp1Error("needs a device tensor; there is no CPU fallback")
    return native.REAL_F64 if t.dtype == torch.float64 else native.REAL_F32


def pose_error_components(curr_pose6: torch.Tensor, goal_pose6: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """kinematics/pose_utils.py:21-30 batched on the GPU through the device function of the step kernel:
    (pos_err[n,3], ori_err[n,3] wrapped per component to [-pi, pi), norms[n,2])."""
    if curr_pose6.shape != goal_pose6.shape or curr_pose6.ndim != 2 or curr_pose6.shape[1] != 6 or curr_pose6.dtype != goal_pose6.dtype:
        raise ValueError("Expected two pose6 tensors of shape (n, 6) and equal dtype")
    rt = _real_type(curr_pose6)
    c, g = curr_pose6.contiguous(), goal_pose6.contiguous()
    n = c.shape[0]
    pe, oe, nr = (torch.empty((n, k), dtype=c.dtype, device=c.device) for k in (3, 3, 2))
    stream = torch.cuda.current_stream(c.device).cuda_stream
    native.check(native.load().kp1_pose_error(c.device.index or 0, rt, C.c_void_p(c.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(pe.data_ptr()),
                                              C.c_void_p(oe.data_ptr()), C.c_void_p(nr.data_ptr()), n, C.c_void_p(stream)))
    return pe, oe, nr


def joint_utils(config: kcfg.EnvConfig, q: torch.Tensor, dq: torch.Tensor) -> dict[str, torch.Tensor]:
    """kinematics/joint_limits.py:133-174 batched on the GPU through the device functions of the hot path:
    clipped = clip(q), margin = joint_limit_margin(clipped), q_norm = normalize_joint_positions(q), dq_norm = normalize_joint_deltas(dq)."""
    if q.shape != dq.shape or q.ndim != 2 or q.shape[1] != kcfg.NJ or q.dtype != dq.dtype:
        raise ValueError("Expected q and dq of shape (n, 7) and equal dtype")
    rt = _real_type(q)
    q, dq = q.contiguous(), dq.contiguous()
    out = {k: torch.empty_like(q) for k in ("clipped", "margin", "q_norm", "dq_norm")}
    stream = torch.cuda.current_stream(q.device).cuda_stream
    native.check(native.load().kp1_joint_utils(q.device.index or 0, rt, C.byref(config.c), C.c_void_p(q.data_ptr()), C.c_void_p(dq.data_ptr()),
                                               *(C.c_void_p(out[k].data_ptr()) for k in ("clipped", "margin", "q_norm", "dq_norm")), q.shape[0],
                                               C.c_void_p(stream)))
    return out
