"""Python handle on the MFMA actor-critic kernels (include/kp1_ppo.h, csrc/kp1_mlp.hip)."""
from __future__ import annotations

import ctypes as C

import torch

from . import native

OBS_DIM, OBS_PAD, ACT_DIM = 56, 64, 7


def _p(t: torch.Tensor | None):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class MlpKernels:
    def __init__(self, hidden: int, device: torch.device, max_batch: int = 8192, obs_dim: int = OBS_DIM) -> None:
        self.L = native.load()
        L = self.L
        vp, i32, f32 = C.c_void_p, C.c_int32, C.c_float
        L.kp1_mlp_create_ex.argtypes = [i32, i32, i32, i32, C.POINTER(vp)]
        L.kp1_mlp_destroy.argtypes = [vp]
        L.kp1_mlp_num_params_ex.argtypes = [i32, i32]
        L.kp1_mlp_num_params_ex.restype = C.c_int64
        L.kp1_mlp_pack_weights.argtypes = [vp, vp, vp]
        L.kp1_mlp_forward.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp]
        L.kp1_mlp_forward_env_step.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.kp1_mlp_loss_grad.argtypes = [vp, vp, i32, vp, i32, vp, vp, vp, vp, f32, f32, vp, f32, f32, f32, f32, vp, vp, i32, vp]
        L.kp1_mlp_adam_step.argtypes = [vp, vp, vp, vp, vp, f32, f32, f32, i32, i32, vp]
        L.kp1_mlp_time_kernels.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp]
        L.kp1_mlp_placement_check.argtypes = [i32, i32, vp, vp]
        L.kp1_mlp_set_option.argtypes = [vp, i32, i32]
        L.kp1_mlp_profile_read.argtypes = [vp, vp, vp]
        self.hidden = hidden
        self.device = device
        self.max_batch = int(max_batch)
        self._h = vp()
        self.obs_dim = int(obs_dim)                      # 56, or 80 with the route observation keys
        self.obs_pad = 64 if self.obs_dim <= 64 else 128   # row pitch the kernels also accept (zero padded)
        native.check(L.kp1_mlp_create_ex(device.index or 0, hidden, self.obs_dim, self.max_batch, C.byref(self._h)))
        self.num_params = int(L.kp1_mlp_num_params_ex(hidden, self.obs_dim))

    def close(self) -> None:
        if self._h.value:
            self.L.kp1_mlp_destroy(self._h)
            self._h = C.c_void_p()

    OPT_FUSED = 1

    def set_fused(self, on: bool) -> None:
        """hidden = 256: whole activation chain of a 32-row tile in one workgroup (default) vs the layer-wise kernels."""
        native.check(self.L.kp1_mlp_set_option(self._h, self.OPT_FUSED, int(bool(on))))

    OPT_ACTOR_EXTRA_STEPS = 2

    def set_actor_extra_steps(self, steps: int) -> None:
        """Adam steps the actor tensors have taken beyond the common count (teacher-anchor side updates, teacher_anchor.py)."""
        native.check(self.L.kp1_mlp_set_option(self._h, self.OPT_ACTOR_EXTRA_STEPS, int(steps)))

    OPT_STEP_COUNT = 3

    def set_step_count(self, steps: int) -> None:
        """Device-resident optimiser step count (restoring a checkpoint's Adam state)."""
        native.check(self.L.kp1_mlp_set_option(self._h, self.OPT_STEP_COUNT, int(steps)))

    OPT_BF16X3_WGRAD = 5

    def set_bf16x3_wgrad(self, on: bool) -> None:
        """EXPERIMENT (default off): weight-gradient GEMMs on operands pre-split into three bf16 pieces, six bf16 MFMAs per product block
        (gemm_tn_bf16x3_kernel).  Not the exact path; bench.py's value never uses it."""
        native.check(self.L.kp1_mlp_set_option(self._h, self.OPT_BF16X3_WGRAD, int(bool(on))))

    OPT_PROFILE = 4
    PROFILE_SLOTS = ("mlp_train_tile", "gemm_tn_split", "grad_finalize", "adam")

    def set_profile(self, on: bool) -> None:
        """HIP-event pairs around every launch of the optimiser step (eager launches only); read with ``profile_read``."""
        native.check(self.L.kp1_mlp_set_option(self._h, self.OPT_PROFILE, int(bool(on))))

    def profile_read(self) -> dict[str, dict[str, float]]:
        """average in-situ duration (us) and launch count of each optimiser-step kernel since the last read"""
        us = (C.c_float * 4)()
        cnt = (C.c_int32 * 4)()
        native.check(self.L.kp1_mlp_profile_read(self._h, C.cast(us, C.c_void_p), C.cast(cnt, C.c_void_p)))
        return {name: {"us": float(us[i]), "launches": int(cnt[i])} for i, name in enumerate(self.PROFILE_SLOTS)}

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def pack(self, flat_params: torch.Tensor) -> None:
        assert flat_params.numel() == self.num_params and flat_params.dtype == torch.float32 and flat_params.is_contiguous()
        native.check(self.L.kp1_mlp_pack_weights(self._h, _p(flat_params), self._stream()))

    def forward(self, obs: torch.Tensor, *, noise=None, mean=None, value=None, action=None, clipped=None, log_prob=None) -> None:
        """obs [n, obs_dim | obs_pad] contiguous f32; outputs written in place (None = skip)."""
        n, stride = obs.shape[0], obs.shape[1]
        assert obs.is_contiguous() and obs.dtype == torch.float32
        native.check(self.L.kp1_mlp_forward(self._h, _p(obs), stride, n, _p(noise), _p(mean), _p(value), _p(action), _p(clipped), _p(log_prob), self._stream()))

    def forward_env_step(self, env, obs: torch.Tensor, *, noise, value, action, log_prob, next_obs, reward, done, terminal_obs) -> None:
        """One rollout step in one launch (kp1_mlp_forward_env_step): policy.forward on `obs` (row m = env m of `env`, an fp32
        ArmKinematicVecEnv), sampling, and VecEnv.step of every env with auto-reset, written into the caller's rollout buffers."""
        assert obs.is_contiguous() and obs.dtype == torch.float32 and obs.shape[0] == env.n_envs
        native.check(self.L.kp1_mlp_forward_env_step(self._h, env._handle, _p(obs), obs.shape[1], _p(noise), _p(value), _p(action), _p(log_prob),
                                                     _p(next_obs), _p(reward), _p(done), _p(terminal_obs), self._stream()))

    def mean_value(self, obs: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        n = obs.shape[0]
        mean = torch.empty((n, ACT_DIM), dtype=torch.float32, device=obs.device)
        value = torch.empty(n, dtype=torch.float32, device=obs.device)
        for s in range(0, n, self.max_batch):
            e = min(s + self.max_batch, n)
            self.forward(obs[s:e], mean=mean[s:e], value=value[s:e])
        return mean, value

    def loss_grad(self, obs: torch.Tensor, idx: torch.Tensor | None, n: int, actions, old_logp, adv, ret, *, clip_range: float, ent_coef: float,
                  vf_coef: float, inv_count: float, grad_out: torch.Tensor, stats_out: torch.Tensor | None, adv_stats: torch.Tensor | None = None,
                  normalize: bool = True, grad_is_zero: bool = False) -> None:
        assert obs.is_contiguous() and grad_out.numel() == self.num_params
        inv_std = 0.0 if normalize else -1.0
        native.check(self.L.kp1_mlp_loss_grad(self._h, _p(obs), obs.shape[-1], _p(idx), n, _p(actions), _p(old_logp), _p(adv), _p(ret), 0.0, inv_std,
                                              _p(adv_stats), clip_range, ent_coef, vf_coef, inv_count, _p(grad_out), _p(stats_out), int(grad_is_zero), self._stream()))

    def adam_step(self, params, grad, exp_avg, exp_avg_sq, *, lr: float, eps: float, max_grad_norm: float, step: int,
                  fused_norm: bool = False) -> None:
        """fused_norm: grad is exactly what the last loss_grad call wrote (no all-reduce in between), so the sum-of-squares
        partials its finalize kernel left are reused and the separate norm reduction launch is skipped."""
        native.check(self.L.kp1_mlp_adam_step(self._h, _p(params), _p(grad), _p(exp_avg), _p(exp_avg_sq), lr, eps, max_grad_norm, step,
                                              2 if fused_norm else 0, self._stream()))

    def placement_check(self, n_rows: int) -> dict[str, int | float]:
        """where the hardware puts the workgroups of a training-tile launch for an n_rows minibatch (the two placement assumptions the update
        kernels' SPEED rests on: include/kp1_ppo.h kp1_mlp_placement_check)"""
        out = (C.c_int32 * 8)()
        native.check(self.L.kp1_mlp_placement_check(self.device.index or 0, int(n_rows), C.cast(out, C.c_void_p), self._stream()))
        wgs, pairs, same, rr, cus, used, arrived, xcds = (int(v) for v in out)
        return {"workgroups": wgs, "compute_units": cus, "compute_units_used": used, "second_tile_on_same_cu": same, "second_tile_pairs": pairs,
                "on_xcd_of_block_index_mod_8": rr, "xcds_among_first_8_blocks": xcds, "all_resident": arrived == wgs}

    def time_kernels(self, obs: torch.Tensor, n: int, iters: int = 20) -> dict[str, dict[str, float]]:
        """HIP-event timings of the MFMA GEMM kernels at minibatch size n (bench.py roofline block)."""
        ms = (C.c_float * 6)()
        fl = (C.c_double * 6)()
        native.check(self.L.kp1_mlp_time_kernels(self._h, _p(obs), obs.shape[-1], n, iters, C.cast(ms, C.c_void_p), C.cast(fl, C.c_void_p), self._stream()))
        names = ("gemm_nt_fwd_l2", "gemm_nt_bwd_dz1", "gemm_tn_dw2", "gemm_nt_fwd_l1", "mlp_train_tile", "gemm_tn_split")
        return {nm: {"ms": float(ms[i]), "flops": float(fl[i]), "tflops": float(fl[i]) / (float(ms[i]) * 1e-3) / 1e12}
                for i, nm in enumerate(names) if ms[i] > 0}
