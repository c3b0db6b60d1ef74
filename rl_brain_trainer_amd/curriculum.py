"""Device-resident PointCurriculumCallback (reference: kinematic_phase1/training/callbacks.py:32-101).

The reference callback runs on the host after every VecEnv step.  Here the same per-episode rule runs as a
one-wave HIP kernel on the rollout stream (include/kp1_ppo.h), so a 4096-env rollout never synchronises with
the host; the env kernel reads the published stage from device memory.  ``summary()`` mirrors the reference's.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import native

MAX_WINDOW = 1024
MAX_HISTORY = 64


class _Event(C.Structure):
    _fields_ = [("total_timesteps", C.c_int64), ("from_stage", C.c_int32), ("to_stage", C.c_int32), ("trigger_success_rate", C.c_double)]


class CurriculumState(C.Structure):
    _fields_ = [
        ("stage_index", C.c_int32), ("stage_episode_count", C.c_int32), ("ring_len", C.c_int32), ("ring_head", C.c_int32),
        ("window_episodes", C.c_int32), ("min_episodes_per_stage", C.c_int32), ("max_stage_index", C.c_int32), ("n_events", C.c_int32),
        ("success_rate_threshold", C.c_double), ("num_timesteps", C.c_int64),
        ("ring", C.c_int32 * MAX_WINDOW), ("events", _Event * MAX_HISTORY),
    ]


class PointCurriculum:
    def __init__(self, *, success_rate_threshold: float, window_episodes: int, min_episodes_per_stage: int, max_stage_index: int,
                 initial_stage_index: int = 0, device: torch.device | int = 0) -> None:
        self.L = native.load()
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        self._st = C.c_void_p()
        native.check(self.L.kp1_curriculum_create(self.device.index or 0, float(success_rate_threshold), int(window_episodes),
                                                  int(min_episodes_per_stage), int(max_stage_index), int(initial_stage_index), C.byref(self._st)))

    @property
    def stage_ptr(self) -> int:
        """device address of the int32 current stage (first word of the state)"""
        return int(self._st.value)

    def attach(self, env) -> None:
        """_on_training_start: the envs follow this tracker's stage from now on (callbacks.py:68-69)."""
        native.check(self.L.kp1_bind_stage_ptr(env._handle, C.c_void_p(self.stage_ptr)))

    def observe(self, dones: torch.Tensor, steps_per_call: int) -> None:
        stream = torch.cuda.current_stream(self.device).cuda_stream
        native.check(self.L.kp1_curriculum_observe(self.device.index or 0, self._st, C.c_void_p(dones.data_ptr()), int(dones.numel()),
                                                   int(steps_per_call), C.c_void_p(stream)))

    def observe_chunk(self, dones_all: torch.Tensor, n_local: int, chunk_steps: int, world: int) -> None:
        """data-parallel rollouts: the all-gathered [world, chunk_steps, n_local] done bytes of a chunk of env steps, replayed in the
        reference's order (step by step, global env id order inside a step)"""
        assert dones_all.numel() == world * chunk_steps * n_local and dones_all.dtype == torch.uint8 and dones_all.is_contiguous()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        native.check(self.L.kp1_curriculum_observe_chunk(self.device.index or 0, self._st, C.c_void_p(dones_all.data_ptr()), int(n_local), int(chunk_steps),
                                                         int(world), C.c_void_p(stream)))

    def read(self) -> CurriculumState:
        out = CurriculumState()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        native.check(self.L.kp1_curriculum_read(self.device.index or 0, self._st, C.byref(out), C.c_void_p(stream)))
        return out

    def summary(self) -> dict[str, object]:
        """callbacks.py:94-101"""
        st = self.read()
        n = st.ring_len
        recent = [st.ring[k] for k in range(n)]
        return {
            "stage_index": int(st.stage_index),
            "stage_episode_count": int(st.stage_episode_count),
            "recent_success_rate": float(sum(recent)) / float(n) if n else 0.0,
            "history": [
                {"from_stage_index": int(e.from_stage), "to_stage_index": int(e.to_stage),
                 "trigger_success_rate": float(e.trigger_success_rate), "total_timesteps": int(e.total_timesteps)}
                for e in list(st.events)[: min(st.n_events, MAX_HISTORY)]
            ],
        }

    def close(self) -> None:
        if self._st.value:
            self.L.kp1_curriculum_destroy(self.device.index or 0, self._st)
            self._st = C.c_void_p()
