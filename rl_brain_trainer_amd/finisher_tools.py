"""Finisher-side tooling on the device engine: the dock reverse-curriculum callback logic and the handoff-state
buffer builder, so the Finisher can be (re)trained from the engine's own Approach policy.

The reverse curriculum (kinematic_phase1/training/callbacks.py:104-212, ``DockReverseCurriculumCallback``) is a one-wave device tracker
(include/kp1_ppo.h, kp1_dock_curriculum_*; csrc/kp1_dock_curriculum.inc): ``DockReverseCurriculum`` below is its host handle and resolves
the YAML stages into the values each stage leaves in the env.  The handoff-state buffer builder
(kinematic_phase1/training/build_finisher_handoff_state_buffer.py:19-143) keeps the JSON layout of ``finisher_handoff_state_buffer.json``
(what ``dock_reset.handoff_state_buffer_path`` points at, reset_samplers.py:131-165) and runs all suite episodes in ONE vectorised
Approach env (``evaluate.run_episodes``).
"""
from __future__ import annotations

import ctypes as C
import json
from pathlib import Path
from typing import Any

import numpy as np
import torch

from . import config as kcfg

# keys a reverse-curriculum stage may override (callbacks.py:131-150): three env scalars, the dock_reset block
_STAGE_ENV_KEYS = ("action_delta_scale", "dock_residual_action_limit", "dock_delta_q_change_limit_scale")
_STAGE_RESET_SCALARS = ("close_bucket_probability", "close_bucket_min_pos_error_m", "close_bucket_max_pos_error_m", "close_bucket_max_ori_error_rad",
                        "handoff_state_probability")
_STAGE_RESET_VECTORS = ("close_init_q_noise", "init_q_noise")
# a stage may also re-filter (or replace) the handoff-state buffer (reset_samplers.py:131-165): the filtered list of every stage is built
# once, when the tracker is attached, and the device buffer holds them back to back; a promotion then only switches (offset, count)
_STAGE_BUFFER_KEYS = ("handoff_state_buffer_path", "handoff_state_max_position_error_m", "handoff_state_max_orientation_error_rad", "handoff_state_max_action_l2")
_BUFFER_DEFAULTS = {"handoff_state_buffer_path": "", "handoff_state_max_position_error_m": 1.0, "handoff_state_max_orientation_error_rad": 10.0,
                    "handoff_state_max_action_l2": 10.0}   # DockResetConfig defaults, reset_samplers.py:109-113
MAX_STAGES, MAX_WINDOW, MAX_HISTORY = 16, 1024, 32


class _Stage(C.Structure):
    _fields_ = [(k, C.c_double) for k in _STAGE_ENV_KEYS + _STAGE_RESET_SCALARS] + [(k, C.c_double * kcfg.NJ) for k in _STAGE_RESET_VECTORS] + \
               [("success_rate_threshold", C.c_double), ("min_episodes", C.c_int32), ("window_episodes", C.c_int32),
                ("handoff_offset", C.c_int32), ("handoff_count", C.c_int32)]


class _Event(C.Structure):
    _fields_ = [("total_timesteps", C.c_int64), ("from_stage", C.c_int32), ("to_stage", C.c_int32), ("stage_episode_count", C.c_int32), ("reserved0", C.c_int32),
                ("trigger_success_rate", C.c_double)]


class _State(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("stage_index", "stage_episode_count", "ring_len", "ring_head", "window_episodes", "n_stages", "n_events", "reserved0")] + \
               [("num_timesteps", C.c_int64), ("stages", _Stage * MAX_STAGES), ("events", _Event * MAX_HISTORY), ("ring", C.c_uint8 * MAX_WINDOW)]


class DockReverseCurriculum:
    """The Finisher's reverse curriculum (``training.dock_reverse_curriculum`` of the YAML: ``stages`` + ``window_episodes``) as a device tracker.
    ``PPO(curriculum=...)`` calls ``attach(env)`` once and ``observe(done_bytes, steps)`` after every env step on the rollout stream, so the
    dock rollout is one hipGraph replay; ``summary()`` has the reference callback's keys."""

    def __init__(self, *, stages: list[dict[str, object]], window_episodes: int, handoff_base_dirs: tuple[Path, ...] = ()) -> None:
        if not stages:
            raise ValueError("DockReverseCurriculumCallback requires at least one stage")
        if len(stages) > MAX_STAGES or int(window_episodes) > MAX_WINDOW:
            raise ValueError(f"the device tracker holds at most {MAX_STAGES} stages and a window of {MAX_WINDOW} episodes")
        self.stages = [dict(s) for s in stages]
        self.window_episodes = max(int(window_episodes), 1)
        self.handoff_base_dirs = tuple(handoff_base_dirs)
        self.env: Any = None
        self._st = C.c_void_p()

    # ---------------------------------------------------------------- host side: what each stage leaves in the env
    def stage_payload(self, index: int) -> dict[str, object]:
        """the ``apply_dock_training_stage`` argument of stage ``index``: only the keys the stage names"""
        stage = self.stages[index]
        payload: dict[str, object] = {key: stage[key] for key in _STAGE_ENV_KEYS if key in stage}
        payload["dock_reset"] = {key: stage[key] for key in _STAGE_RESET_SCALARS + _STAGE_RESET_VECTORS + _STAGE_BUFFER_KEYS if key in stage}
        return payload

    def stage_buffers(self, config: kcfg.EnvConfig) -> list[list[dict[str, list[float]]]] | None:
        """per stage, the handoff states its dock resets may draw from -- or None when no stage touches the buffer keys (the env keeps the
        buffer its config was built with).  The settings accumulate over the stages like every other override."""
        if not any(key in stage for stage in self.stages for key in _STAGE_BUFFER_KEYS):
            return None
        base = dict(((config.source or {}).get("env", {}) or {}).get("dock_reset", {}) or {})
        live = {k: base.get(k, default) for k, default in _BUFFER_DEFAULTS.items()}
        out = []
        for stage in self.stages:
            live.update({k: stage[k] for k in _STAGE_BUFFER_KEYS if k in stage})
            out.append(kcfg.load_handoff_states(str(live["handoff_state_buffer_path"] or ""), max_position_error_m=float(live["handoff_state_max_position_error_m"]),
                                                max_orientation_error_rad=float(live["handoff_state_max_orientation_error_rad"]),
                                                max_action_l2=float(live["handoff_state_max_action_l2"]), base_dirs=self.handoff_base_dirs))
        return out

    def resolved_stages(self, config: kcfg.EnvConfig, buffers: list[list[Any]] | None = None):
        """per stage, the full set of overridable values the env holds after stages 0..k were applied in order to ``config``; ``buffers``
        (stage_buffers) lays the stages' handoff lists back to back and records each stage's slice"""
        c = config.c
        live: dict[str, Any] = {k: float(getattr(c.env, k)) for k in _STAGE_ENV_KEYS}
        live.update({k: float(getattr(c.dock_reset, k)) for k in _STAGE_RESET_SCALARS})
        live.update({k: [float(v) for v in getattr(c.dock_reset, k)[:]] for k in _STAGE_RESET_VECTORS})
        out = (_Stage * len(self.stages))()
        for k, stage in enumerate(self.stages):
            for key in live:
                if key in stage:
                    live[key] = [float(v) for v in stage[key]] if key in _STAGE_RESET_VECTORS else float(stage[key])
            for key, value in live.items():
                if key in _STAGE_RESET_VECTORS:
                    getattr(out[k], key)[:] = value
                else:
                    setattr(out[k], key, value)
            out[k].success_rate_threshold = float(stage.get("success_rate_threshold", 1.0))
            out[k].min_episodes = max(int(stage.get("min_episodes", self.window_episodes)), 1)
            out[k].window_episodes = max(int(stage.get("window_episodes", self.window_episodes)), 1)
            out[k].handoff_offset, out[k].handoff_count = 0, -1
            if buffers is not None:
                out[k].handoff_offset, out[k].handoff_count = sum(len(b) for b in buffers[:k]), len(buffers[k])
        return out

    # ---------------------------------------------------------------- device side
    def attach(self, env: Any) -> None:
        """_on_training_start: allocate the tracker next to the env; stage 0 is applied on the device before the first reset"""
        from . import native

        L = env.L
        vp, i32 = C.c_void_p, C.c_int32
        L.kp1_dock_curriculum_create.argtypes = [vp, C.POINTER(_Stage), i32, i32, C.POINTER(vp)]
        L.kp1_dock_curriculum_destroy.argtypes = [vp, vp]
        L.kp1_dock_curriculum_observe.argtypes = [vp, vp, vp, i32, i32, i32, vp]
        L.kp1_dock_curriculum_read.argtypes = [vp, vp, C.POINTER(_State), vp]
        self.env = env
        buffers = self.stage_buffers(env.config)
        if buffers is not None:
            env.set_handoff_states([state for stage_states in buffers for state in stage_states])
        table = self.resolved_stages(env.config, buffers)
        with torch.cuda.device(env.device):
            native.check(L.kp1_dock_curriculum_create(env._handle, table, len(self.stages), self.window_episodes, C.byref(self._st)))
        env.launch_args_version = getattr(env, "launch_args_version", 0) + 1

    on_training_start = attach

    def observe(self, dones: torch.Tensor, steps_per_call: int) -> None:
        self.observe_chunk(dones, int(dones.numel()), 1, 1)

    def observe_chunk(self, dones_all: torch.Tensor, n_local: int, chunk_steps: int, world: int) -> None:
        from . import native

        stream = torch.cuda.current_stream(self.env.device).cuda_stream
        native.check(self.env.L.kp1_dock_curriculum_observe(self.env._handle, self._st, C.c_void_p(dones_all.data_ptr()), int(n_local), int(chunk_steps),
                                                            int(world), C.c_void_p(stream)))

    def read(self) -> _State:
        from . import native

        out = _State()
        stream = torch.cuda.current_stream(self.env.device).cuda_stream
        native.check(self.env.L.kp1_dock_curriculum_read(self.env._handle, self._st, C.byref(out), C.c_void_p(stream)))
        # the Python-side config mirror follows the stage the device applied (the C handle's mirror was updated by the call above)
        live, c = out.stages[out.stage_index], self.env.config.c
        for key in _STAGE_ENV_KEYS:
            setattr(c.env, key, float(getattr(live, key)))
        for key in _STAGE_RESET_SCALARS:
            setattr(c.dock_reset, key, float(getattr(live, key)))
        for key in _STAGE_RESET_VECTORS:
            getattr(c.dock_reset, key)[:] = list(getattr(live, key)[:])
        return out

    @property
    def current_stage_index(self) -> int:
        return int(self.read().stage_index) if self.env is not None else 0

    def _name(self, index: int) -> object:
        return self.stages[index].get("name", f"stage_{index}")

    def summary(self) -> dict[str, object]:
        st = self.read()
        n, cap, head = int(st.ring_len), int(st.window_episodes), int(st.ring_head)
        wins = sum(int(st.ring[(head + j) % cap]) for j in range(n))
        events = [st.events[k] for k in range(min(int(st.n_events), MAX_HISTORY))]
        return {"stage_index": int(st.stage_index), "stage_name": self._name(int(st.stage_index)), "stage_episode_count": int(st.stage_episode_count),
                "recent_success_rate": float(wins) / float(n) if n else 0.0,
                "history": [{"from_stage_index": int(e.from_stage), "from_stage_name": self._name(int(e.from_stage)), "to_stage_index": int(e.to_stage),
                             "to_stage_name": self._name(int(e.to_stage)), "trigger_success_rate": float(e.trigger_success_rate),
                             "stage_episode_count": int(e.stage_episode_count), "total_timesteps": int(e.total_timesteps)} for e in events]}

    def close(self) -> None:
        if self.env is not None and self._st.value:
            self.env.L.kp1_dock_curriculum_destroy(self.env._handle, self._st)
            self._st = C.c_void_p()


def build_finisher_handoff_state_buffer(*, approach_policy, approach_cfg: kcfg.EnvConfig, artifact_root: str | Path | None = None, episodes: int = 500,
                                        seed: int = 700001, stage_index: int = 0, handoff_confirm_steps: int = 2, handoff_mode: str = "final_settled",
                                        source_checkpoint_name: str = "engine", device: int = 0, obs_stride: int = 56) -> dict[str, Any]:
    """build_finisher_handoff_state_buffer.main with the policy passed as a callable; curriculum-region suites only (every shipped
    approach config enables the curriculum)."""
    import torch

    from . import evaluate as ev
    from .vec_env import ArmKinematicVecEnv

    if handoff_mode not in ("final_settled", "first_confirmed", "final_always"):
        raise ValueError(f"unknown handoff mode {handoff_mode}")
    suite = ev.build_curriculum_local_eval_suite(approach_cfg, seed=seed, stage_index=stage_index, n_episodes=episodes, device=device)
    r = approach_cfg.c.reward
    env = ArmKinematicVecEnv(approach_cfg, episodes, device=device, seed=seed)
    env.set_curriculum_stage(stage_index)
    if obs_stride != 56:
        env.set_obs_stride(obs_stride)
    res, hand = ev.run_episodes(env, approach_policy, {"initial_q": suite["initial_q"], "goal_q": suite["goal_q"], "goal_pose6": suite["goal_pose6"],
                                                       "policy_mode": "approach"}, ready_cfg=r, handoff_confirm_steps=handoff_confirm_steps)
    env.close()
    final_ready = ev.finisher_ready(res["final_position_error"], res["final_orientation_error"], res["final_action_magnitude"], res["final_dq_norm"], r)
    if handoff_mode == "final_settled":
        take, src = final_ready, res
    elif handoff_mode == "first_confirmed":
        take, src = hand["valid"], {k: hand.get(k, torch.zeros_like(v)) for k, v in res.items()}
    else:
        take, src = torch.ones_like(final_ready), res

    def cpu(t):
        return t.detach().cpu().numpy()

    take_h, fr = cpu(take), cpu(final_ready)
    S = {k: cpu(src[k]) for k in ("state_q", "state_dq", "state_prev_action", "state_goal_q", "state_goal_pose6", "final_position_error",
                                  "final_orientation_error", "final_action_magnitude", "final_dq_norm", "step_count")}
    A = {k: cpu(res[k]) for k in ("final_position_error", "final_orientation_error", "final_action_magnitude", "final_dq_norm")}
    states, summaries = [], []
    for e in range(episodes):
        if take_h[e]:
            states.append({
                "episode_id": e, "step_index": int(S["step_count"][e]), "initial_q": S["state_q"][e].tolist(), "initial_dq": S["state_dq"][e].tolist(),
                "initial_prev_action": S["state_prev_action"][e].tolist(), "goal_q": S["state_goal_q"][e].tolist(), "goal_pose6": S["state_goal_pose6"][e].tolist(),
                "position_error_norm": float(S["final_position_error"][e]), "orientation_error_norm": float(S["final_orientation_error"][e]),
                "dwell_count": int(approach_cfg.c.env.dwell_steps_target), "action_l2": float(S["final_action_magnitude"][e]),
                "dq_norm": float(S["final_dq_norm"][e]), "source_checkpoint_name": source_checkpoint_name, "handoff_mode": handoff_mode})
        summaries.append({"episode_id": e, "stored_handoff": bool(take_h[e]), "final_ready": bool(fr[e]), "final_position_error": float(A["final_position_error"][e]),
                          "final_orientation_error": float(A["final_orientation_error"][e]), "final_action_magnitude": float(A["final_action_magnitude"][e]),
                          "final_dq_norm": float(A["final_dq_norm"][e])})

    def mean_of(key):
        vals = [float(s[key]) for s in states]
        return float(np.mean(vals)) if vals else None

    summary = {
        "handoff_mode": handoff_mode, "eval_scope": "curriculum_region", "episode_count": episodes, "stored_handoff_count": len(states),
        "stored_handoff_rate": float(len(states) / episodes) if episodes else 0.0, "mean_position_error": mean_of("position_error_norm"),
        "mean_orientation_error": mean_of("orientation_error_norm"), "mean_action_l2": mean_of("action_l2"), "mean_dq_norm": mean_of("dq_norm"),
        "states": states, "episode_summaries": summaries,
    }
    if artifact_root is not None:
        root = Path(artifact_root)
        root.mkdir(parents=True, exist_ok=True)
        (root / "finisher_handoff_state_buffer.json").write_text(json.dumps(summary, indent=2))
        (root / "finisher_handoff_state_buffer_suite.json").write_text(json.dumps({"suite": [
            {"episode_id": e, "initial_q": suite["initial_q"][e].tolist(), "goal_q": suite["goal_q"][e].tolist(), "goal_pose6": suite["goal_pose6"][e].tolist()}
            for e in range(episodes)]}, indent=2))
    return summary
