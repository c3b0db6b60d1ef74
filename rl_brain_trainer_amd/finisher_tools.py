"""Finisher-side tooling on the device engine: the dock reverse-curriculum callback logic and the handoff-state
buffer builder, so the Finisher can be (re)trained from the engine's own Approach policy.

Mirror of kinematic_phase1/training/callbacks.py:104-212 (``DockReverseCurriculumCallback``) and
kinematic_phase1/training/build_finisher_handoff_state_buffer.py:19-143: same stage payload keys, promotion rule, JSON layout
of ``finisher_handoff_state_buffer.json`` (what ``dock_reset.handoff_state_buffer_path`` points at, reset_samplers.py:131-165).
All suite episodes of the buffer builder run in ONE vectorised Approach env (``evaluate.run_episodes``).
"""
from __future__ import annotations

import json
from collections import deque
from pathlib import Path
from typing import Any, Sequence

import numpy as np

from . import config as kcfg

_STAGE_ENV_KEYS = ("action_delta_scale", "dock_residual_action_limit", "dock_delta_q_change_limit_scale")
_STAGE_RESET_KEYS = ("close_bucket_probability", "close_bucket_min_pos_error_m", "close_bucket_max_pos_error_m", "close_bucket_max_ori_error_rad",
                     "close_init_q_noise", "init_q_noise", "handoff_state_probability", "handoff_state_buffer_path", "handoff_state_max_position_error_m",
                     "handoff_state_max_orientation_error_rad", "handoff_state_max_action_l2")


class DockReverseCurriculum:
    def __init__(self, *, stages: list[dict[str, object]], window_episodes: int) -> None:
        if not stages:
            raise ValueError("DockReverseCurriculumCallback requires at least one stage")
        self.stages = list(stages)
        self.window_episodes = max(int(window_episodes), 1)
        self.current_stage_index = 0
        self.stage_episode_count = 0
        self.recent_successes: deque[int] = deque(maxlen=self.window_episodes)
        self.history: list[dict[str, object]] = []
        self.num_timesteps = 0
        self.training_env: Any = None

    @staticmethod
    def stage_payload(stage: dict[str, object]) -> dict[str, object]:
        payload: dict[str, object] = {"dock_reset": {}}
        for key in _STAGE_ENV_KEYS:
            if key in stage:
                payload[key] = stage[key]
        for key in _STAGE_RESET_KEYS:
            if key in stage:
                payload["dock_reset"][key] = stage[key]
        return payload

    def _apply_stage(self, stage_index: int) -> None:
        self.training_env.env_method("apply_dock_training_stage", self.stage_payload(self.stages[stage_index]))

    def on_training_start(self, env: Any) -> None:
        self.training_env = env
        self._apply_stage(self.current_stage_index)

    def _promote(self, next_stage_index: int, trigger_success_rate: float) -> None:
        prev_stage, next_stage = self.stages[self.current_stage_index], self.stages[next_stage_index]
        self._apply_stage(next_stage_index)
        self.history.append({
            "from_stage_index": self.current_stage_index, "from_stage_name": prev_stage.get("name", f"stage_{self.current_stage_index}"),
            "to_stage_index": next_stage_index, "to_stage_name": next_stage.get("name", f"stage_{next_stage_index}"),
            "trigger_success_rate": float(trigger_success_rate), "stage_episode_count": int(self.stage_episode_count),
            "total_timesteps": int(self.num_timesteps)})
        self.current_stage_index = next_stage_index
        self.stage_episode_count = 0
        self.recent_successes.clear()

    def on_step(self, dones: Sequence[Any], success: Sequence[Any]) -> bool:
        """_on_step over one vectorised step (arrays in env order)."""
        self.num_timesteps += len(dones)
        for i, done in enumerate(dones):
            if not done:
                continue
            self.stage_episode_count += 1
            self.recent_successes.append(1 if bool(success[i]) else 0)
            if self.current_stage_index >= len(self.stages) - 1:
                continue
            stage = self.stages[self.current_stage_index]
            min_episodes = max(int(stage.get("min_episodes", self.window_episodes)), 1)
            threshold = float(stage.get("success_rate_threshold", 1.0))
            stage_window = max(int(stage.get("window_episodes", self.window_episodes)), 1)
            if self.stage_episode_count < min_episodes:
                continue
            if len(self.recent_successes) < min(stage_window, self.window_episodes):
                continue
            recent = list(self.recent_successes)[-min(stage_window, len(self.recent_successes)):]
            rate = float(sum(recent)) / float(len(recent))
            if rate >= threshold:
                self._promote(self.current_stage_index + 1, rate)
        return True

    def summary(self) -> dict[str, object]:
        rate = float(sum(self.recent_successes)) / float(len(self.recent_successes)) if self.recent_successes else 0.0
        return {"stage_index": self.current_stage_index, "stage_name": self.stages[self.current_stage_index].get("name", f"stage_{self.current_stage_index}"),
                "stage_episode_count": self.stage_episode_count, "recent_success_rate": rate, "history": list(self.history)}


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
