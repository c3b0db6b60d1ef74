#!/usr/bin/env python3
"""Golden trace of the reference's DockReverseCurriculumCallback (training/callbacks.py:104-212), driven without SB3 (the base class
degrades to ``object``; an instance is built without __init__).  Runs ONLY in the build container.

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden_dock_curriculum.py
"""
from __future__ import annotations

import json
import sys
from collections import deque
from pathlib import Path

import numpy as np

REF = Path("/root/reference/hrl_ws/src/hrl_trainer")
sys.dont_write_bytecode = True
sys.path.insert(0, str(REF))

from hrl_trainer.kinematic_phase1.training.callbacks import DockReverseCurriculumCallback  # noqa: E402

OUT = Path(__file__).resolve().parent

STAGES = [
    {"name": "close", "dock_residual_action_limit": 0.35, "close_bucket_probability": 1.0, "close_bucket_max_pos_error_m": 0.004, "init_q_noise": [0.001] * 7,
     "min_episodes": 12, "success_rate_threshold": 0.7, "window_episodes": 10},
    {"name": "mid", "action_delta_scale": 0.012, "close_bucket_probability": 0.5, "handoff_state_probability": 0.5, "min_episodes": 20,
     "success_rate_threshold": 0.6},
    {"name": "wide", "dock_delta_q_change_limit_scale": 0.5, "close_bucket_probability": 0.1, "handoff_state_max_action_l2": 0.4},
]


class FakeVecEnv:
    def __init__(self):
        self.calls = []

    def env_method(self, name, payload):
        self.calls.append([name, payload])


def trace(seed: int, p: float) -> dict:
    cb = object.__new__(DockReverseCurriculumCallback)
    cb.stages = list(STAGES)
    cb.window_episodes = 16
    cb.current_stage_index = 0
    cb.stage_episode_count = 0
    cb.recent_successes = deque(maxlen=16)
    cb.history = []
    cb.training_env = FakeVecEnv()
    cb.num_timesteps = 0
    cb._on_training_start()
    rng = np.random.default_rng(seed)
    steps = []
    for _ in range(120):
        n = 5
        dones = (rng.random(n) < 0.3).tolist()
        infos = [{"success": bool(rng.random() < p)} for _ in range(n)]
        cb.num_timesteps += n
        cb.locals = {"infos": infos, "dones": dones}
        cb._on_step()
        steps.append({"dones": dones, "success": [i["success"] for i in infos], "stage": cb.current_stage_index, "count": cb.stage_episode_count})
    return {"steps": steps, "summary": cb.summary(), "calls": cb.training_env.calls}


def main() -> None:
    payload = {"stages": STAGES, "window_episodes": 16, "traces": [trace(1, 0.85), trace(2, 0.4)]}
    (OUT / "dock_reverse_curriculum.json").write_text(json.dumps(payload))
    for t in payload["traces"]:
        print("final stage", t["summary"]["stage_index"], "promotions", len(t["summary"]["history"]), "env calls", len(t["calls"]))


if __name__ == "__main__":
    main()
