#!/usr/bin/env python3
"""Golden vectors for compute_approach_reward / compute_dock_reward under RANDOMISED configurations, generated
by importing the reference.  Runs ONLY in the build container.

The step traces in this directory exercise the reward functions with the YAML configs the reference ships, where many weights are 0.
Here every field of ApproachRewardConfig / DockRewardConfig is drawn at random (thresholds in ranges where the zone tests flip), the
poses / flags / counters are random, and the reference's own functions produce the total and every component.  The five hand-written
cases of the reference's tests/test_kinematic_phase1_approach_reward.py are appended with the constants those tests assert.

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden_reward_fuzz.py
"""
from __future__ import annotations

import dataclasses
import json
import sys
from pathlib import Path

import numpy as np

REF = Path("/root/reference/hrl_ws/src/hrl_trainer")
sys.dont_write_bytecode = True
sys.path.insert(0, str(REF))

from hrl_trainer.kinematic_phase1.envs.reward_approach import ApproachRewardConfig, compute_approach_reward  # noqa: E402
from hrl_trainer.kinematic_phase1.envs.reward_dock import DockRewardConfig, compute_dock_reward  # noqa: E402

OUT = Path(__file__).resolve().parent


def draw_field(rng: np.random.Generator, name: str, default):
    if isinstance(default, bool):
        return bool(rng.random() < 0.5)
    if isinstance(default, int):
        return int(rng.integers(0, 6))
    if isinstance(default, tuple):
        return default
    if rng.random() < 0.15:
        return float(default)
    if name.endswith("_m") or "radius_m" in name or "tolerance_m" in name:
        return float(rng.uniform(0.0, 0.15))
    if name.endswith("_rad"):
        return float(rng.uniform(0.0, 0.7))
    if name.endswith("_power"):
        return float(rng.uniform(1.0, 3.0))
    if name.endswith("_threshold"):
        return float(rng.uniform(0.0, 1.2))
    if "multiplier" in name or name.endswith("_scale") or name.endswith("_decay"):
        return float(rng.uniform(0.3, 2.0))
    return float(rng.uniform(0.0, 2.0))


def draw_config(rng: np.random.Generator, cls):
    values = {}
    for f in dataclasses.fields(cls):
        values[f.name] = draw_field(rng, f.name, f.default)
    if cls is ApproachRewardConfig:
        n = int(rng.integers(0, 4))
        thr = sorted((float(x) for x in rng.uniform(0.05, 1.0, size=n)), reverse=True)
        values["orientation_milestone_thresholds_rad"] = tuple(thr)
        values["orientation_milestone_bonuses"] = tuple(float(x) for x in rng.uniform(0.0, 0.5, size=n))
    return values


def draw_inputs(rng: np.random.Generator) -> dict:
    goal = np.concatenate([rng.uniform(-0.5, 0.5, 3), rng.uniform(-np.pi, np.pi, 3)])

    def near(scale_p, scale_o):
        return goal + np.concatenate([rng.normal(0, scale_p, 3), rng.normal(0, scale_o, 3)])

    tight = rng.random() < 0.4      # all four readiness tests (position, orientation, action, dq) pass together only close to the goal
    sp = float(10 ** (rng.uniform(-3.2, -1.8) if tight else rng.uniform(-3.2, -0.6)))
    so = float(10 ** (rng.uniform(-2.5, -1.0) if tight else rng.uniform(-2.5, 0.3)))
    prev, curr = near(sp, so), near(sp, so)
    cross = rng.random()
    if tight and cross < 0.2:        # entering a zone: previous pose well outside
        prev = near(12 * sp, 8 * so)
    elif tight and cross < 0.4:      # leaving a zone
        curr = near(12 * sp, 8 * so)
    if rng.random() < 0.25:      # small step between the two poses (progress terms with the same sign structure as a real step)
        curr = prev + np.concatenate([rng.normal(0, 0.2 * sp, 3), rng.normal(0, 0.2 * so, 3)])
    a_scale = 1.0 if (rng.random() < 0.6 and not tight) else float(10 ** rng.uniform(-2.5, -0.5))   # small actions: low-motion / tiny-correction branches
    dq_hi = 0.02 if tight else 0.2
    return {
        "prev_pose6": prev.tolist(), "curr_pose6": curr.tolist(), "goal_pose6": goal.tolist(),
        "action": (a_scale * rng.uniform(-1, 1, 7)).tolist(), "prev_action": (a_scale * rng.uniform(-1, 1, 7)).tolist(),
        "curr_in_pre_near_goal": bool(rng.random() < 0.6), "prev_in_near_goal": bool(rng.random() < 0.5), "curr_in_near_goal": bool(rng.random() < 0.5),
        "dwell_count": int(rng.integers(0, 8)), "near_goal_entry_count": int(rng.integers(0, 5)), "near_goal_drift_count": int(rng.integers(0, 7)),
        "joint_limit_margin_min": float(rng.uniform(0.0, 1.0)), "success": bool(rng.random() < 0.2),
        "dq_norm": float(rng.uniform(0, dq_hi)), "prev_dq_norm": float(rng.uniform(0, dq_hi)), "delta_q_change_l2": float(rng.uniform(0, 0.2)),
        "entry_pos_error_norm": float(rng.uniform(0, 0.1)), "entry_ori_error_norm": float(rng.uniform(0, 0.5)),
        "entry_action_l2": float(rng.uniform(0, 2.0)), "entry_dq_norm": float(rng.uniform(0, 0.2)),
    }


APPROACH_ARGS = ("prev_pose6", "curr_pose6", "goal_pose6", "action", "prev_action", "curr_in_pre_near_goal", "prev_in_near_goal", "curr_in_near_goal", "dwell_count",
                 "joint_limit_margin_min", "success", "near_goal_entry_count", "near_goal_drift_count", "dq_norm", "prev_dq_norm")
DOCK_ARGS = ("prev_pose6", "curr_pose6", "goal_pose6", "action", "prev_action", "prev_in_near_goal", "curr_in_near_goal", "dwell_count", "joint_limit_margin_min",
             "success", "near_goal_entry_count", "near_goal_drift_count", "delta_q_change_l2", "dq_norm", "entry_pos_error_norm", "entry_ori_error_norm",
             "entry_action_l2", "entry_dq_norm")


def run_case(mode: str, cfg: dict, inp: dict) -> dict:
    if mode == "approach":
        total, comps = compute_approach_reward(config=ApproachRewardConfig(**cfg), **{k: inp[k] for k in APPROACH_ARGS})
    else:
        total, comps = compute_dock_reward(config=DockRewardConfig(**cfg), **{k: inp[k] for k in DOCK_ARGS})
    return {"mode": mode, "config": {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}, "inputs": inp, "reward": float(total),
            "components": {k: float(v) for k, v in comps.items()}}


def reference_unit_cases() -> list[dict]:
    """tests/test_kinematic_phase1_approach_reward.py: the arguments of its five tests and the constants they assert."""
    z7 = [0.0] * 7
    base = {"action": z7, "prev_action": z7, "joint_limit_margin_min": 1.0, "success": False, "near_goal_entry_count": 0, "near_goal_drift_count": 0, "dq_norm": 0.0,
            "prev_dq_norm": 0.0, "delta_q_change_l2": 0.0, "entry_pos_error_norm": 0.0, "entry_ori_error_norm": 0.0, "entry_action_l2": 0.0, "entry_dq_norm": 0.0}
    dflt = {f.name: f.default for f in dataclasses.fields(ApproachRewardConfig)}
    out = []

    def add(name, cfg_over, expect, **inp):
        case = run_case("approach", {**dflt, **cfg_over}, {**base, **inp})
        case["name"] = name
        case["expect"] = expect
        for k, v in expect.items():
            assert case["components"][k] == v, (name, k, case["components"][k], v)
        out.append(case)

    g0 = [0.0] * 6
    add("reentry_bonus_first", {"near_goal_bonus": 1.0, "near_goal_bonus_decay": 0.5}, {"near_goal_bonus": 1.0}, goal_pose6=g0, prev_pose6=[0.04, 0, 0, 0, 0, 0],
        curr_pose6=[0.02, 0, 0, 0, 0, 0], curr_in_pre_near_goal=True, prev_in_near_goal=False, curr_in_near_goal=True, dwell_count=1, near_goal_entry_count=1)
    add("reentry_bonus_second", {"near_goal_bonus": 1.0, "near_goal_bonus_decay": 0.5}, {"near_goal_bonus": 0.5}, goal_pose6=g0, prev_pose6=[0.04, 0, 0, 0, 0, 0],
        curr_pose6=[0.02, 0, 0, 0, 0, 0], curr_in_pre_near_goal=True, prev_in_near_goal=False, curr_in_near_goal=True, dwell_count=1, near_goal_entry_count=2)
    add("leave_penalty", {"near_goal_leave_penalty": 0.35}, {"near_goal_leave_penalty": -0.35}, goal_pose6=g0, prev_pose6=[0.02, 0, 0, 0, 0, 0],
        curr_pose6=[0.04, 0, 0, 0, 0, 0], curr_in_pre_near_goal=True, prev_in_near_goal=True, curr_in_near_goal=False, dwell_count=0, near_goal_entry_count=1,
        near_goal_drift_count=1)
    gp = [0.10, 0, 0, 0, 0, 0]
    for tag, pre in (("far", False), ("near", True)):
        add(f"near_field_orientation_{tag}", {"orientation_progress_weight": 1.0, "near_field_orientation_progress_weight": 2.0}, {}, goal_pose6=gp,
            prev_pose6=[0, 0, 0, 0.30, 0, 0], curr_pose6=[0, 0, 0, 0.20, 0, 0], curr_in_pre_near_goal=pre, prev_in_near_goal=False, curr_in_near_goal=False, dwell_count=0)
    assert out[-1]["components"]["orientation_progress"] > out[-2]["components"]["orientation_progress"]
    for tag, roll, positive in (("good", 0.10, True), ("bad", 0.60, False)):
        add(f"coarse_orientation_bonus_{tag}", {"coarse_orientation_bonus": 0.05, "coarse_orientation_bonus_threshold_rad": 0.35}, {} if positive else {"coarse_orientation_bonus": 0.0},
            goal_pose6=gp, prev_pose6=[0, 0, 0, 0.40, 0, 0], curr_pose6=[0, 0, 0, roll, 0, 0], curr_in_pre_near_goal=True, prev_in_near_goal=False, curr_in_near_goal=False,
            dwell_count=0)
    assert out[-2]["components"]["coarse_orientation_bonus"] > 0.0
    for tag, drift in (("early", 1), ("late", 4)):
        add(f"drift_escalation_{tag}", {"drift_penalty_weight": 2.0, "drift_penalty_escalation_start": 2, "drift_penalty_escalation_per_count": 0.5}, {}, goal_pose6=g0,
            prev_pose6=[0.02, 0, 0, 0, 0, 0], curr_pose6=[0.025, 0, 0, 0, 0, 0], curr_in_pre_near_goal=True, prev_in_near_goal=True, curr_in_near_goal=True, dwell_count=2,
            near_goal_entry_count=1, near_goal_drift_count=drift)
    assert out[-1]["components"]["drift_penalty"] < out[-2]["components"]["drift_penalty"]
    assert out[-1]["components"]["drift_penalty_scale"] > out[-2]["components"]["drift_penalty_scale"]
    return out


def main() -> None:
    rng = np.random.default_rng(20260518)
    cases = []
    for mode, cls in (("approach", ApproachRewardConfig), ("dock", DockRewardConfig)):
        for _ in range(120):
            cases.append(run_case(mode, draw_config(rng, cls), draw_inputs(rng)))
    # targeted dock cases: inside the near-strict zone but outside the tight pose, improving, with a small action (tiny-correction branch)
    for _ in range(10):
        cfg = draw_config(rng, DockRewardConfig)
        cfg.update({"tight_pose_pos_threshold_m": 0.002, "tight_pose_ori_threshold_rad": 0.02, "near_strict_pos_threshold_m": 0.05,
                    "near_strict_ori_threshold_rad": 0.4, "tiny_correction_bonus": float(rng.uniform(0.1, 1.0)),
                    "tiny_correction_action_threshold": float(rng.choice([0.0, 0.5]))})
        inp = draw_inputs(rng)
        goal = np.asarray(inp["goal_pose6"])
        d = np.concatenate([rng.normal(0, 0.008, 3), rng.normal(0, 0.06, 3)])
        inp["prev_pose6"] = (goal + d).tolist()
        inp["curr_pose6"] = (goal + float(rng.uniform(0.5, 0.95)) * d).tolist()
        inp["action"] = (0.05 * rng.uniform(-1, 1, 7)).tolist()
        cases.append(run_case("dock", cfg, inp))
    unit = reference_unit_cases()
    (OUT / "reward_fuzz.json").write_text(json.dumps({"cases": cases, "reference_unit_cases": unit}, separators=(",", ":")))
    nz = {}
    for c in cases:
        for k, v in c["components"].items():
            if v != 0.0:
                nz[(c["mode"], k)] = nz.get((c["mode"], k), 0) + 1
    names = {(c["mode"], k) for c in cases for k in c["components"]}
    print("cases", len(cases), "unit", len(unit), "components never non-zero:", sorted(names - set(nz)))


if __name__ == "__main__":
    main()
