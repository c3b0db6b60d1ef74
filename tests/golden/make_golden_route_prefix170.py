#!/usr/bin/env python3
"""Golden fixtures for BASELINE configs[4]: the route curriculum at the 170-waypoint prefix (SURVEY.md 8d "Config 5").

Runs ONLY in the build container (imports the reference from /root/reference through make_golden_route.py's helpers); tests read the
two files it writes.  The config is the reference's route_curriculum_prefix120_routeobs_sequence2 block with the route window moved
to 170 exactly as ``set_route_window(max_route_index=170)`` moves it (route/route_env.py:99-120: only min / max route index change,
the segment window 81..120 stays), on the committed synthetic 484-waypoint route.

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden_route_prefix170.py
"""
from __future__ import annotations

import copy
import json
import sys
from pathlib import Path

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, str(Path(__file__).resolve().parent))

import make_golden_route as mg  # noqa: E402  (puts the reference on sys.path)

from hrl_trainer.kinematic_phase1.route.route_dataset import load_route_dataset  # noqa: E402
from hrl_trainer.kinematic_phase1.route.route_reset_samplers import RouteResetSamplerConfig, sample_route_reset  # noqa: E402
from hrl_trainer.kinematic_phase1.route.route_sequence_env import RouteSequenceConfig, RouteSequenceKinematicEnv  # noqa: E402
from hrl_trainer.kinematic_phase1.training.policy_config import to_env_config  # noqa: E402

OUT = mg.OUT
PREFIX = 170


def main() -> None:
    route = load_route_dataset(OUT / "synthetic_route.json")
    cfg = copy.deepcopy(mg.route_cfgs()["route_curriculum_prefix120_routeobs_sequence2"])
    cfg["route"]["reset"]["max_route_index"] = PREFIX
    (OUT / "configs" / "route_curriculum_prefix170_routeobs_sequence2.json").write_text(json.dumps(cfg, indent=1, sort_keys=True))

    reset_cfg = dict(cfg["route"]["reset"])
    rc = RouteResetSamplerConfig(**reset_cfg)
    specs = to_env_config(cfg).joint_specs
    rng = np.random.default_rng(817)
    rec = {k: [] for k in ("initial_q", "initial_dq", "initial_prev_action", "goal_q", "route_index", "start_index", "mode", "rng_before", "rng_after")}
    for _ in range(256):
        rec["rng_before"].append(mg.rng_words(rng))
        s = sample_route_reset(rng=rng, route=route, joint_specs=specs, config=rc)
        rec["rng_after"].append(mg.rng_words(rng))
        for k, v in (("initial_q", s.initial_q), ("initial_dq", s.initial_dq), ("initial_prev_action", s.initial_prev_action), ("goal_q", s.goal_q),
                     ("route_index", s.route_index), ("start_index", s.start_route_index), ("mode", mg.MODES.index(s.reset_mode))):
            rec[k].append(v)
    np.savez_compressed(OUT / f"route_resets_prefix{PREFIX}.npz", **{k: np.array(v) for k, v in rec.items()}, reset_config=json.dumps(reset_cfg),
                        config="route_curriculum_prefix170_routeobs_sequence2", seed=817)
    print("resets: route_index range", int(np.min(rec["route_index"])), int(np.max(rec["route_index"])))

    env = RouteSequenceKinematicEnv(route=route, config=mg.env_config(cfg, PREFIX), sequence_config=RouteSequenceConfig(**cfg["route"]["sequence"]))
    mg.trace(f"seq_prefix{PREFIX}", env, route, seed=1709, steps=1400, kinds=["servo", "servo_noisy", "random", "servo", "zero"], sequence=True)
    g = np.load(OUT / f"route_trace_seq_prefix{PREFIX}.npz")
    print("trace: route_index range", int(g["route_index"].min()), int(g["route_index"].max()), "episodes", len(g["reset_at"]))


if __name__ == "__main__":
    main()
