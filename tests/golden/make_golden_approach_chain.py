#!/usr/bin/env python3
"""Resolved configs of the Approach fine-tuning lineage behind the reference's published Stage 0-5 table
(docs/PHASE1_APPROACH_DOCK_CLOSEOUT.md:36-47: Approach config = approach_finisher_ready_v2_settle.yaml), i.e. the precision curriculum
3 cm -> 8 mm -> 2 cm / 0.18 rad dock-coarse-ready -> 5 mm / 0.05 rad finisher-ready that its YAMLs spell out.  Build container only (imports the reference's own YAML loader, as
make_golden.py does); writes tests/golden/configs/<name>.json, from which tools/make_builtin_configs.py emits the builtin YAML overlays.

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden_approach_chain.py
"""
from __future__ import annotations

import json
import sys
from pathlib import Path

REF = Path("/root/reference/hrl_ws/src/hrl_trainer")
sys.dont_write_bytecode = True
sys.path.insert(0, str(REF))

from hrl_trainer.kinematic_phase1.training.policy_config import (  # noqa: E402
    approach_default_config_path, config_dir, deep_merge, load_yaml_file, ppo_default_config_path, to_env_config)

OUT = Path(__file__).resolve().parent / "configs"
CHAIN = ("approach_workspace_default", "approach_workspace_handoff_ready_8mm_12env", "approach_workspace_handoff_ready_8mm_ft_12env",
         "approach_dock_coarse_ready_v1", "approach_finisher_ready_v1")


def main() -> None:
    base = deep_merge(load_yaml_file(approach_default_config_path()), load_yaml_file(ppo_default_config_path()))   # train_approach_policy.py:32-36
    for name in CHAIN:
        cfg = deep_merge(base, load_yaml_file(config_dir() / f"{name}.yaml"))
        to_env_config(cfg)      # the reference itself accepts it
        (OUT / f"{name}.json").write_text(json.dumps(cfg, indent=1, sort_keys=True))
        t, a = cfg["env"]["termination"], cfg["algorithms"]["ppo"]
        print(name, "steps", t["max_episode_steps"], "success", t["success_pos_threshold_m"], t["success_ori_threshold_rad"], "require_ori", t["require_orientation"],
              "lr", a["learning_rate"], "epochs", a["n_epochs"], "gamma", a["gamma"], "clip", a["clip_range"], "total", a["total_timesteps"],
              "stages", len(cfg["env"]["curriculum"]["stages"]))


if __name__ == "__main__":
    main()
