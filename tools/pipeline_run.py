"""Approach -> Finisher pipeline on the engine's own policies (GPU box): handoff-state buffer from a trained Approach zip, a Finisher trained
on it with train_dock, then the workspace-expansion evaluator with both policies (what the reference's 0.93 / 2.89 mm figure measures).

    python tools/pipeline_run.py <approach.zip> <dock_timesteps> <out.json>      env: KP1_DOCK_LR (1e-4), KP1_DOCK_ENT (1e-4)
"""
import json
import os
import sys
import tempfile
from pathlib import Path

import yaml

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from rl_brain_trainer_amd import config as kcfg  # noqa: E402
from rl_brain_trainer_amd import evaluate as ev  # noqa: E402
from rl_brain_trainer_amd import train_dock  # noqa: E402
from rl_brain_trainer_amd.finisher_tools import build_finisher_handoff_state_buffer  # noqa: E402
from rl_brain_trainer_amd.ppo import InferencePolicy  # noqa: E402

approach_zip, dock_steps, out_path = sys.argv[1], int(float(sys.argv[2])), sys.argv[3]
cfg_a = kcfg.load_workspace_expansion_config(kcfg.builtin_config_dir() / "workspace_expansion_bigtrain.yaml")
cfg_a["env"]["curriculum"]["stages"] = cfg_a["env"]["curriculum"]["stages"][:6]
env_cfg_a = kcfg.to_env_config(cfg_a)
approach = InferencePolicy.load(approach_zip)
tmp = Path(tempfile.mkdtemp())
states = []
for stage in range(6):     # handoff states from every stage the Approach policy was trained on
    buf = build_finisher_handoff_state_buffer(approach_policy=approach.predict, approach_cfg=env_cfg_a, artifact_root=None, episodes=2048, seed=700001 + stage,
                                              stage_index=stage, handoff_mode="final_always")
    states += buf["states"]
    print(f"stage {stage}: mean handoff position error {buf['mean_position_error']:.4f} m, orientation {buf['mean_orientation_error']:.3f} rad", flush=True)
(tmp / "finisher_handoff_state_buffer.json").write_text(json.dumps({"states": states}))
dock = yaml.safe_load((kcfg.builtin_config_dir() / "dock_workspace_handoff_noop_ft_12env.yaml").read_text())
dock["env"]["dock_reset"]["handoff_state_buffer_path"] = str(tmp / "finisher_handoff_state_buffer.json")
dock["algorithms"]["ppo"].update({"learning_rate": float(os.environ.get("KP1_DOCK_LR", "1e-4")), "ent_coef": float(os.environ.get("KP1_DOCK_ENT", "1e-4")),
                                  "n_epochs": 4, "clip_range": 0.1})
(tmp / "dock.yaml").write_text(yaml.safe_dump(dock))
s = train_dock.main(["--config", str(tmp / "dock.yaml"), "--artifact-root", str(tmp / "dock"), "--total-timesteps", str(dock_steps), "--n-envs", "4096", "--n-steps", "36",
                     "--log-every", "50"])
finisher = InferencePolicy.load(s["model_path"])
dock_cfg = kcfg.to_env_config(kcfg.load_dock_config(tmp / "dock.yaml"), handoff_base_dirs=(tmp,))
gate = cfg_a.get("workspace_expansion", {}).get("gate", {})
rows = {}
for name, fin, fcfg in (("approach_only", None, None), ("approach_plus_finisher", finisher.predict, dock_cfg)):
    res = ev.evaluate_workspace_expansion(approach_policy=approach.predict, finisher_policy=fin, approach_cfg=env_cfg_a, finisher_cfg=fcfg, episodes=200, seed=700001,
                                          stage_indices=list(range(6)), gate_config=gate)
    rows[name] = {k: {m: v[m] for m in v if m in ("success_rate", "mean_final_position_error", "mean_final_orientation_error", "final_success_rate",
                                                   "finisher_success_rate", "mean_finisher_final_position_error")} for k, v in res["stage_metrics"].items()}
    print(name, json.dumps(rows[name]), flush=True)
out = {"approach_checkpoint": approach_zip, "handoff_states": len(states), "dock_training": {k: s[k] for k in ("num_timesteps", "wall_seconds", "env_steps_per_second", "dock_eval_summary")},
       "evaluation": rows}
Path(out_path).parent.mkdir(parents=True, exist_ok=True)
Path(out_path).write_text(json.dumps(out, indent=1))
