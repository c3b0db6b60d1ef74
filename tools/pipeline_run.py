"""Approach -> Finisher pipeline on the engine's own policies (GPU box): the measurement behind the reference's Stage 0-5 table
(docs/PHASE1_APPROACH_DOCK_CLOSEOUT.md:52-63: Approach = approach_finisher_ready_v2_settle, Dock = dock_workspace_handoff_noop_ft_12env,
first-confirmed handoff, six curriculum regions).

    python tools/pipeline_run.py <approach.zip> <out.json> [--approach-config approach_finisher_ready_v2_settle]
                                 [--dock-steps 1.2e9] [--dock-ft-steps 3e8] [--episodes 200]

1. handoff-state buffer from the trained Approach policy over stages 0..5 (first-confirmed finisher-ready states, 2048 episodes per stage);
2. Finisher trained on it with train_dock: from scratch at lr 1e-4, then fine-tuned FROM THAT CHECKPOINT with the config's own
   algorithms.ppo (lr 3e-6, gamma 0.98, clip 0.08, 5 epochs), as the reference fine-tunes its Finisher from an earlier checkpoint;
3. evaluate_workspace_expansion (eval_workspace_expansion.py:86-211 protocol) with both policies, 200 held-out episodes per stage.
"""
import argparse
import json
import sys
import tempfile
from pathlib import Path

import yaml

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from rl_brain_trainer_amd import config as kcfg  # noqa: E402
from rl_brain_trainer_amd import evaluate as ev  # noqa: E402
from rl_brain_trainer_amd import train_dock  # noqa: E402
from rl_brain_trainer_amd.finisher_tools import build_finisher_handoff_state_buffer  # noqa: E402
from rl_brain_trainer_amd.ppo import InferencePolicy  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("approach_zip")
ap.add_argument("out")
ap.add_argument("--approach-config", default="approach_finisher_ready_v2_settle")
ap.add_argument("--dock-steps", type=float, default=1.2e9)
ap.add_argument("--dock-ft-steps", type=float, default=3e8)
ap.add_argument("--dock-zip", default="", help="skip Finisher training and use this checkpoint")
ap.add_argument("--episodes", type=int, default=200)
ap.add_argument("--handoff-mode", default="first_confirmed")
ap.add_argument("--dock-envs", type=int, default=4096)
ap.add_argument("--dock-n-steps", type=int, default=36)
ap.add_argument("--dock-batch", type=int, default=0, help="minibatch rows; 0 = rollout/64")
ap.add_argument("--dock-hidden", type=int, default=256)
ap.add_argument("--dock-scratch-lr", type=float, default=1e-4)
ap.add_argument("--dock-scratch-epochs", type=int, default=4)
ap.add_argument("--dock-seed", type=int, default=7)
args = ap.parse_args()

cfg_a = kcfg.load_workspace_expansion_config(kcfg.builtin_config_dir() / f"{args.approach_config}.yaml")
cfg_a["env"]["curriculum"]["stages"] = cfg_a["env"]["curriculum"]["stages"][:6]
env_cfg_a = kcfg.to_env_config(cfg_a)
approach = InferencePolicy.load(args.approach_zip)
tmp = Path(tempfile.mkdtemp())
states = []
buffer_stats = {}
for stage in range(6):     # handoff states from every region the Approach policy was trained on
    buf = build_finisher_handoff_state_buffer(approach_policy=approach.predict, approach_cfg=env_cfg_a, artifact_root=None, episodes=2048, seed=710001 + stage,
                                              stage_index=stage, handoff_mode=args.handoff_mode)
    states += buf["states"]
    buffer_stats[stage] = {k: buf[k] for k in ("stored_handoff_rate", "mean_position_error", "mean_orientation_error")}
    print(f"stage {stage}: stored {buf['stored_handoff_rate']:.3f} handoff pos {buf['mean_position_error']} ori {buf['mean_orientation_error']}", flush=True)
if not states:
    raise SystemExit("the Approach policy produced no finisher-ready handoff state: nothing to train the Finisher on")
(tmp / "finisher_handoff_state_buffer.json").write_text(json.dumps({"states": states}))
dock = yaml.safe_load((kcfg.builtin_config_dir() / "dock_workspace_handoff_noop_ft_12env.yaml").read_text())
dock["env"]["dock_reset"]["handoff_state_buffer_path"] = str(tmp / "finisher_handoff_state_buffer.json")
(tmp / "dock_ft.yaml").write_text(yaml.safe_dump(dock))                      # the config's own algorithms.ppo: lr 3e-6, clip 0.08, 5 epochs
scratch = json.loads(json.dumps(dock))
scratch["algorithms"]["ppo"].update({"learning_rate": args.dock_scratch_lr, "ent_coef": 1e-4, "n_epochs": args.dock_scratch_epochs, "clip_range": 0.1})
(tmp / "dock_scratch.yaml").write_text(yaml.safe_dump(scratch))
training = {}
scale = ["--n-envs", str(args.dock_envs), "--n-steps", str(args.dock_n_steps), "--batch-size", str(args.dock_batch), "--hidden", str(args.dock_hidden),
         "--seed", str(args.dock_seed), "--log-every", "100"]
if args.dock_zip:
    dock_zip = args.dock_zip
else:
    s1 = train_dock.main(["--config", str(tmp / "dock_scratch.yaml"), "--artifact-root", str(tmp / "dock_scratch"), "--total-timesteps", str(int(args.dock_steps)),
                          *scale])
    training["scratch"] = {k: s1[k] for k in ("num_timesteps", "wall_seconds", "env_steps_per_second", "dock_eval_summary")}
    dock_zip = s1["model_path"]
    if args.dock_ft_steps > 0:
        s2 = train_dock.main(["--config", str(tmp / "dock_ft.yaml"), "--artifact-root", str(tmp / "dock_ft"), "--total-timesteps", str(int(args.dock_ft_steps)),
                              *scale, "--resume-from", dock_zip])
        training["fine_tune"] = {k: s2[k] for k in ("num_timesteps", "wall_seconds", "env_steps_per_second", "dock_eval_summary")}
        dock_zip = s2["model_path"]
finisher = InferencePolicy.load(dock_zip)
dock_cfg = kcfg.to_env_config(kcfg.load_dock_config(tmp / "dock_ft.yaml"), handoff_base_dirs=(tmp,))
gate = cfg_a.get("workspace_expansion", {}).get("gate", {})
rows = {}
keep = ("success_rate", "finisher_ready_hit_rate", "dwell_success_rate", "mean_final_position_error", "mean_final_orientation_error", "mean_handoff_position_error",
        "mean_handoff_orientation_error", "handoff_rate")
for name, fin, fcfg in (("approach_only", None, None), ("approach_plus_finisher", finisher.predict, dock_cfg)):
    res = ev.evaluate_workspace_expansion(approach_policy=approach.predict, finisher_policy=fin, approach_cfg=env_cfg_a, finisher_cfg=fcfg, episodes=args.episodes,
                                          seed=700001, stage_indices=list(range(6)), gate_config=gate)
    rows[name] = {k: {m: v[m] for m in keep if m in v} for k, v in res["stage_metrics"].items()}
    for k, v in res["stage_metrics"].items():
        mine = [r for r in res["target_rows"] if str(r.get("stage_index", r.get("stage"))) == str(k)]
        rows[name][k]["failure_reason_counts"] = v.get("failure_reason_counts")
        if mine:
            rows[name][k]["mean_approach_final_position_error"] = sum(r["approach_final_position_error"] for r in mine) / len(mine)
            rows[name][k]["mean_approach_final_orientation_error"] = sum(r["approach_final_orientation_error"] for r in mine) / len(mine)
    print(name, json.dumps(rows[name]), flush=True)
out = {"command": "python " + " ".join(sys.argv), "approach_checkpoint": args.approach_zip, "approach_config": args.approach_config, "handoff_mode": args.handoff_mode, "handoff_states": len(states),
       "handoff_buffer": buffer_stats, "dock_scale": {"n_envs": args.dock_envs, "n_steps": args.dock_n_steps, "batch": args.dock_batch, "hidden": args.dock_hidden},
       "dock_training": training, "evaluation": rows,
       "target": {"stage5_success": 0.93, "stage5_final_position_error_m": 0.00289, "source": "report/OFFICIAL_ARTIFACTS.md:26"}}
Path(args.out).parent.mkdir(parents=True, exist_ok=True)
Path(args.out).write_text(json.dumps(out, indent=1))
