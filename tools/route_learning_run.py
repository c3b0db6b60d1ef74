"""Route-curriculum learning check (GPU box): train_route on the synthetic route fixture from scratch and report how far the prefix
curriculum gets and what the sequential evaluator says.  Env knobs: KP1_ROUTE_STEPS, KP1_ROUTE_LR, KP1_ROUTE_ENVS, KP1_ROUTE_OUT."""
import json
import os
import sys
import tempfile
from pathlib import Path

import yaml

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from rl_brain_trainer_amd import train_route  # noqa: E402

G = ROOT / "tests" / "golden"
cfg = json.loads((G / "configs" / "route_curriculum_prefix120_routeobs_sequence2.json").read_text())
cfg["route"].pop("init_checkpoint", None)
cfg["route"]["route_path"] = str(G / "synthetic_route.json")
cfg["route"]["curriculum"] = {**cfg["route"].get("curriculum", {}), "prefix_stages": [20, 40, 80, 120]}
cfg.setdefault("training", {})["checkpoint_freq"] = 10 ** 9     # the YAML's 250 k-step period is a checkpoint every 0.1 s at this speed
cfg["route"]["sequential_gate"] = {"enabled": True, "prefixes": [20, 40, 80, 120], "full_end_index": None}
cfg["algorithms"]["ppo"].update({"learning_rate": float(os.environ.get("KP1_ROUTE_LR", "3e-4")), "n_epochs": int(os.environ.get("KP1_ROUTE_EPOCHS", "4")), "ent_coef": float(os.environ.get("KP1_ROUTE_ENT", "1e-3")), "clip_range": float(os.environ.get("KP1_ROUTE_CLIP", "0.2"))})
steps = int(os.environ.get("KP1_ROUTE_STEPS", "20000000"))
envs = int(os.environ.get("KP1_ROUTE_ENVS", "2048"))
with tempfile.TemporaryDirectory() as td:
    p = Path(td) / "route.yaml"
    p.write_text(yaml.safe_dump(cfg))
    s = train_route.main(["--config", str(p), "--run-id", "route_learning", "--output-dir", str(Path(td) / "run"), "--total-timesteps", str(steps), "--n-envs", str(envs),
                          "--n-steps", "64", "--batch-size", str(envs * 64 // 16), "--seed", "1", "--log-every", "10"])
out = {"steps": s["num_timesteps"], "wall_seconds": s["wall_seconds"], "env_steps_per_second": s["env_steps_per_second"], "observation_dim": s["observation_dim"],
       "curriculum": {k: v for k, v in s["curriculum_summary"].items()}, "sequential_eval": s["route_eval_sequential_summary"],
       "gate": {k: {kk: vv for kk, vv in v.items() if kk in ("success_rate", "longest_success_prefix", "first_failure_index", "first_failure_reason")}
                for k, v in s["route_gate_summary"].get("prefix_results", {}).items()}}
dst = Path(os.environ.get("KP1_ROUTE_OUT", str(ROOT / "gpurun_out" / "route_learning.json")))
dst.parent.mkdir(parents=True, exist_ok=True)
dst.write_text(json.dumps(out, indent=1))
print(json.dumps(out["curriculum"] | {"fps": out["env_steps_per_second"], "seq": out["sequential_eval"].get("success_rate")}, default=str))
