"""End-to-end run of the trainer CLI (mirror of train_workspace_expansion.py) with the eval gate and the final evaluation."""
from __future__ import annotations

import json

import pytest
import yaml

pytestmark = pytest.mark.gpu


def test_train_cli_with_eval_gate_and_final_eval(tmp_path):
    import torch

    from rl_brain_trainer_amd import checkpoint, config as kcfg, train
    from rl_brain_trainer_amd import ppo as P
    from rl_brain_trainer_amd.vec_env import ArmKinematicVecEnv

    cfg_dir = kcfg.builtin_config_dir()
    # a Finisher checkpoint to hand over to: a freshly initialised policy saved in the SB3 layout
    dock_cfg_path = cfg_dir / "dock_workspace_handoff_noop_ft_12env.yaml"
    dock_dict = kcfg.load_yaml_file(dock_cfg_path)
    dock_dict["env"]["dock_reset"]["handoff_state_probability"] = 0.0      # the reference's buffer file is not shipped
    dock_yaml = tmp_path / "dock.yaml"
    dock_yaml.write_text(yaml.safe_dump(dock_dict))
    fenv = ArmKinematicVecEnv(kcfg.to_env_config(dock_dict), 8, seed=1)
    fin = P.PPO(fenv, P.PPOConfig(n_steps=4, batch_size=32, n_epochs=1, hidden=256, seed=1), backend="hip", use_graphs=False)
    fin_zip = checkpoint.save(tmp_path / "finisher", fin)
    fenv.close()
    overlay = {"base_config": str(cfg_dir / "workspace_expansion_bigtrain.yaml"),
               "workspace_expansion": {"finisher_checkpoint": str(fin_zip), "finisher_config": str(dock_yaml), "eval_interval": 4096, "gate_eval_episodes": 3,
                                       "final_eval_episodes": 3, "init_approach_checkpoint": ""}}
    cfg_path = tmp_path / "run.yaml"
    cfg_path.write_text(yaml.safe_dump(overlay))
    root = tmp_path / "run"
    summary = train.main(["--config", str(cfg_path), "--run-id", "t", "--artifact-root", str(root), "--total-timesteps", "8192", "--n-envs", "256",
                          "--n-steps", "16", "--batch-size", "1024", "--hidden", "256", "--log-every", "0"])
    torch.cuda.synchronize()
    assert summary["num_timesteps"] == 8192
    hist = [json.loads(l) for l in (root / "eval_history.jsonl").read_text().splitlines()]
    assert [h["timesteps"] for h in hist] == [4096, 8192]
    assert {"score", "retention_ok", "highest_passed_stage", "candidate"} <= set(hist[0])
    for f in ("latest_checkpoint/model_latest.zip", "model_latest.zip", "training_summary.json", "config_resolved.yaml", "final_eval/stage_metrics.json",
              "stage_metrics.json", "gate_candidates/candidate_step_4096.zip", "gate_evals/eval_step_4096/workspace_eval_summary.json"):
        assert (root / f).exists(), f
    ts = json.loads((root / "training_summary.json").read_text())
    assert ts["final_workspace_eval"]["episodes_per_stage"] == 3 and len(ts["final_workspace_eval"]["stage_metrics"]) == 10
    sd = checkpoint.load_policy_state_dict(root / "model_latest.zip")
    assert sd["mlp_extractor.policy_net.0.weight"].shape == (256, 56)


def test_train_dock_cli_with_reverse_curriculum(tmp_path):
    """Finisher trainer CLI (mirror of train_dock_policy.py) with the dock reverse curriculum on the per-step hook."""
    import torch

    from rl_brain_trainer_amd import config as kcfg, train_dock

    dock = kcfg.load_yaml_file(kcfg.builtin_config_dir() / "dock_workspace_handoff_noop_ft_12env.yaml")
    dock["env"]["dock_reset"]["handoff_state_probability"] = 0.0           # the reference's handoff buffer file is not shipped
    dock.setdefault("training", {})["dock_reverse_curriculum"] = {
        "enabled": True, "window_episodes": 64,
        "stages": [{"name": "anchor", "min_episodes": 64, "window_episodes": 64, "success_rate_threshold": 0.0, "dock_residual_action_limit": 0.2,
                    "close_bucket_probability": 1.0, "close_bucket_min_pos_error_m": 0.001, "close_bucket_max_pos_error_m": 0.004,
                    "close_bucket_max_ori_error_rad": 0.05},
                   {"name": "wide", "close_bucket_probability": 0.2, "dock_residual_action_limit": 0.35}]}
    cfg_path = tmp_path / "dock.yaml"
    cfg_path.write_text(yaml.safe_dump(dock))
    root = tmp_path / "dock_run"
    summary = train_dock.main(["--config", str(cfg_path), "--run-id", "d", "--artifact-root", str(root), "--total-timesteps", "16384", "--n-envs", "256",
                               "--n-steps", "32", "--batch-size", "1024", "--eval-episodes", "64", "--log-every", "0"])
    torch.cuda.synchronize()
    assert summary["num_timesteps"] == 16384 and summary["policy_type"] == "dock"
    cur = summary["dock_reverse_curriculum"]
    assert cur["stage_index"] == 1 and cur["history"][0]["from_stage_name"] == "anchor"     # threshold 0: promoted after 64 episodes
    assert 0.0 <= summary["dock_eval_summary"]["success_rate"] <= 1.0 and summary["dock_eval_summary"]["episodes"] == 64
    for f in ("model_latest.zip", "training_summary.json", "dock_eval/dock_eval_summary.json"):
        assert (root / f).exists(), f
