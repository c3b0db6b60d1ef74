"""Train the Approach policy from scratch through the point curriculum (stages 0 -> 5) on one MI355X and evaluate stage 5.

    python tools/convergence_run.py [total_timesteps] [learning_rate] [out_json]

BASELINE.md quotes 0.93 success / 2.89 mm at stage 5 for the reference after its multi-stage fine-tuning schedule; this script is
the engine's own end-to-end evidence that rollout, curriculum tracker, fused MFMA update and evaluator learn the task.

Env knobs: KP1_CONFIG (approach_default | bigtrain), KP1_EPOCHS, KP1_CLIP, KP1_ENT, KP1_BATCH, KP1_START_STAGE, KP1_SAVE (write the final
model as an SB3 zip), KP1_INIT (resume from one: weights + Adam state).  The two-phase schedule of DESIGN.md section 5
(profiles/r01_learning_finetune_bigtrain_*.json):

    KP1_CONFIG=approach_default KP1_EPOCHS=4 KP1_CLIP=0.2 KP1_ENT=0 KP1_SAVE=/tmp/phase1 python tools/convergence_run.py 1e8 3e-4 p1.json
    KP1_INIT=/tmp/phase1.zip KP1_START_STAGE=5 KP1_EPOCHS=4 KP1_CLIP=0.1 KP1_ENT=1e-4 python tools/convergence_run.py 1.2e10 3e-5 p2.json
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rl_brain_trainer_amd import config as kcfg
from rl_brain_trainer_amd import evaluate as ev
from rl_brain_trainer_amd.curriculum import PointCurriculum
from rl_brain_trainer_amd.ppo import PPO, PPOConfig
from rl_brain_trainer_amd.vec_env import ArmKinematicVecEnv

total = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 3e-4
out = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/convergence.json"
if os.environ.get("KP1_CONFIG", "bigtrain") == "approach_default":   # BASELINE config 1: 20-step episodes, terminate on success, default stages 0..5
    cfg_dict = kcfg.deep_merge(kcfg.load_yaml_file(kcfg.builtin_config_dir() / "approach_default.yaml"), kcfg.load_yaml_file(kcfg.builtin_config_dir() / "ppo_default.yaml"))
    cfg_dict.setdefault("workspace_expansion", {}).setdefault("gate", {})
else:
    cfg_dict = kcfg.load_workspace_expansion_config(kcfg.builtin_config_dir() / "workspace_expansion_bigtrain.yaml")
    cfg_dict["env"]["curriculum"]["stages"] = cfg_dict["env"]["curriculum"]["stages"][:6]      # stages 0..5 (SURVEY config 2 pins stage 5)
env_cfg = kcfg.to_env_config(cfg_dict)
cur = cfg_dict["env"]["curriculum"]
N = 4096
env = ArmKinematicVecEnv(env_cfg, N, seed=806)
curriculum = PointCurriculum(success_rate_threshold=float(cur.get("success_rate_threshold", 0.8)), window_episodes=int(cur.get("window_episodes", 20)),
                             min_episodes_per_stage=int(cur.get("min_episodes_per_stage", 30)), max_stage_index=min(5, env_cfg.n_stages - 1),
                             initial_stage_index=int(os.environ.get("KP1_START_STAGE", "0")),
                             device=0)
epochs = int(os.environ.get("KP1_EPOCHS", "8"))
clip = float(os.environ.get("KP1_CLIP", "0.1"))
ent = float(os.environ.get("KP1_ENT", "3e-4"))
batch = int(os.environ.get("KP1_BATCH", "8192"))
ppo = PPO(env, PPOConfig(n_steps=128, batch_size=batch, n_epochs=epochs, hidden=256, learning_rate=lr, gamma=0.995, gae_lambda=0.95, clip_range=clip, ent_coef=ent,
                         seed=806), curriculum=curriculum, backend="hip")
if os.environ.get("KP1_INIT"):     # fine-tuning schedule: continue from a checkpoint of an earlier phase (weights + Adam state; this phase's constants)
    print("resumed:", ppo.load_checkpoint(os.environ["KP1_INIT"]), flush=True)
start_steps = ppo.num_timesteps
t0 = time.time()
log = []
it = 0
while ppo.num_timesteps - start_steps < total:
    ppo.collect_rollouts()
    ppo.train()
    it += 1
    if it % 20 == 0:
        st = curriculum.read()
        rec = {"it": it, "steps": ppo.num_timesteps, "wall_s": round(time.time() - t0, 1), "stage": int(st.stage_index), "reward_mean": float(ppo.rew_buf.mean()),
               **{k: round(v, 5) if isinstance(v, float) else v for k, v in ppo.last_stats.items()}}
        log.append(rec)
        print(json.dumps(rec), flush=True)
torch.cuda.synchronize()
wall = time.time() - t0
done = ppo.done_buf
train_success = float(((done & 4) != 0).sum()) / max(float(((done & 3) != 0).sum()), 1.0)
print("training-time success rate of the last rollout:", train_success, "episodes", int(((done & 3) != 0).sum()), flush=True)
res_s = ev.evaluate_workspace_expansion(approach_policy=lambda o: ppo.predict(o, deterministic=False), finisher_policy=None, approach_cfg=env_cfg,
                                        finisher_cfg=None, episodes=200, seed=700001, stage_indices=list(range(min(6, env_cfg.n_stages))),
                                        gate_config=cfg_dict["workspace_expansion"]["gate"], obs_stride=ppo.obs_w)
print("stochastic eval:", {k: (v["success_rate"], round(v["mean_final_position_error"], 4)) for k, v in res_s["stage_metrics"].items()}, flush=True)
res = ev.evaluate_workspace_expansion(approach_policy=ppo.predict, finisher_policy=None, approach_cfg=env_cfg, finisher_cfg=None, episodes=200, seed=700001,
                                      stage_indices=list(range(min(6, env_cfg.n_stages))), gate_config=cfg_dict["workspace_expansion"]["gate"], obs_stride=ppo.obs_w)
summary = {"hyper": {"epochs": epochs, "clip": clip, "ent_coef": ent, "batch": batch}, "total_timesteps": ppo.num_timesteps, "wall_seconds": wall, "env_steps_per_second": ppo.num_timesteps / wall, "learning_rate": lr,
           "final_stage": int(curriculum.read().stage_index), "curriculum": curriculum.summary(),
           "stage_metrics": {k: {m: v[m] for m in ("success_rate", "mean_final_position_error", "mean_final_orientation_error")} for k, v in res["stage_metrics"].items()},
           "log": log}
if os.environ.get("KP1_SAVE"):
    from rl_brain_trainer_amd import checkpoint

    print("saved", checkpoint.save(os.environ["KP1_SAVE"], ppo, env_cfg), flush=True)
os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
json.dump(summary, open(out, "w"), indent=1)
print(json.dumps({k: summary[k] for k in ("total_timesteps", "wall_seconds", "final_stage", "stage_metrics")}, indent=1))
