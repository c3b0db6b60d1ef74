"""Evaluator and checkpoint tests on the GPU."""
from __future__ import annotations

import io
import json
import math
import zipfile

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden_config
from oracle import oracle as orc
from rl_brain_trainer_amd import checkpoint, config as kcfg
from rl_brain_trainer_amd import evaluate as ev
from rl_brain_trainer_amd import ppo as P
from rl_brain_trainer_amd.vec_env import ArmKinematicVecEnv

pytestmark = pytest.mark.gpu


def test_curriculum_local_eval_suite_golden():
    g = np.load(GOLDEN / "eval_suites.npz")
    cfg = load_golden_config("workspace_expansion_1h_extend")
    for k in range(cfg.n_stages):
        s = ev.build_curriculum_local_eval_suite(cfg, seed=700001 + 1009 * k, stage_index=k, n_episodes=8)
        assert np.array_equal(s["initial_q"], g[f"s{k}_initial_q"])
        assert np.array_equal(s["goal_q"], g[f"s{k}_goal_q"])
        assert np.max(np.abs(s["goal_pose6"] - g[f"s{k}_goal_pose6"])) <= 1e-12


def _servo_policy_gpu(env: ArmKinematicVecEnv, gain: float):
    dl = torch.tensor(env.config.c.joints.delta_limit[:], device="cuda", dtype=torch.float64)
    scale = env.config.c.env.dock_action_delta_scale or env.config.c.env.action_delta_scale

    def policy(obs):
        info = env.info()
        a = gain * (info["goal_q"].double().t() - info["q"].double().t()) / (dl * scale)
        return a.clamp(-1, 1).to(env.dtype)

    return policy


def _serial_oracle_episode(cfg, options, gain, ready_cfg, confirm):
    """The reference's serial loop (_run_approach_with_handoff / _run_policy) on the CPU oracle."""
    env = orc.OracleEnv(cfg)
    env.reset(options=options)
    dl = np.array(cfg.c.joints.delta_limit[:])
    scale = cfg.c.env.dock_action_delta_scale or cfg.c.env.action_delta_scale
    streak = max_streak = steps = 0
    first_ready = None
    hand = None
    done = False
    info = None
    while not done:
        st = env.state()
        a = np.clip(gain * (st["goal_q"] - st["q"]) / (dl * scale), -1, 1)
        an = float(np.linalg.norm(a))
        _, info = env.step(a)
        steps += 1
        pos, ori, dqn = info["position_error_norm"], info["orientation_error_norm"], info["executed_delta_q_l2"]
        r = ready_cfg
        rdy = bool(r is not None and r.dock_coarse_ready_pos_threshold_m > 0 and r.dock_coarse_ready_ori_threshold_rad > 0
                   and pos <= r.dock_coarse_ready_pos_threshold_m and ori <= r.dock_coarse_ready_ori_threshold_rad
                   and (r.dock_coarse_ready_action_threshold <= 0 or an <= r.dock_coarse_ready_action_threshold)
                   and (r.dock_coarse_ready_dq_threshold <= 0 or dqn <= r.dock_coarse_ready_dq_threshold))
        if rdy:
            first_ready = first_ready or steps
            streak += 1
        else:
            streak = 0
        max_streak = max(max_streak, streak)
        if hand is None and confirm and streak >= confirm:
            hand = {"step": steps, "pos": pos, "state": env.state()}
        done = info["terminated"] or info["truncated"]
    return {"success": info["success"], "pos": info["position_error_norm"], "ori": info["orientation_error_norm"], "steps": steps,
            "max_streak": max_streak, "first_ready": first_ready, "hand": hand, "state": env.state()}


def test_batched_handoff_runner_matches_serial_oracle():
    """f64 env + servo controller: per-episode results of the batched runner equal the serial oracle loop."""
    cfg = load_golden_config("workspace_expansion_bigtrain")
    suite = ev.build_curriculum_local_eval_suite(cfg, seed=700001 + 1009 * 5, stage_index=5, n_episodes=12)
    env = ArmKinematicVecEnv(cfg, 12, seed=1, real="f64")
    opts = {"initial_q": suite["initial_q"], "goal_q": suite["goal_q"], "goal_pose6": suite["goal_pose6"], "policy_mode": "approach"}
    res, hand = ev.run_episodes(env, _servo_policy_gpu(env, 0.7), opts, ready_cfg=cfg.c.reward, handoff_confirm_steps=2)
    assert bool(hand["valid"].any())                     # the servo reaches the handoff zone: the path under test is exercised
    for e in range(12):
        o = {k: v[e] for k, v in opts.items() if k != "policy_mode"}
        o["policy_mode"] = "approach"
        ref = _serial_oracle_episode(cfg, o, 0.7, cfg.c.reward, 2)
        assert bool(res["success"][e]) == ref["success"], e
        assert int(res["step_count"][e]) == ref["steps"]
        assert abs(float(res["final_position_error"][e]) - ref["pos"]) <= 1e-10
        assert abs(float(res["final_orientation_error"][e]) - ref["ori"]) <= 1e-10
        assert int(res["max_ready_streak"][e]) == ref["max_streak"]
        assert int(res["first_ready_step"][e]) == (ref["first_ready"] or -1)
        assert bool(hand["valid"][e]) == (ref["hand"] is not None)
        if ref["hand"] is not None:
            assert int(hand["step_count"][e]) == ref["hand"]["step"]
            assert np.max(np.abs(hand["state_q"][e].cpu().numpy() - ref["hand"]["state"]["q"])) <= 1e-12
        assert np.max(np.abs(res["state_q"][e].cpu().numpy() - ref["state"]["q"])) <= 1e-12
    env.close()


def test_workspace_expansion_eval_schema_and_finisher():
    """Full Approach -> Finisher evaluation with (untrained) MFMA policies: output schema of the reference's
    stage_metrics / best_model_selection, finite metrics, deterministic across two runs."""
    acfg = load_golden_config("workspace_expansion_bigtrain")
    fcfg = load_golden_config("dock_workspace_handoff_noop_ft_12env_raw")
    pol_a = P.InferencePolicy(P.ActorCritic(256, torch.device("cuda", 0), seed=1).state_dict())
    pol_f = P.InferencePolicy(P.ActorCritic(256, torch.device("cuda", 0), seed=2).state_dict())
    gate = {"retention_stage0_4_success": 0.95, "retention_stage5_success": 0.85, "promotion_stage_success": 0.8, "promotion_ready_rate": 0.8,
            "max_mean_position_error_m": 0.02, "max_mean_orientation_error_rad": 0.15}
    outs = [ev.evaluate_workspace_expansion(approach_policy=pol_a, finisher_policy=pol_f, approach_cfg=acfg, finisher_cfg=fcfg, episodes=6,
                                            seed=700001, stage_indices=[0, 5, 9], gate_config=gate, obs_stride=64) for _ in range(2)]
    a, b = outs
    assert json.dumps(a["stage_metrics"], sort_keys=True) == json.dumps(b["stage_metrics"], sort_keys=True)
    assert set(a["stage_metrics"]) == {"0", "5", "9"}
    keys = {"episode_count", "success_rate", "finisher_ready_hit_rate", "dwell_success_rate", "mean_final_position_error",
            "mean_final_orientation_error", "mean_final_action_magnitude", "mean_final_dq_norm", "regression_rate", "failure_reason_counts"}
    assert set(a["stage_metrics"]["5"]) == keys
    assert set(a["best_model_selection"]) == {"score", "current_stage", "retention_ok", "highest_passed_stage", "current_stage_success_rate",
                                              "current_stage_ready_rate", "retention_mean_success_rate", "error_score"}
    assert all(np.isfinite(v["mean_final_position_error"]) for v in a["stage_metrics"].values())
    assert len(a["target_rows"]) == 18


def test_checkpoint_zip_roundtrip(tmp_path):
    cfg = load_golden_config("workspace_expansion_bigtrain")
    env = ArmKinematicVecEnv(cfg, 256, seed=806)
    env.set_curriculum_stage(5)
    ppo = P.PPO(env, P.PPOConfig(n_steps=8, batch_size=1024, n_epochs=1, hidden=256, learning_rate=1e-4, seed=5), use_graphs=False)
    ppo.collect_rollouts()
    ppo.train()
    path = checkpoint.save(tmp_path / "model_latest", ppo, cfg)
    assert path.name == "model_latest.zip"
    with zipfile.ZipFile(path) as z:
        assert set(z.namelist()) == {"data", "pytorch_variables.pth", "policy.pth", "policy.optimizer.pth", "_stable_baselines3_version", "system_info.txt"}
        opt = torch.load(io.BytesIO(z.read("policy.optimizer.pth")), weights_only=True)
    assert list(opt["param_groups"][0]["params"]) == list(range(13)) and opt["param_groups"][0]["eps"] == 1e-5
    sd = checkpoint.load_policy_state_dict(path)
    assert list(sd) == [n for n, _ in P.param_spec(256)]
    assert sd["mlp_extractor.policy_net.0.weight"].shape == (256, 56) and sd["action_net.weight"].shape == (7, 256)
    data = checkpoint.load_data(path)
    assert data["policy_kwargs"] == {"net_arch": {"pi": [256, 256], "vf": [256, 256]}} and data["n_steps"] == 8
    pol = P.InferencePolicy.load(str(path))
    obs = env.current_observation()
    a1 = pol.predict(obs)
    a2 = ppo.predict(obs)
    assert torch.equal(a1, a2)
    # reference-default width (SB3 2x64) loads through the torch path
    small = P.InferencePolicy(P.ActorCritic(64, torch.device("cuda", 0), seed=9).state_dict())
    assert small.predict(obs).shape == (256, 7)
    env.close()


def test_ppo_learns_stage0_reaching():
    """End-to-end sanity: a few PPO iterations on the 20-step Stage-0 env raise the mean step reward and the update is
    identical with and without hipGraph replay (same seeds)."""
    cfg = load_golden_config("approach_default")
    finals = []
    for graphs in (False, True):
        env = ArmKinematicVecEnv(cfg, 1024, seed=7)
        env.set_curriculum_stage(0)
        ppo = P.PPO(env, P.PPOConfig(n_steps=20, batch_size=4096, n_epochs=4, hidden=256, learning_rate=3e-4, gamma=0.99, clip_range=0.2, seed=7),
                    use_graphs=graphs)
        rewards = []
        for _ in range(12):
            ppo.collect_rollouts()
            rewards.append(float(ppo.rew_buf.mean()))
            ppo.train()
        finals.append((rewards, ppo.policy.flat.clone()))
        env.close()
    r = finals[0][0]
    assert np.mean(r[-3:]) > np.mean(r[:3]) + 0.002, r
    assert all(np.isfinite(r))


def test_resume_restores_adam_state(tmp_path):
    """PPO.load for a resumed run (train_workspace_expansion.py:187-197, train_route_curriculum.py:129-139): after load_checkpoint the
    Adam moments, the common step count, the actor tensors' extra step count and the device-resident counter are those of the saved
    model, so the next optimiser step is bitwise the one the saved model would have taken."""
    cfg = load_golden_config("workspace_expansion_bigtrain")

    def make(seed):
        env = ArmKinematicVecEnv(cfg, 128, seed=806)
        env.set_curriculum_stage(5)
        return env, P.PPO(env, P.PPOConfig(n_steps=8, batch_size=512, n_epochs=2, hidden=256, learning_rate=1e-3, seed=seed), use_graphs=True)

    env_a, a = make(5)
    for _ in range(2):
        a.collect_rollouts()
        a.train()
    a.actor_extra_steps = 3                      # as after three teacher-anchor steps
    a._mlp.set_actor_extra_steps(3)
    path = checkpoint.save(tmp_path / "model", a, cfg)
    env_b, b = make(99)
    b.cfg.gamma, b.cfg.ent_coef, b.cfg.clip_range, b.cfg.n_epochs = 0.9, 0.123, 0.3, 5
    restored = b.load_checkpoint(str(path), restore_timesteps=True, restore_hyperparameters=True)
    assert restored["optimizer"] and restored["adam_steps"] == a.adam_t == 8 and restored["actor_extra_steps"] == 3
    assert (b.cfg.gamma, b.cfg.ent_coef, b.cfg.clip_range, b.cfg.n_epochs) == (a.cfg.gamma, a.cfg.ent_coef, a.cfg.clip_range, a.cfg.n_epochs)
    assert b.cfg.seed == 99 and set(restored["hyperparameters"]) >= {"gamma", "gae_lambda", "ent_coef", "vf_coef", "max_grad_norm", "n_epochs", "clip_range"}
    assert b.num_timesteps == a.num_timesteps
    assert torch.equal(b.policy.flat, a.policy.flat) and torch.equal(b.adam_m, a.adam_m) and torch.equal(b.adam_v, a.adam_v)
    g = torch.Generator(device="cuda").manual_seed(3)
    grad = 1e-3 * torch.randn(a.policy.flat.numel(), device="cuda", generator=g)
    for ppo in (a, b):   # device-resident step count (what the graph-replayed update uses): both must read 8 (+3 for the actor tensors)
        ppo._mlp.adam_step(ppo.policy.flat, grad.clone(), ppo.adam_m, ppo.adam_v, lr=1e-3, eps=1e-5, max_grad_norm=0.5, step=0)
    assert torch.equal(b.policy.flat, a.policy.flat) and torch.equal(b.adam_m, a.adam_m)
    # and the host-step form agrees with torch's formula for step 9 on a non-actor tensor (log_std) and step 12 on an actor tensor
    env_c, c = make(7)
    c.load_checkpoint(str(path))
    before = c.policy.flat.clone()
    m0, v0 = c.adam_m.clone(), c.adam_v.clone()
    c._mlp.adam_step(c.policy.flat, grad.clone(), c.adam_m, c.adam_v, lr=1e-3, eps=1e-5, max_grad_norm=0.0, step=c.adam_t + 1)
    for idx, step in ((0, 9), (7 + 5, 12)):     # element 0 = log_std[0]; element 12 = policy_net.0.weight[0, 5]
        gi = grad[idx].item()
        mi = 0.9 * m0[idx].item() + 0.1 * gi
        vi = 0.999 * v0[idx].item() + 0.001 * gi * gi
        expect = before[idx].item() - 1e-3 / (1 - 0.9 ** step) * mi / (math.sqrt(vi) / math.sqrt(1 - 0.999 ** step) + 1e-5)
        assert abs(c.policy.flat[idx].item() - expect) <= 2e-6 * max(1.0, abs(expect)), (idx, step)
    for e in (env_a, env_b, env_c):
        e.close()


@pytest.mark.parametrize("cfg_name,real,confirm,masked", [("approach_dock_coarse_ready_v1", "f32", 2, False), ("fuzz1_approach", "f32", None, True),
                                                          ("fuzz1_approach", "f64", 3, True), ("fuzz2_approach", "f32", 2, False),
                                                          ("fuzz2_approach", "f32", 0, True)])
def test_device_eval_bookkeeping_equals_tensor_expressions(cfg_name, real, confirm, masked):
    """kp1_eval_accumulate (one launch per env step) against the tensor-expression form of the evaluator's bookkeeping
    (_run_episodes_reference): 600 episodes from sampled resets under a noisy servo of mixed gains -- episodes succeed and terminate early
    (the fuzz configs terminate on success), hover around the ready thresholds or time out -- every result and handoff tensor bit for bit."""
    cfg = load_golden_config(cfg_name)
    E = 600
    g = np.random.default_rng(4)
    gains = torch.tensor(g.choice([0.0, 0.15, 0.5, 0.9], size=E), device="cuda", dtype=torch.float64)[:, None]
    active = torch.tensor(g.random(E) < 0.8, device="cuda") if masked else None

    def make():
        env = ArmKinematicVecEnv(cfg, E, seed=5, real=real)
        dl = torch.tensor(env.config.c.joints.delta_limit[:], device="cuda", dtype=torch.float64) * env.config.c.env.action_delta_scale
        noise = torch.Generator(device="cuda").manual_seed(9)

        def policy(obs):
            info = env.info()
            a = gains * (info["goal_q"].double().t() - info["q"].double().t()) / dl
            a = a + 0.02 * torch.randn(a.shape, device="cuda", dtype=torch.float64, generator=noise)
            return a.clamp(-1.5, 1.5).to(env.dtype)       # unclipped range on purpose: |action| is taken before the env clips

        return env, policy

    out = []
    for fn in (ev.run_episodes, ev._run_episodes_reference):
        env, policy = make()
        out.append(fn(env, policy, None, ready_cfg=cfg.c.reward, handoff_confirm_steps=confirm, active=active))
        env.close()
    (res, hand), (ref, ref_hand) = out
    assert set(res) == set(ref)
    for k in ref:
        assert res[k].dtype == ref[k].dtype and torch.equal(res[k], ref[k]), k
    assert bool(ref["ready_hit"].any()) and not bool(ref["ready_hit"].all())          # the scenario exercises the ready bookkeeping ...
    if cfg.c.termination.terminate_on_success:
        steps = ref["step_count"][ref["step_count"] > 0]
        assert int(steps.max()) > int(steps.min()) and bool(ref["success"].any())     # ... and episodes of different lengths
    if confirm is None:                                      # _run_policy: no handoff bookkeeping at all
        assert hand is None and ref_hand is None
    else:
        if confirm > 0:
            assert bool(ref_hand["valid"].any()) and not bool(ref_hand["valid"].all())
        else:                                                # ready_streak >= 0 holds at step 1 (eval_pipeline_ablation.py:103): every episode hands over there
            live = active if active is not None else torch.ones(E, dtype=torch.bool, device="cuda")
            assert torch.equal(ref_hand["valid"], live) and bool((ref_hand["step_count"][live] == 1).all())
        assert set(hand) == set(ref_hand) and {"mean_action_magnitude", "mean_dq_norm"} <= set(hand)
        for k in ref_hand:                                   # the reference creates a handoff entry when the first episode hands over
            assert hand[k].dtype == ref_hand[k].dtype and torch.equal(hand[k], ref_hand[k]), k
    if masked:
        assert torch.equal(res["step_count"][~active], torch.zeros_like(res["step_count"][~active]))
