"""PPO on the route-curriculum envs with the 80-float route observation: rollout buffers, MFMA loss/gradient against torch autograd on
rows the device env produced, the teacher-anchor side step against torch.optim.Adam, and the trainer CLI end to end."""
from __future__ import annotations

import json
import math

import numpy as np
import pytest
import torch
import yaml

from conftest import GOLDEN
from rl_brain_trainer_amd import config as kcfg
from rl_brain_trainer_amd import ppo as P
from rl_brain_trainer_amd import route_config as rcfg
from rl_brain_trainer_amd.route_env import RouteVecEnv
from rl_brain_trainer_amd.teacher_anchor import ACTOR_TENSORS, RouteTeacherAnchor, TeacherAnchorConfig

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _cfg() -> dict:
    return json.loads((GOLDEN / "configs" / "route_curriculum_prefix120_routeobs_sequence2.json").read_text())


def _make(n_envs=128, use_graphs=False, **kw):
    cfgd = _cfg()
    route_q = rcfg.load_route_q(GOLDEN / "synthetic_route.json")
    env = RouteVecEnv(kcfg.to_env_config(cfgd), rcfg.route_config_from_dict(cfgd, max_route_index=40), route_q, n_envs, seed=11)
    pc = P.PPOConfig(n_steps=16, batch_size=512, n_epochs=2, hidden=256, learning_rate=3e-4, gamma=0.99, clip_range=0.2, ent_coef=1e-3, seed=5, **kw)
    return env, P.PPO(env, pc, use_graphs=use_graphs), route_q


@pytest.mark.parametrize("use_graphs", [False, True])
def test_route_ppo_rollout_and_gradient(use_graphs):
    env, ppo, _ = _make(use_graphs=use_graphs)
    assert env.obs_dim == 80 and ppo.obs_dim == 80 and ppo.obs_w == 128 and ppo.policy.views["mlp_extractor.policy_net.0.weight"].shape == (256, 80)
    for _ in range(12):                      # 120-step episodes: a few 16-step rollouts until episodes end inside one
        ppo.collect_rollouts()
        if (ppo.done_buf & 3).any():
            break
    T, N = 16, 128
    obs = ppo.obs_buf[:T].reshape(T * N, 128)
    assert torch.all(obs[:, 80:] == 0) and torch.isfinite(obs).all()
    d = RouteVecEnv.obs_dict(obs)
    assert d["route_q_error"].abs().max() <= 1 and d["route_scalar"].min() >= 0 and d["route_tangent"].abs().max() > 0
    assert (ppo.done_buf & 3).any()          # episodes end inside the rollout (auto-reset path exercised)
    # one minibatch: fused MFMA path vs SB3's loss under torch autograd on the same rows
    g = torch.Generator(device=DEV).manual_seed(1)
    idx = torch.randperm(T * N, device=DEV, generator=g)[:1024]
    act, old_logp = ppo.act_buf.view(-1, 7), ppo.logp_buf.view(-1)
    adv, ret = ppo.adv_buf.view(-1), ppo.ret_buf.view(-1)
    grad = torch.empty(ppo._mlp.num_params, device=DEV)
    ppo._mlp.loss_grad(obs, idx, 1024, act, old_logp, adv, ret, clip_range=0.2, ent_coef=1e-3, vf_coef=0.5, inv_count=1.0 / 1024, grad_out=grad, stats_out=None)
    a = adv[idx]
    a = (a - a.mean()) / (a.std() + 1e-8)
    ref = P.ppo_loss_and_grad_torch(ppo.policy.flat, ppo.policy.spec, obs[idx, :80], act[idx], old_logp[idx], a, ret[idx], clip_range=0.2, ent_coef=1e-3,
                                    vf_coef=0.5)
    off = 0
    for name, shape in ppo.policy.spec:
        cnt = math.prod(shape)
        gk, gr = grad[off:off + cnt], ref[off:off + cnt]
        scale = gr.abs().max().item() + 1e-12
        assert (gk - gr).abs().max().item() <= 2e-4 * scale + 1e-7, name      # fp32 summation-order tolerance, as in test_ppo_kernels_gpu
        off += cnt
    before = ppo.policy.flat.clone()
    ppo.train()
    ppo.collect_rollouts()
    ppo.train()
    assert torch.isfinite(ppo.policy.flat).all() and (ppo.policy.flat != before).any()
    assert all(math.isfinite(v) for v in ppo.last_stats.values())
    env.close()


def test_teacher_anchor_step_matches_torch_adam(tmp_path):
    """After a PPO update (non-trivial Adam state) one anchor step must equal torch.optim.Adam on the actor tensors with their per-tensor
    step counts, and the next HIP Adam step must use step+extra for exactly those tensors."""
    env, ppo, route_q = _make(use_graphs=False)
    ppo.collect_rollouts()
    ppo.train()
    t0 = ppo.adam_t
    assert t0 > 0
    # teacher dataset: observations of the device env, teacher = a servo toward the current route goal
    ppo.collect_rollouts()
    obs = ppo.obs_buf[:16].reshape(-1, 128)[:, :80].cpu().numpy()
    d = {k: obs[:, o:o + w] for k, (o, w) in rcfg.ROUTE_OBS_LAYOUT.items()}
    actions = np.clip(0.8 * d["route_q_error"], -1, 1).astype(np.float32)
    route_index = np.arange(len(obs)) % 60
    path = tmp_path / "teacher.npz"
    np.savez(path, actions=actions, route_index=route_index.astype(np.int32), **{f"obs__{k}": v for k, v in d.items()})
    anchor = RouteTeacherAnchor(TeacherAnchorConfig(enabled=True, dataset_path=str(path), loss_weight=0.5, batch_size=256, max_route_index=40))
    anchor.on_training_start(ppo)
    assert anchor.summary()["sample_count"] == int((route_index <= 40).sum())

    # torch.optim.Adam reference on Parameters carrying the same state
    spec_off, o = {}, 0
    for name, shape in ppo.policy.spec:
        spec_off[name] = (o, shape)
        o += math.prod(shape)
    params = {n: torch.nn.Parameter(ppo.policy.views[n].detach().clone()) for n in ACTOR_TENSORS}
    opt = torch.optim.Adam(list(params.values()), lr=ppo.cfg.learning_rate, eps=ppo.cfg.adam_eps)
    for n, p in params.items():
        s, shape = spec_off[n]
        cnt = math.prod(shape)
        opt.state[p] = {"step": torch.tensor(float(t0)), "exp_avg": ppo.adam_m[s:s + cnt].view(shape).clone(), "exp_avg_sq": ppo.adam_v[s:s + cnt].view(shape).clone()}
    idx = np.random.default_rng(0).integers(0, anchor.summary()["sample_count"], size=256)
    keep = np.nonzero(route_index <= 40)[0]
    bo = torch.as_tensor(obs[keep][idx], device=DEV)
    ba = torch.as_tensor(actions[keep][idx], device=DEV)
    views = {**ppo.policy.views, **params}
    mean, _ = P.mlp_forward(views, bo)
    loss = torch.nn.functional.mse_loss(mean, ba) * 0.5
    opt.zero_grad()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(list(params.values()), max_norm=0.5)
    opt.step()

    untouched = {n: ppo.policy.views[n].clone() for n, _ in ppo.policy.spec if n not in ACTOR_TENSORS}
    anchor.on_rollout_end(ppo)
    assert abs(anchor.last_loss - loss.item()) <= 1e-6 * max(1.0, abs(loss.item()))
    for n, p in params.items():
        assert torch.allclose(ppo.policy.views[n], p.detach(), rtol=1e-5, atol=1e-7), n     # same fp32 formulas, op order may differ by an ulp
        s, shape = spec_off[n]
        cnt = math.prod(shape)
        assert torch.allclose(ppo.adam_m[s:s + cnt].view(shape), opt.state[p]["exp_avg"], rtol=1e-5, atol=1e-9)
    for n, v in untouched.items():
        assert torch.equal(ppo.policy.views[n], v), n
    assert anchor.actor_extra_steps == 1
    # the packed kernel weights follow the flat vector
    m1, _ = ppo._forward(ppo.obs_buf[0])
    m2, _ = P.mlp_forward(ppo.policy.views, ppo.obs_buf[0][:, :80])
    assert torch.allclose(m1, m2, rtol=1e-4, atol=2e-5)

    # next optimiser step on the device: per-tensor step counts t0+1 (+1 for the actor tensors)
    g = torch.Generator(device=DEV).manual_seed(9)
    grad = 1e-3 * torch.randn(ppo.policy.flat.numel(), device=DEV, generator=g)
    allp = {n: torch.nn.Parameter(ppo.policy.views[n].detach().clone()) for n, _ in ppo.policy.spec}
    opt2 = torch.optim.Adam(list(allp.values()), lr=ppo.cfg.learning_rate, eps=ppo.cfg.adam_eps)
    for n, p in allp.items():
        s, shape = spec_off[n]
        cnt = math.prod(shape)
        opt2.state[p] = {"step": torch.tensor(float(t0 + (1 if n in ACTOR_TENSORS else 0))), "exp_avg": ppo.adam_m[s:s + cnt].view(shape).clone(),
                         "exp_avg_sq": ppo.adam_v[s:s + cnt].view(shape).clone()}
        p.grad = grad[s:s + cnt].view(shape).clone()
    torch.nn.utils.clip_grad_norm_(list(allp.values()), max_norm=ppo.cfg.max_grad_norm)
    opt2.step()
    ppo._mlp.adam_step(ppo.policy.flat, grad.clone(), ppo.adam_m, ppo.adam_v, lr=ppo.cfg.learning_rate, eps=ppo.cfg.adam_eps, max_grad_norm=ppo.cfg.max_grad_norm,
                       step=t0 + 1, fused_norm=False)
    for n, p in allp.items():
        assert torch.allclose(ppo.policy.views[n], p.detach(), rtol=2e-5, atol=1e-7), n
    env.close()


def test_train_route_cli(tmp_path):
    from rl_brain_trainer_amd import train_route

    cfgd = _cfg()
    route_path = GOLDEN / "synthetic_route.json"
    # teacher dataset from a servo on a few device envs
    route_q = rcfg.load_route_q(route_path)
    env = RouteVecEnv(kcfg.to_env_config(cfgd), rcfg.route_config_from_dict(cfgd, max_route_index=20), route_q, 64, seed=3)
    rows, acts, ridx = [], [], []
    obs = env.reset()
    for _ in range(20):
        d = RouteVecEnv.obs_dict(obs)
        a = (0.8 * d["route_q_error"]).clamp(-1, 1)
        rows.append(obs.cpu().numpy().copy())
        acts.append(a.cpu().numpy().copy())
        ridx.append(env.info()["route_index"].cpu().numpy().copy())
        obs, _, _ = env.step(a)
    env.close()
    rows, acts, ridx = np.concatenate(rows), np.concatenate(acts), np.concatenate(ridx)
    npz = tmp_path / "teacher.npz"
    np.savez(npz, actions=acts.astype(np.float32), route_index=ridx.astype(np.int32),
             **{f"obs__{k}": rows[:, o:o + w] for k, (o, w) in rcfg.ROUTE_OBS_LAYOUT.items()})
    cfgd["route"]["curriculum"] = {**cfgd["route"].get("curriculum", {}), "prefix_stages": [10, 20], "promotion_window_episodes": 32, "min_episodes_per_stage": 32,
                                   "promotion_success_rate": 0.0, "promotion_route_ready_hit_rate": 0.0, "promotion_orientation_hit_rate": 0.0,
                                   "promotion_max_regression_rate": 1.0}
    cfgd["route"]["teacher_anchor"] = {"enabled": True, "dataset_path": str(npz), "loss_weight": 0.02, "batch_size": 128, "max_route_index": 20}
    cfgd["route"]["sequential_gate"] = {"enabled": True, "prefixes": [5, 10], "full_end_index": 12}
    cfgd["route"]["route_path"] = str(route_path)
    cfgd["route"].pop("init_checkpoint", None)      # the reference's artefact path; a resumed run is covered by test_train_cli_gpu
    cfgd.setdefault("training", {})["checkpoint_freq"] = 4096
    cfg_path = tmp_path / "route.yaml"
    cfg_path.write_text(yaml.safe_dump(cfgd))
    out = tmp_path / "run"
    summary = train_route.main(["--config", str(cfg_path), "--run-id", "t", "--output-dir", str(out), "--total-timesteps", "32768", "--n-envs", "128", "--n-steps", "16",
                                "--batch-size", "512", "--seed", "4"])
    assert summary["schema_version"] == "v5.route_curriculum.training_summary.v1" and summary["observation_dim"] == 80
    assert summary["curriculum_summary"]["prefix_end_index"] == 20 and len(summary["curriculum_summary"]["history"]) == 1   # promoted once (thresholds open)
    assert summary["teacher_anchor_summary"]["enabled"] and summary["teacher_anchor_summary"]["sample_count"] > 0
    assert summary["route_gate_summary"]["schema_version"] == "v5.route_gate.v1" and summary["route_gate_summary"]["accepted"] is False
    assert set(summary["route_gate_summary"]["prefix_results"]) == {"prefix_5", "prefix_10"}
    for name in ("model_latest.zip", "curriculum_history.json", "training_summary.json", "route_eval_sequential/route_eval_sequential_summary.json",
                 "route_gate/route_gate_summary.json", "route_gate/full_12/route_eval_sequential_summary.json", "checkpoints/model_4096_steps.zip"):
        assert (out / name).exists(), name
    # resumed run: PPO.load(init_checkpoint) + learn(reset_num_timesteps=False) -- the step clock and the Adam state carry on
    out2 = tmp_path / "run2"
    s2 = train_route.main(["--config", str(cfg_path), "--run-id", "t2", "--output-dir", str(out2), "--total-timesteps", "4096", "--n-envs", "128", "--n-steps", "16",
                           "--batch-size", "512", "--seed", "4", "--init-checkpoint", str(out / "model_latest.zip")])
    assert s2["init_checkpoint"] == str(out / "model_latest.zip") and s2["num_timesteps"] == summary["num_timesteps"] + 4096
    from rl_brain_trainer_amd import checkpoint as ck
    o1, o2 = ck.load_optimizer_state_dict(out / "model_latest.zip"), ck.load_optimizer_state_dict(out2 / "model_latest.zip")
    assert float(o2["state"][0]["step"]) > float(o1["state"][0]["step"]) > 0
    assert float(o2["state"][1]["step"]) - float(o2["state"][0]["step"]) > float(o1["state"][1]["step"]) - float(o1["state"][0]["step"]) > 0   # anchor steps
    # the zip reloads as an 80-input policy
    pol = P.InferencePolicy.load(str(out / "model_latest.zip"))
    assert pol.obs_dim == 80
    a = pol.predict(torch.zeros((3, 80), device=DEV))
    assert a.shape == (3, 7) and torch.isfinite(a).all()


def test_anchor_consumes_a_teacher_dataset(tmp_path):
    """The anchor callback on a dataset in the reference recorder's format (collect_route_teacher_rollout.py:96-113: ``obs__<key>`` arrays in
    Dict-key layout, ``actions``, ``route_index``, ``step``).  The recorder itself is a data-collection tool outside the hot path
     and is not rebuilt; the file is written here from a servo teacher's (observation, action) pairs."""
    cfgd = _cfg()
    route_q = rcfg.load_route_q(GOLDEN / "synthetic_route.json")
    env0 = RouteVecEnv(kcfg.to_env_config(cfgd), rcfg.route_config_from_dict(cfgd, max_route_index=120), route_q, 64, seed=5)
    rows, acts, idxs = [], [], []
    obs = env0.reset()
    for _ in range(6):
        a = (0.8 * RouteVecEnv.obs_dict(obs)["route_q_error"]).clamp(-1, 1)
        rows.append(obs[:, :env0.obs_dim].cpu().numpy().copy())
        acts.append(a.cpu().numpy().copy())
        idxs.append(env0.info()["route_index"].cpu().numpy().copy())
        obs, _, _ = env0.step(a)
    env0.close()
    obs_mat, actions = np.concatenate(rows).astype(np.float32), np.concatenate(acts).astype(np.float32)
    arrays = {"actions": actions, "route_index": np.concatenate(idxs).astype(np.int32), "step": np.repeat(np.arange(6, dtype=np.int32), 64)}
    for key, (off, width) in rcfg.ROUTE_OBS_LAYOUT.items():
        arrays[f"obs__{key}"] = obs_mat[:, off:off + width]
    path = tmp_path / "teacher_route_anchor_dataset.npz"
    np.savez_compressed(path, **arrays)
    env, ppo, _ = _make(n_envs=32)
    anchor = RouteTeacherAnchor(TeacherAnchorConfig(enabled=True, dataset_path=str(path), batch_size=64, max_route_index=120))
    anchor.on_training_start(ppo)
    before = ppo.policy.flat.clone()
    anchor.on_rollout_end(ppo)
    assert anchor.summary()["sample_count"] == actions.shape[0] and anchor.last_loss > 0 and (ppo.policy.flat != before).any()
    env.close()


def test_device_prefix_curriculum_matches_reference_callback():
    """The reference callback's recorded (dones, infos) stream (tests/golden/route_eval.json) fed to the DEVICE tracker through the env's own
    flag buffers: stage / episode count after every step, the promotion history with its four window rates and timesteps, the final window
    rates, and the reset window the promotions leave in the device route config."""
    from rl_brain_trainer_amd.route_curriculum import RoutePrefixCurriculumDevice, build_prefix_stages

    gold = json.loads((GOLDEN / "route_eval.json").read_text())
    cfgd = _cfg()
    route_q = rcfg.load_route_q(GOLDEN / "synthetic_route.json")
    for trace in gold["callback"]:
        n = len(trace["steps"][0]["dones"])
        env = RouteVecEnv(kcfg.to_env_config(cfgd), rcfg.route_config_from_dict(cfgd, max_route_index=120), route_q, n, seed=2)
        cb = RoutePrefixCurriculumDevice(stages=build_prefix_stages([20, 40, 80]), promotion_success_rate=0.75, promotion_route_ready_hit_rate=0.75,
                                         promotion_orientation_hit_rate=0.85, promotion_max_regression_rate=0.30, window_episodes=16, min_episodes_per_stage=24)
        cb.attach(env)
        env.reset()
        assert int(env.info()["route_index"].max()) <= 20          # the first prefix was applied
        info = env.info()
        views = [info["route_ready"], info["route_orientation_hit"], info["route_regression"]]
        for step in trace["steps"]:
            infos = step["infos"]
            done = torch.tensor([(1 if d else 0) | (4 if (d and i["success"]) else 0) for d, i in zip(step["dones"], infos)], dtype=torch.uint8, device=DEV)
            # a success flag on a not-done env must be ignored, like info["success"] of an unfinished episode
            done |= torch.tensor([4 if (i["success"] and not d) else 0 for d, i in zip(step["dones"], infos)], dtype=torch.uint8, device=DEV)
            for v, key in zip(views, ("route_ready", "route_orientation_hit", "route_regression")):
                v.copy_(torch.tensor([int(i[key]) for i in infos], dtype=torch.uint8, device=DEV))
            cb.observe(done, n)
            st = cb.read()
            assert (int(st.stage_index), int(st.stage_episode_count)) == (step["stage"], step["count"])
        assert json.loads(json.dumps(cb.summary())) == trace["summary"]
        assert env.route_cfg.reset.max_route_index == trace["summary"]["prefix_end_index"]
        env.reset()                                                  # the device config carries the promoted window
        idx = env.info()["route_index"]
        assert int(idx.max()) <= trace["summary"]["prefix_end_index"]
        cb.close()
        env.close()
