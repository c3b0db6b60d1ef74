"""Finisher tooling: dock reverse curriculum vs the reference callback's golden trace (CPU); handoff buffer builder
vs the serial oracle loop, and the buffer it writes feeds the dock reset sampler (GPU)."""
from __future__ import annotations

import json

import numpy as np
import pytest

from conftest import GOLDEN, load_golden_config
from rl_brain_trainer_amd import finisher_tools as ft


class _FakeEnv:
    def __init__(self):
        self.calls = []

    def env_method(self, name, payload):
        self.calls.append([name, payload])


def test_dock_reverse_curriculum_matches_reference_callback():
    g = json.loads((GOLDEN / "dock_reverse_curriculum.json").read_text())
    for trace in g["traces"]:
        cb = ft.DockReverseCurriculum(stages=g["stages"], window_episodes=g["window_episodes"])
        env = _FakeEnv()
        cb.on_training_start(env)
        for step in trace["steps"]:
            cb.on_step(step["dones"], step["success"])
            assert (cb.current_stage_index, cb.stage_episode_count) == (step["stage"], step["count"])
        assert json.loads(json.dumps(cb.summary())) == trace["summary"]
        assert json.loads(json.dumps(env.calls)) == trace["calls"]
    with pytest.raises(ValueError):
        ft.DockReverseCurriculum(stages=[], window_episodes=4)


@pytest.mark.gpu
def test_handoff_buffer_builder_matches_serial_oracle_and_feeds_the_dock_sampler(tmp_path):
    import torch

    from rl_brain_trainer_amd import config as kcfg
    from rl_brain_trainer_amd import evaluate as ev
    from rl_brain_trainer_amd.vec_env import ArmKinematicVecEnv
    from test_eval_checkpoint_gpu import _serial_oracle_episode

    cfg = load_golden_config("workspace_expansion_bigtrain")
    gain, episodes, stage = 0.7, 24, 5
    holder = {}
    orig = ev.run_episodes

    def run(env, policy, opts, **kw):
        holder["env"] = env
        return orig(env, policy, opts, **kw)

    def policy(obs):
        env = holder["env"]
        dl = torch.tensor(env.config.c.joints.delta_limit[:], device="cuda", dtype=torch.float64)
        info = env.info()
        a = gain * (info["goal_q"].double().t() - info["q"].double().t()) / (dl * env.config.c.env.action_delta_scale)
        return a.clamp(-1, 1).to(env.dtype)

    ev.run_episodes = run
    try:
        out = {m: ft.build_finisher_handoff_state_buffer(approach_policy=policy, approach_cfg=cfg, artifact_root=tmp_path / m, episodes=episodes, seed=700001 + 1009 * stage,
                                                         stage_index=stage, handoff_mode=m) for m in ("final_settled", "first_confirmed", "final_always")}
    finally:
        ev.run_episodes = orig
    assert out["final_always"]["stored_handoff_count"] == episodes
    assert out["first_confirmed"]["stored_handoff_count"] > 0
    suite = ev.build_curriculum_local_eval_suite(cfg, seed=700001 + 1009 * stage, stage_index=stage, n_episodes=episodes)
    stored = {s["episode_id"]: s for s in out["first_confirmed"]["states"]}
    for e in range(episodes):
        o = {k: suite[k][e] for k in ("initial_q", "goal_q", "goal_pose6")}
        ref = _serial_oracle_episode(cfg, {**o, "policy_mode": "approach"}, gain, cfg.c.reward, 2)
        assert (ref["hand"] is not None) == (e in stored), e
        if ref["hand"] is not None:
            assert stored[e]["step_index"] == ref["hand"]["step"]
            assert np.max(np.abs(np.array(stored[e]["initial_q"]) - ref["hand"]["state"]["q"])) <= 2e-5
            assert abs(stored[e]["position_error_norm"] - ref["hand"]["pos"]) <= 2e-5
    # the written file is a valid dock_reset.handoff_state_buffer_path
    dock = json.loads((GOLDEN / "configs" / "dock_workspace_handoff_noop_ft_12env_raw.json").read_text())
    dock["env"]["dock_reset"]["handoff_state_buffer_path"] = str(tmp_path / "first_confirmed" / "finisher_handoff_state_buffer.json")
    dock["env"]["dock_reset"]["handoff_state_probability"] = 1.0
    dcfg = kcfg.to_env_config(dock)
    assert len(dcfg.handoff_states) > 0
    env = ArmKinematicVecEnv(dcfg, 64, seed=3)
    env.reset()
    goal = env.get_state()["goal_q"]
    stored_goals = np.array([s["goal_q"] for s in out["first_confirmed"]["states"]])
    assert all(np.min(np.abs(stored_goals - gq).max(axis=1)) <= 1e-6 for gq in goal)
    env.close()
