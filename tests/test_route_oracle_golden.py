"""Route-curriculum oracle (oracle/kp1_route_oracle.c) pinned to the reference's outputs (tests/golden/route_*.npz, written by
tests/golden/make_golden_route.py from the imported reference).  CPU only."""
from __future__ import annotations

import ctypes as C
import json

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import oracle as orc
from oracle import route_oracle as ro
from rl_brain_trainer_amd import config as kcfg
from rl_brain_trainer_amd import route_config as rcfg


def _cfg_dict(name: str) -> dict:
    return json.loads((GOLDEN / "configs" / f"{name}.json").read_text())


@pytest.fixture(scope="module")
def route() -> ro.Route:
    return ro.Route(rcfg.load_route_q(GOLDEN / "synthetic_route.json"))


def _golden_words(w: np.ndarray) -> np.ndarray:
    """make_golden_route.py stores (state lo, hi, inc lo, hi, has, uint); the oracle reports (hi, lo, hi, lo, has, uint)"""
    return np.array([w[1], w[0], w[3], w[2], w[4], w[5]], dtype=np.uint64)


def test_route_dataset_matches_reference(route):
    g = np.load(GOLDEN / "route_dataset.npz")
    assert np.array_equal(route.q, g["q_goals"])
    assert np.max(np.abs(route.poses6 - g["poses6"])) <= 1e-13
    assert np.max(np.abs(route.progress - g["progress"])) <= 1e-12
    assert np.array_equal(route.next_q_delta, g["next_q_delta"])
    assert np.array_equal(route.chunk_id, g["chunk_id"])


def test_numpy_normal_and_choice_streams_bit_exact():
    g = np.load(GOLDEN / "route_rng_streams.npz")
    L = ro.lib()
    r = orc.ORng()
    L.kp1o_rng_seed(C.byref(r), 817)
    mine = np.array([[0.0 + 0.0008 * L.kp1o_rng_standard_normal(C.byref(r)) for _ in range(7)] for _ in range(g["normals"].shape[0])])
    assert np.array_equal(mine, g["normals"])
    L.kp1o_rng_seed(C.byref(r), 5)
    p = np.ascontiguousarray(g["choice_p"], dtype=np.float64)
    ch = np.array([L.kp1o_rng_choice_p(C.byref(r), p.ctypes.data_as(C.POINTER(C.c_double)), 5) for _ in range(g["choices"].shape[0])])
    assert np.array_equal(ch, g["choices"])
    # the restated sampler against numpy itself on a long stream (tail and wedge paths included)
    L.kp1o_rng_seed(C.byref(r), 123456)
    ref = np.random.default_rng(123456).standard_normal(200000)
    mine = np.array([L.kp1o_rng_standard_normal(C.byref(r)) for _ in range(200000)])
    assert np.array_equal(mine, ref)


def test_route_reward_cases(route):
    g = np.load(GOLDEN / "route_reward.npz")
    assert [str(s) for s in g["component_names"]] == rcfg.COMPONENT_NAMES
    cfg = rcfg.route_config_from_dict(_cfg_dict("route_curriculum_prefix120_routeobs_sequence2"))
    n = g["reward"].shape[0]
    worst = 0.0
    for i in range(n):
        goal = g["goal_q"][i]
        idx = int(np.argmin(np.linalg.norm(route.q - goal, axis=1)))
        reward, comps = ro.route_reward(cfg.reward, prev_q=g["prev_q"][i], curr_q=g["curr_q"][i], goal_q=goal, prev_pose6=orc.fk_pose6(g["prev_q"][i][None])[0],
                                        curr_pose6=orc.fk_pose6(g["curr_q"][i][None])[0], goal_pose6=route.poses6[idx], tangent=g["tangent"][i],
                                        action=g["action"][i], prev_action=g["prev_action"][i], prev_dq=g["prev_dq"][i], curr_dq=g["curr_dq"][i],
                                        ready_streak=int(g["streak"][i]), nearest=float(g["nearest"][i]))
        worst = max(worst, abs(reward - g["reward"][i]), float(np.max(np.abs(comps - g["components"][i]))))
    assert worst <= 1e-12, worst


@pytest.mark.parametrize("tag", ["prefix120", "prefix20", "replay", "forced_segment", "prefix170"])
def test_route_reset_sampler_streams(route, tag):
    g = np.load(GOLDEN / f"route_resets_{tag}.npz")
    cfg = rcfg.route_config_from_dict({"route": {"reset": json.loads(str(g["reset_config"]))}})
    base = kcfg.to_env_config(_cfg_dict(str(g["config"])))
    L = ro.lib()
    r = orc.ORng()
    L.kp1o_rng_seed(C.byref(r), int(g["seed"]))
    s = ro.RouteSample()
    for i in range(g["route_index"].shape[0]):
        assert np.array_equal(orc.rng_words(r), _golden_words(g["rng_before"][i])), i
        L.kp1o_route_sample_reset(C.byref(r), route._h, C.byref(base.c.joints), C.byref(cfg.reset), C.byref(s))
        assert np.array_equal(orc.rng_words(r), _golden_words(g["rng_after"][i])), i
        assert (s.route_index, s.start_index, s.mode) == (int(g["route_index"][i]), int(g["start_index"][i]), int(g["mode"][i])), i
        for name in ("initial_q", "initial_dq", "initial_prev_action", "goal_q"):
            assert np.array_equal(np.array(getattr(s, name)[:]), g[name][i]), (i, name)
    assert len(set(g["mode"].tolist())) >= (1 if tag == "forced_segment" else 3)


@pytest.mark.parametrize("name,cfg_name,max_index", [("seq_prefix120", "route_curriculum_prefix120_routeobs_sequence2", 120),
                                                     ("seq_prefix170", "route_curriculum_prefix170_routeobs_sequence2", 170),   # BASELINE configs[4]
                                                     ("seq_prefix20", "route_curriculum_prefix20_sequence2", 20),
                                                     ("single_default", "route_curriculum_default", 20)])
def test_route_env_traces(route, name, cfg_name, max_index):
    g = np.load(GOLDEN / f"route_trace_{name}.npz")
    cfgd = _cfg_dict(cfg_name)
    env = ro.OracleRouteEnv(kcfg.to_env_config(cfgd), rcfg.route_config_from_dict(cfgd, max_route_index=max_index), route)
    assert env.obs_dim == g["obs"].shape[1]
    resets = set(g["reset_at"].tolist())
    row = 0            # index into the recorded rows (reset rows + step rows)
    t = 0              # env steps taken
    k = 0              # reset counter
    T = int(np.sum(~np.isnan(g["reward"])))
    worst_r = worst_c = 0.0
    while t < T:
        assert t in resets
        if k > 0:  # before the first reset the reference's stream is entropy-seeded
            assert np.array_equal(env.rng_words(), _golden_words(g["rng_before"][k]))
        obs = env.reset(seed=int(g["seed"]) if k == 0 else None)
        assert np.array_equal(env.rng_words(), _golden_words(g["rng_after"][k]))
        assert env.field("reset_mode") == int(g["reset_mode"][k]) and env.field("start_route_index") == int(g["start_index"][k])
        if g["last_index"][k] >= 0:
            assert env.field("last_route_index") == int(g["last_index"][k])
        assert np.array_equal(obs, g["obs"][row]), (name, row)
        assert env.field("current_route_index") == int(g["route_index"][row])
        row += 1
        k += 1
        done = False
        while not done and t < T:
            obs, out = env.step(g["action"][row])
            assert np.array_equal(obs, g["obs"][row]), (name, row, np.max(np.abs(obs - g["obs"][row])))
            worst_r = max(worst_r, abs(out.reward - g["reward"][row]))
            worst_c = max(worst_c, float(np.max(np.abs(np.array(out.components[:]) - g["components"][row]))))
            for key, val in (("terminated", out.terminated), ("truncated", out.truncated), ("route_index", out.route_index), ("ready", out.route_ready),
                             ("streak", out.ready_streak), ("success", out.success), ("waypoint_success", out.waypoint_success),
                             ("completed", out.completed_waypoints)):
                assert int(val) == int(g[key][row]), (name, row, key, val, g[key][row])
            assert abs(out.q_error_norm - g["q_error"][row]) <= 1e-13 and abs(out.nearest_route_q_distance - g["nearest"][row]) <= 1e-13
            assert np.max(np.abs(env.base_state()["q"] - g["q"][row])) <= 1e-15
            done = bool(out.terminated or out.truncated)
            row += 1
            t += 1
    assert worst_r <= 1e-12 and worst_c <= 1e-12, (worst_r, worst_c)
    assert int(g["waypoint_success"].sum()) > 0     # the traces exercise waypoint hand-over / success
