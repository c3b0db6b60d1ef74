"""Pin the CPU oracle (oracle/) against the golden vectors captured from the imported reference.

Bar: integers, flags, RNG words bit-exact; fp64 quantities <= 1e-12 abs (the reference's numpy
matmul / norm summation order is not fully visible, so last-bit differences are allowed);
float32 observations <= 1 ulp-ish (6e-8).
"""
from __future__ import annotations

import json

import numpy as np
import pytest

from conftest import GOLDEN, load_golden_config
from oracle import oracle as orc
from rl_brain_trainer_amd import config as kcfg

TOL = 1e-12

TRACES = {
    "approach_default_s0_seed7": "approach_default",
    "approach_default_s5_seed0": "approach_default",
    "bigtrain_s5_seed806": "workspace_expansion_bigtrain",
    "bigtrain_s0_seed123": "workspace_expansion_bigtrain",
    "extend_s11_seed0": "workspace_expansion_1h_extend",
    "extend_s8_seed7": "workspace_expansion_1h_extend",
    "dynscale_s9_seed123": "workspace_expansion_dynamic_scale_big",
    "randomstart_s10_seed931": "workspace_full_coverage_randomstart_overnight",
    "settle_v2_s5_seed7": "approach_finisher_ready_v2_settle",
    "dock_noop_seed7": "dock_workspace_handoff_noop_ft_12env_raw",
    "dock_noop_merged_seed0": "dock_workspace_handoff_noop_ft_12env",
    "dock_default_seed123": "dock_default",
    # randomised env-level + reward configuration (tests/golden/make_golden_env_fuzz.py)
    "fuzz0_approach": "fuzz0_approach",
    "fuzz1_approach": "fuzz1_approach",
    "fuzz2_approach": "fuzz2_approach",
    "fuzz3_dock": "fuzz3_dock",
    "fuzz4_dock": "fuzz4_dock",
    "fuzz5_dock": "fuzz5_dock",
}
RESETS = {
    "approach_default_s0": "approach_default",
    "approach_default_s3": "approach_default",
    "bigtrain_s5": "workspace_expansion_bigtrain",
    "bigtrain_s9": "workspace_expansion_bigtrain",
    "extend_s11": "workspace_expansion_1h_extend",
    "extend_s1": "workspace_expansion_1h_extend",
    "randomstart_s10": "workspace_full_coverage_randomstart_overnight",
    "randomstart_s4": "workspace_full_coverage_randomstart_overnight",
    "dock_noop": "dock_workspace_handoff_noop_ft_12env_raw",
    "dock_default": "dock_default",
}


def ref_obs_to_kp1(obs_ref: np.ndarray) -> np.ndarray:
    """reference dict-order flat obs -> SB3/kp1 order"""
    out = np.zeros(obs_ref.shape[:-1] + (kcfg.OBS_DIM,), dtype=np.float32)
    off = 0
    for key in kcfg.OBS_KEYS_REFERENCE_ORDER:
        start, n = kcfg.OBS_LAYOUT[key]
        out[..., start:start + n] = obs_ref[..., off:off + n]
        off += n
    assert off == kcfg.OBS_DIM
    return out


def test_fk_golden():
    g = np.load(GOLDEN / "fk.npz")
    out = orc.fk_pose6(g["q"])
    assert np.max(np.abs(out - g["pose6"])) <= TOL
    # known answers from SURVEY.md 8a (a1)
    assert np.allclose(out[0], [-0.18001025845587856, 0.0017638930333356549, 1.1004499999999167,
                                1.5707963267948963, 6.72e-15, -1.5707963267948928], atol=1e-12)
    assert np.allclose(out[1], [0.0075883570617028065, 0.20862492597027565, 1.0469532800504862,
                                1.5858008218590405, -0.6388109212845965, -2.0705752330192593], atol=1e-12)


def test_pose_error_golden():
    g = np.load(GOLDEN / "pose_error.npz")
    for i in range(g["curr"].shape[0]):
        pe, oe = orc.pose_error(g["curr"][i], g["goal"][i])
        assert np.array_equal(pe, g["pos_err"][i])
        assert np.max(np.abs(oe - g["ori_err"][i])) <= 1e-15, i
        assert np.all(oe >= -np.pi) and np.all(oe < np.pi)


def test_joint_utils_golden():
    import ctypes as C

    g = np.load(GOLDEN / "joint_utils.npz")
    cfg = kcfg.default_config()
    assert np.array_equal(np.array(cfg.joints.lower[:]), g["lower"])
    assert np.array_equal(np.array(cfg.joints.upper[:]), g["upper"])
    assert np.array_equal(np.array(cfg.joints.delta_limit[:]), g["delta_limits"])
    L = orc.lib()
    out = np.empty(7)
    for i in range(g["q"].shape[0]):
        q = np.ascontiguousarray(g["q"][i])
        L.kp1o_clip_q(C.byref(cfg.joints), orc._dp(q), orc._dp(out))
        assert np.array_equal(out, g["clipped"][i])
        clipped = out.copy()
        L.kp1o_joint_limit_margin(C.byref(cfg.joints), orc._dp(clipped), orc._dp(out))
        assert np.max(np.abs(out - g["margin"][i])) <= 1e-15
        L.kp1o_normalize_q(C.byref(cfg.joints), orc._dp(q), orc._dp(out))
        assert np.max(np.abs(out - g["q_norm"][i])) <= 1e-15
        dq = np.ascontiguousarray(g["dq"][i])
        L.kp1o_normalize_dq(C.byref(cfg.joints), orc._dp(dq), orc._dp(out))
        assert np.max(np.abs(out - g["dq_norm"][i])) <= 1e-15


def test_default_config_matches_oracle_default():
    import ctypes as C

    a = kcfg.default_config()
    b = kcfg.Kp1Config()
    orc.lib().kp1o_config_default(C.byref(b))
    assert bytes(a) == bytes(b)


@pytest.mark.parametrize("name", sorted(TRACES))
def test_step_trace_golden(name):
    g = np.load(GOLDEN / f"trace_{name}.npz")
    meta = json.loads(str(g["meta"]))
    cfg = load_golden_config(TRACES[name])
    env = orc.OracleEnv(cfg)
    env.set_curriculum_stage(meta["stage"])
    mode = kcfg.MODE_NAMES[meta["mode"]]
    assert list(g["component_keys"]) == orc.component_names(mode)
    assert list(g["obs_keys"]) == kcfg.OBS_KEYS_REFERENCE_ORDER

    reset_at = {int(s): k for k, s in enumerate(g["reset_at_step"])}
    obs_gold = ref_obs_to_kp1(g["obs"])
    reset_obs_gold = ref_obs_to_kp1(g["reset_obs"])

    def check_reset(k, first):
        if first:
            obs = env.reset(seed=meta["seed"])
        else:
            assert np.array_equal(env.rng_words(), g["reset_rng_before"][k]), f"rng before reset {k}"
            obs = env.reset()
        assert np.array_equal(env.rng_words(), g["reset_rng_after"][k]), f"rng after reset {k}"
        st = env.state()
        assert np.array_equal(st["q"], g["reset_initial_q"][k]), k
        assert np.array_equal(st["dq"], g["reset_initial_dq"][k])
        assert np.array_equal(st["prev_action"], g["reset_initial_prev_action"][k])
        assert np.array_equal(st["goal_q"], g["reset_goal_q"][k])
        assert np.max(np.abs(st["goal_pose6"] - g["reset_goal_pose6"][k])) <= TOL
        assert np.max(np.abs(st["ee_pose6"] - g["reset_ee_pose6"][k])) <= TOL
        assert np.max(np.abs(st["entry_metrics"] - g["reset_entry_metrics"][k])) <= TOL
        assert np.max(np.abs(obs - reset_obs_gold[k])) <= 6e-8

    check_reset(0, True)
    T = g["action"].shape[0]
    for t in range(T):
        obs, info = env.step(g["action"][t])
        ctx = f"{name} step {t}"
        for key, gk in (("terminated", "terminated"), ("truncated", "truncated"), ("success", "success"),
                        ("dwell_count", "dwell"), ("near_goal_entry_count", "entry"), ("near_goal_drift_count", "drift"),
                        ("pre_near_goal_hit", "pre_near_hit"), ("near_goal_hit", "near_hit"), ("step_count", "episode_step")):
            assert int(info[key]) == int(g[gk][t]), f"{ctx}: {key}"
        for key, gk in (("reward", "reward"), ("position_error_norm", "pos_err"), ("orientation_error_norm", "ori_err"),
                        ("min_position_error", "min_pos_error"), ("executed_delta_q_l2", "exec_dq_l2"), ("action_l2", "action_l2"),
                        ("delta_q_change_l2", "dq_change_l2"), ("dock_action_limit", "dock_action_limit"),
                        ("dock_delta_q_change_limit_scale", "dock_dq_change_limit_scale"), ("joint_limit_margin_min", "margin_min")):
            assert abs(info[key] - g[gk][t]) <= TOL, f"{ctx}: {key} {info[key]} vs {g[gk][t]}"
        assert np.max(np.abs(info["q"] - g["q"][t])) <= TOL, ctx
        assert np.max(np.abs(info["dq"] - g["dq"][t])) <= TOL, ctx
        assert np.max(np.abs(info["ee_pose6"] - g["ee_pose6"][t])) <= TOL, ctx
        diff = np.abs(info["components"] - g["components"][t])
        assert np.max(diff) <= TOL, f"{ctx}: component {g['component_keys'][int(np.argmax(diff))]}"
        assert np.max(np.abs(obs - obs_gold[t])) <= 6e-8, ctx
        if (t + 1) in reset_at:
            assert info["terminated"] or info["truncated"]
            check_reset(reset_at[t + 1], False)


@pytest.mark.parametrize("name", sorted(RESETS))
def test_reset_stream_golden(name):
    g = np.load(GOLDEN / f"resets_{name}.npz")
    meta = json.loads(str(g["meta"]))
    env = orc.OracleEnv(load_golden_config(RESETS[name]))
    env.set_curriculum_stage(meta["stage"])
    obs_gold = ref_obs_to_kp1(g["obs"])
    for i in range(g["initial_q"].shape[0]):
        if i == 0:
            obs = env.reset(seed=meta["seed"])
        else:
            assert np.array_equal(env.rng_words(), g["rng_before"][i])
            obs = env.reset()
        assert np.array_equal(env.rng_words(), g["rng_after"][i]), i
        st = env.state()
        assert np.array_equal(st["q"], g["initial_q"][i]), i
        assert np.array_equal(st["dq"], g["initial_dq"][i])
        assert np.array_equal(st["prev_action"], g["initial_prev_action"][i])
        assert np.array_equal(st["goal_q"], g["goal_q"][i])
        assert np.max(np.abs(st["goal_pose6"] - g["goal_pose6"][i])) <= TOL
        assert np.max(np.abs(obs - obs_gold[i])) <= 6e-8


def test_curriculum_tracker_golden():
    import ctypes as C

    cases = json.loads((GOLDEN / "curriculum_tracker.json").read_text())["cases"]
    L = orc.lib()
    for case in cases:
        t = orc.OTracker()
        L.kp1o_tracker_init(C.byref(t), case["threshold"], case["window"], case["min_episodes"], case["n_stages"] - 1, 0)
        promoted, stages, rates = [], [], []
        for i, s in enumerate(case["successes"]):
            if L.kp1o_tracker_record(C.byref(t), s):
                promoted.append(i)
                rates.append(t.last_trigger_rate)
            stages.append(t.stage_index)
        assert promoted == case["promoted_at"]
        assert stages == case["stage_after"]
        assert rates == case["trigger_rates"]


def test_reference_env_unit_behaviours():
    """The reference's own env unit tests (tests/test_kinematic_phase1_env.py:39-70), restated on the oracle."""
    cfg = kcfg.EnvConfig(c=kcfg.default_config())
    env = orc.OracleEnv(cfg)
    env.reset(seed=123)
    obs, info = env.step(np.full(7, 10.0))  # clipped state after an out-of-range action (:39-49)
    lo, hi = np.array(cfg.c.joints.lower[:]), np.array(cfg.c.joints.upper[:])
    assert np.all(info["q"] >= lo) and np.all(info["q"] <= hi)
    assert np.all(np.abs(obs) <= 1.0)
    # success termination from an exact-goal reset after success_dwell_steps zero actions (:51-60)
    env = orc.OracleEnv(cfg)
    q = np.array([0.0, 0.1, -0.1, 0.2, -0.2, 0.1, 0.0])
    env.reset(seed=5, options={"initial_q": q, "goal_q": q})
    done = False
    for _ in range(int(cfg.c.termination.success_dwell_steps)):
        _, info = env.step(np.zeros(7))
        done = info["terminated"]
    assert done and info["success"]
    # curriculum stage reset consistency goal_pose6 == FK(goal_q) to 1e-8 (:62-70)
    env = orc.OracleEnv(cfg)
    env.set_curriculum_stage(3)
    env.reset(seed=11)
    st = env.state()
    assert np.allclose(orc.fk_pose6(st["goal_q"])[0], st["goal_pose6"], atol=1e-8)
    with pytest.raises(ValueError):
        env.step(np.zeros(6))


def _reward_eval(case):
    """oracle compute_*_reward on one fixture case -> (total, {component: value})"""
    import ctypes as C

    mode = case["mode"]
    block = "reward" if mode == "approach" else "dock_reward"
    cfg = kcfg.to_env_config({"env": {"mode": mode, block: dict(case["config"])}})
    L = orc.lib()
    dp = C.POINTER(C.c_double)
    L.kp1o_reward_eval.restype = C.c_double
    L.kp1o_reward_eval.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp, dp, C.POINTER(C.c_int32), dp, dp, C.POINTER(C.c_int32)]
    i = case["inputs"]

    def arr(v):
        return (C.c_double * len(v))(*[float(x) for x in v])

    flags = (C.c_int32 * 7)(int(i["curr_in_pre_near_goal"]), int(i["prev_in_near_goal"]), int(i["curr_in_near_goal"]), int(i["dwell_count"]),
                            int(i["near_goal_entry_count"]), int(i["near_goal_drift_count"]), int(i["success"]))
    scalars = arr([i["joint_limit_margin_min"], i["dq_norm"], i["prev_dq_norm"], i["delta_q_change_l2"], i["entry_pos_error_norm"], i["entry_ori_error_norm"],
                   i["entry_action_l2"], i["entry_dq_norm"]])
    comps = (C.c_double * orc.MAX_COMPONENTS)()
    n = C.c_int32(0)
    mode_id = 1 if mode == "dock" else 0
    total = L.kp1o_reward_eval(C.byref(cfg.c), mode_id, arr(i["prev_pose6"]), arr(i["curr_pose6"]), arr(i["goal_pose6"]), arr(i["action"]), arr(i["prev_action"]),
                               flags, scalars, comps, C.byref(n))
    L.kp1o_component_name.restype = C.c_char_p
    names = [L.kp1o_component_name(mode_id, k).decode() for k in range(n.value)]
    return total, dict(zip(names, list(comps)[:n.value]))


def test_reward_functions_under_random_configs_golden():
    """compute_approach_reward / compute_dock_reward with EVERY config field randomised (tests/golden/make_golden_reward_fuzz.py ran the
    reference's functions): total and each of the 50 / 60 components to 1e-12, every component non-zero in some case -- the YAML configs
    of the step traces leave many of these weights at 0.  Plus the reference's own five reward unit tests with the constants they assert."""
    data = json.loads((GOLDEN / "reward_fuzz.json").read_text())
    seen = set()
    worst = 0.0
    for case in data["cases"] + data["reference_unit_cases"]:
        total, comps = _reward_eval(case)
        assert set(comps) == set(case["components"]), case["mode"]
        assert abs(total - case["reward"]) <= 1e-12 * max(1.0, abs(case["reward"])), (case["mode"], total, case["reward"])
        for k, v in case["components"].items():
            err = abs(comps[k] - v)
            worst = max(worst, err / max(1.0, abs(v)))
            assert err <= 1e-12 * max(1.0, abs(v)), (case["mode"], k, comps[k], v)
            if v != 0.0:
                seen.add((case["mode"], k))
        for k, v in case.get("expect", {}).items():
            assert comps[k] == v, (case.get("name"), k, comps[k], v)       # known answers of tests/test_kinematic_phase1_approach_reward.py
    all_names = {(c["mode"], k) for c in data["cases"] for k in c["components"]}
    assert seen == all_names
    u = {c["name"]: _reward_eval(c)[1] for c in data["reference_unit_cases"]}
    assert u["near_field_orientation_near"]["orientation_progress"] > u["near_field_orientation_far"]["orientation_progress"]
    assert u["near_field_orientation_near"]["near_field_orientation_progress"] > 0.0
    assert u["coarse_orientation_bonus_good"]["coarse_orientation_bonus"] > 0.0 and u["coarse_orientation_bonus_bad"]["coarse_orientation_bonus"] == 0.0
    assert u["drift_escalation_late"]["drift_penalty"] < u["drift_escalation_early"]["drift_penalty"]
    assert u["drift_escalation_late"]["drift_penalty_scale"] > u["drift_escalation_early"]["drift_penalty_scale"]
