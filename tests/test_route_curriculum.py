"""Route curriculum host logic (config defaults, gate decisions: CPU) and the sequential route evaluator (GPU) against the reference's outputs in
tests/golden/route_eval.json / route_gate.json.  The prefix-curriculum callback itself runs on the device: its replay of the reference
callback's recorded stream is tests/test_route_ppo_gpu.py::test_device_prefix_curriculum_matches_reference_callback."""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np
import pytest

from conftest import GOLDEN
from rl_brain_trainer_amd import route_config as rcfg


@pytest.fixture(scope="module")
def gold():
    return json.loads((GOLDEN / "route_eval.json").read_text())


def test_prefix_stages_and_config_defaults():
    from rl_brain_trainer_amd.route_curriculum import RoutePrefixCurriculumDevice

    cfg = json.loads((GOLDEN / "configs" / "route_curriculum_prefix120_routeobs_sequence2.json").read_text())
    assert rcfg.prefix_stages(cfg, 484) == [120]
    assert rcfg.prefix_stages({}, 484) == [20, 40, 80, 120, 180, 260, 360, 483]
    cb = RoutePrefixCurriculumDevice.from_config(cfg, 484)      # host handle only: nothing is allocated before attach()
    assert [s.prefix_end_index for s in cb.stages] == [120]
    assert cb.window_episodes == 256 and cb.min_episodes_per_stage == 1024 and cb.promotion_max_regression_rate == 0.30
    with pytest.raises(TypeError):
        rcfg.route_config_from_dict({"route": {"reward": {"nope": 1.0}}})


@pytest.mark.gpu
def test_sequential_route_evaluator_matches_reference(tmp_path, gold):
    """The reference's _roll_one loop with a scripted servo model vs the device evaluator with the same controller (fp64 env)."""
    import torch

    from rl_brain_trainer_amd.route_curriculum import evaluate_sequential_route

    cfg = json.loads((GOLDEN / "configs" / "route_curriculum_prefix120_routeobs_sequence2.json").read_text())
    route_q = rcfg.load_route_q(GOLDEN / "synthetic_route.json")
    for case in gold["sequential"]:
        gain = case["gain"]

        def make_policy(env, gain=gain):
            dl = torch.tensor(env.config.c.joints.delta_limit[:], device="cuda", dtype=torch.float64) * env.config.c.env.action_delta_scale
            rq = torch.tensor(route_q, device="cuda", dtype=torch.float64)

            def policy(obs):
                info = env.info()
                goal = rq[info["route_index"].long()]
                return (gain * (goal - info["q"].double().t()) / dl).clamp(-1, 1)

            return policy

        out = evaluate_sequential_route(policy=make_policy, policy_needs_env=True, cfg=cfg, route_q=route_q, artifact_root=tmp_path / f"g{gain}",
                                        start_index=case["start_index"], end_index=case["end_index"], real="f64")
        assert len(out["rows"]) == len(case["rows"])
        for mine, ref in zip(out["rows"], case["rows"]):
            for k, v in ref.items():
                if isinstance(v, float):
                    assert abs(mine[k] - v) <= 1e-9 * max(1.0, abs(v)), (case["gain"], ref["route_index"], k, mine[k], v)
                else:
                    assert mine[k] == v, (case["gain"], ref["route_index"], k, mine[k], v)
        for k, v in case["summary"].items():
            if isinstance(v, float):
                assert abs(out[k] - v) <= 1e-9 * max(1.0, abs(v)), k
            else:
                assert out[k] == v, k
        cm = json.loads(json.dumps(out["chunk_metrics"]))
        assert cm.keys() == case["chunk_metrics"].keys()
        for name, ref_chunk in case["chunk_metrics"].items():
            assert cm[name].keys() == ref_chunk.keys() and cm[name]["target_count"] == ref_chunk["target_count"]
            assert all(abs(cm[name][k] - v) <= 1e-9 * max(1.0, abs(v)) for k, v in ref_chunk.items())
        assert np.max(np.abs(np.array(out["final_q"]) - np.array(case["final_q"]))) <= 1e-12
        assert (tmp_path / f"g{gain}" / "route_eval_sequential_summary.json").exists()


def test_route_gate_matches_reference_decisions(tmp_path):
    """eval_route_gate.evaluate_route_gate's accept / reject rules on 60 scripted sets of per-prefix summaries (the fixture was
    produced by the reference's own function with its evaluator replaced by the same scripted summaries)."""
    from rl_brain_trainer_amd.route_curriculum import evaluate_route_gate

    cases = json.loads((GOLDEN / "route_gate.json").read_text())["cases"]
    assert sum(c["accepted"] for c in cases) > 10 and sum(not c["accepted"] for c in cases) > 10
    for n, case in enumerate(cases):
        calls = []

        def evaluate(*, artifact_root, start_index, end_index, case=case, calls=calls):
            calls.append({"artifact_dir": Path(artifact_root).name, "start_index": int(start_index), "end_index": int(end_index)})
            return dict(case["summaries"][str(int(end_index))])

        out = evaluate_route_gate(evaluate=evaluate, artifact_root=tmp_path / f"g{n}", prefixes=case["prefixes"], full_end_index=case["full_end_index"],
                                  **case["criteria"])
        assert calls == case["calls"]
        assert out["accepted"] == case["accepted"] and out["rejection_reasons"] == case["rejection_reasons"], n
        assert out["schema_version"] == case["schema_version"] and sorted(out.keys()) == case["result_keys"]
        assert json.loads((tmp_path / f"g{n}" / "route_gate_summary.json").read_text())["accepted"] == case["accepted"]


def test_teacher_anchor_batch_order_and_flatten():
    """The anchor batches are default_rng(0).integers(0, M, size=B) draws in call order (teacher_anchor.py:41, 62-69) and the Dict
    observation is flattened in sorted-key order (SB3 CombinedExtractor) -- both host-only."""
    from rl_brain_trainer_amd import route_config as rc
    from rl_brain_trainer_amd.teacher_anchor import RouteTeacherAnchor, TeacherAnchorConfig, flatten_observation

    a = RouteTeacherAnchor(TeacherAnchorConfig(enabled=True, dataset_path="x.npz", batch_size=7))
    import torch

    a._actions = torch.zeros((50, 7))
    ref = np.random.default_rng(0)
    for _ in range(5):
        assert np.array_equal(a.sample_indices(), ref.integers(0, 50, size=7))
    rng = np.random.default_rng(3)
    obs = {k: rng.random((4, w)).astype(np.float32) for k, (_, w) in rc.ROUTE_OBS_LAYOUT.items()}
    flat = flatten_observation(obs, rc.ROUTE_OBS_DIM)
    assert flat.shape == (4, 80)
    assert np.array_equal(flat, np.concatenate([obs[k] for k in sorted(obs)], axis=1))
    with pytest.raises(ValueError):
        flatten_observation({k: v for k, v in obs.items() if k != "route_tangent"}, rc.ROUTE_OBS_DIM)
    with pytest.raises(ValueError):
        RouteTeacherAnchor(TeacherAnchorConfig(enabled=True))
