"""CPU-side tests: config plumbing, C-ABI exports (no compute without a GPU), host-side helpers."""
from __future__ import annotations

import ctypes as C
import json

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, load_golden_config
from rl_brain_trainer_amd import config as kcfg
from rl_brain_trainer_amd import native


def test_library_exports_every_declared_symbol():
    L = native.load()
    missing = [s for s in native.declared_symbols() if not hasattr(L, s)]
    assert not missing, missing
    assert L.kp1_config_size() == C.sizeof(kcfg.Kp1Config)


def test_library_default_config_matches_binding():
    L = native.load()
    c = kcfg.Kp1Config()
    assert L.kp1_config_default(C.byref(c)) == 0
    assert bytes(c) == bytes(kcfg.default_config())


def test_seed_state_matches_numpy():
    L = native.load()
    for seed in (0, 7, 806, 931, 2**32 + 5, 2**63 + 1):
        st = kcfg.RngState()
        assert L.kp1_rng_seed_state(seed, C.byref(st)) == 0
        ref = np.random.default_rng(seed).bit_generator.state["state"]
        assert (st.state_hi << 64 | st.state_lo) == ref["state"]
        assert (st.inc_hi << 64 | st.inc_lo) == ref["inc"]


def test_create_without_gpu_fails_loudly():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = native.load()
    h = C.c_void_p()
    rc = L.kp1_create(C.byref(kcfg.default_config()), 4, 0, 0, 0, 0, None, C.byref(h))
    assert rc == -2 and b"no HIP device" in L.kp1_last_error()
    from rl_brain_trainer_amd.vec_env import ArmKinematicVecEnv

    with pytest.raises(native.Kp1Error):
        ArmKinematicVecEnv(kcfg.EnvConfig(c=kcfg.default_config()), 4)


def test_unknown_reward_key_raises_like_reference_dataclass():
    cfg = json.loads((GOLDEN / "configs" / "approach_default.json").read_text())
    cfg["env"]["reward"]["not_a_field"] = 1.0
    with pytest.raises(TypeError):
        kcfg.to_env_config(cfg)
    cfg = json.loads((GOLDEN / "configs" / "approach_default.json").read_text())
    cfg["env"]["mode"] = "bridge"
    with pytest.raises(ValueError):
        kcfg.to_env_config(cfg)


def test_dwell_steps_target_quirk():
    # policy_config.py:146: dwell_steps_target comes from termination.success_dwell_steps
    cfg = load_golden_config("dock_workspace_handoff_noop_ft_12env_raw")
    assert cfg.c.env.dwell_steps_target == 5 and cfg.c.termination.success_dwell_steps == 5
    assert len(cfg.handoff_states) > 0 and cfg.c.dock_reset.handoff_state_probability == 0.95


BUILTIN = {
    "approach_default": ("workspace", None),
    "workspace_expansion_bigtrain": ("workspace", "workspace_expansion_bigtrain.yaml"),
    "workspace_expansion_1h_extend": ("workspace", "workspace_expansion_1h_extend.yaml"),
    "workspace_full_coverage_randomstart_overnight": ("workspace", "workspace_full_coverage_randomstart_overnight.yaml"),
    "dock_default": ("dock", None),
    # the fine-tuning chain behind the reference's published Stage-5 figure, and the Approach settle config
    "workspace_expansion_dynamic_scale_big": ("workspace", "workspace_expansion_dynamic_scale_big.yaml"),
    "workspace_expansion_late_stage_ft": ("workspace", "workspace_expansion_late_stage_ft.yaml"),
    "approach_finisher_ready_v2_settle": ("workspace", "approach_finisher_ready_v2_settle.yaml"),
    # the Approach policy's precision curriculum (3 cm -> 8 mm -> 5 mm finisher-ready)
    "approach_workspace_default": ("workspace", "approach_workspace_default.yaml"),
    "approach_workspace_handoff_ready_8mm_12env": ("workspace", "approach_workspace_handoff_ready_8mm_12env.yaml"),
    "approach_workspace_handoff_ready_8mm_ft_12env": ("workspace", "approach_workspace_handoff_ready_8mm_ft_12env.yaml"),
    "approach_dock_coarse_ready_v1": ("workspace", "approach_dock_coarse_ready_v1.yaml"),
    "approach_finisher_ready_v1": ("workspace", "approach_finisher_ready_v1.yaml"),
}
ROUTE_BUILTIN = ("route_curriculum_default", "route_curriculum_prefix20_sequence2", "route_curriculum_prefix120_routeobs_sequence2")


@pytest.mark.parametrize("name", ROUTE_BUILTIN)
def test_builtin_route_yaml_resolves_to_reference_config(name):
    """route configs: the whole merged dict (env, algorithms, route block) equals the reference's resolved config"""
    cfg = kcfg.load_workspace_expansion_config(kcfg.builtin_config_dir() / f"{name}.yaml")
    gold_cfg = json.loads((GOLDEN / "configs" / f"{name}.json").read_text())
    assert cfg == gold_cfg
    assert bytes(kcfg.to_env_config(cfg).c) == bytes(kcfg.to_env_config(gold_cfg).c)


def test_checkpoint_data_member_has_every_sb3_key():
    """the `data` member of a written zip must carry every key stable-baselines3 2.8 saves, JSON-typed, and say in a
    machine-readable way that the pickled members still need tools/finish_sb3_zip.py (PPO.load compatibility itself is unpinned: SB3 is
    not importable here)."""
    from types import SimpleNamespace

    from rl_brain_trainer_amd import checkpoint as ck
    from rl_brain_trainer_amd.ppo import PPOConfig

    fake = SimpleNamespace(cfg=PPOConfig(hidden=64, n_epochs=8), n_envs=12, dist=SimpleNamespace(world_size=1), num_timesteps=24576, adam_t=384,
                           n_train_calls=1, obs_dim=56)
    data = ck._policy_data(fake, None)
    assert set(ck.SB3_PPO_DATA_KEYS) <= set(data)
    json.loads(json.dumps(data))                                   # JSON-typed all the way down
    assert data["_n_updates"] == 8 and data["kp1_engine"]["adam_steps"] == 384     # SB3 counts n_epochs per train() call, not Adam steps
    assert data["policy_kwargs"] == {} and data["kp1_engine"]["sb3_loadable"] is False
    # the six pickled members: payloads assembled without gymnasium / SB3; read back here from the OPCODE stream only (nothing is unpickled:
    # the classes they name are not importable in this image) -- the GLOBALs must be exactly the classes SB3's loader expects
    from rl_brain_trainer_amd import sb3_pickle as sbp

    refs = {k: sbp.describe(data[k][":serialized:"]) for k in ck.PICKLED_MEMBERS}
    assert refs["policy_class"] == [("stable_baselines3.common.policies", "MultiInputActorCriticPolicy")]
    assert refs["rollout_buffer_class"] == [("stable_baselines3.common.buffers", "DictRolloutBuffer")]
    for k in ("lr_schedule", "clip_range"):
        assert set(refs[k]) == {("stable_baselines3.common.utils", "FloatSchedule"), ("stable_baselines3.common.utils", "ConstantSchedule")}
    assert ("gymnasium.spaces.dict", "Dict") in refs["observation_space"] and ("gymnasium.spaces.box", "Box") in refs["observation_space"]
    assert ("gymnasium.spaces.box", "Box") in refs["action_space"]
    assert all(m.split(".")[0] in ("gymnasium", "stable_baselines3", "numpy", "copy_reg", "copyreg", "_codecs") for v in refs.values() for m, _ in v)
    fake.cfg = PPOConfig(hidden=256)
    assert ck._policy_data(fake, None)["policy_kwargs"] == {"net_arch": {"pi": [256, 256], "vf": [256, 256]}}


@pytest.mark.parametrize("name", sorted(BUILTIN))
def test_builtin_yaml_resolves_to_reference_config(name):
    """Our own YAML files + loaders must resolve to the same env struct as the reference's resolved config."""
    kind, fname = BUILTIN[name]
    path = kcfg.builtin_config_dir() / fname if fname else None
    cfg = kcfg.load_workspace_expansion_config(path) if kind == "workspace" else kcfg.load_dock_config(path)
    ours = kcfg.to_env_config(cfg, handoff_base_dirs=(GOLDEN,))
    gold = load_golden_config(name)
    assert bytes(ours.c) == bytes(gold.c)
    assert ours.stage_names == gold.stage_names
    gold_cfg = json.loads((GOLDEN / "configs" / f"{name}.json").read_text())
    assert kcfg.to_algorithm_kwargs(cfg) == kcfg.to_algorithm_kwargs(gold_cfg)


def test_finisher_yaml_resolves_to_reference_config():
    raw = kcfg.load_yaml_file(kcfg.builtin_config_dir() / "dock_workspace_handoff_noop_ft_12env.yaml")
    raw["env"]["dock_reset"]["handoff_state_buffer_path"] = "handoff_state_buffer.json"
    ours = kcfg.to_env_config(raw, handoff_base_dirs=(GOLDEN,))
    gold = load_golden_config("dock_workspace_handoff_noop_ft_12env_raw")
    assert bytes(ours.c) == bytes(gold.c)


def test_sb3_pickle_payloads_rebuild_the_documented_state(monkeypatch):
    """The payloads of rl_brain_trainer_amd/sb3_pickle.py are well-formed pickle streams that rebuild, under modules of the real names, objects
    with the attribute state the module documents.  The modules here are EMPTY stand-ins created by this test (gymnasium / SB3 are not
    importable in this image: what the real classes do with that state is the unpinned part); the streams are this build's own output."""
    import base64
    import pickle
    import sys
    import types

    import numpy as np

    from rl_brain_trainer_amd import sb3_pickle as sbp

    for module, names in (("gymnasium.spaces.box", ("Box",)), ("gymnasium.spaces.dict", ("Dict",)),
                          ("stable_baselines3.common.utils", ("FloatSchedule", "ConstantSchedule")),
                          ("stable_baselines3.common.policies", ("MultiInputActorCriticPolicy",)), ("stable_baselines3.common.buffers", ("DictRolloutBuffer",))):
        parts = module.split(".")
        for k in range(1, len(parts) + 1):
            name = ".".join(parts[:k])
            if name not in sys.modules:
                monkeypatch.setitem(sys.modules, name, types.ModuleType(name))
        for cls in names:
            setattr(sys.modules[module], cls, type(cls, (), {"__module__": module}))
    load = lambda payload: pickle.loads(base64.b64decode(payload))   # noqa: E731
    space = load(sbp.serialized(sbp.dict_space({"b_key": sbp.box(-1.0, 1.0, 7), "a_key": sbp.box(0.0, 1.0, 2)})))
    assert type(space).__name__ == "Dict" and list(space.spaces) == ["a_key", "b_key"] and space._shape is None and space._np_random is None
    b = space.spaces["b_key"]
    assert type(b).__module__ == "gymnasium.spaces.box" and b._shape == (7,) and b.dtype == np.float32
    assert b.low.dtype == np.float32 and np.array_equal(b.low, -np.ones(7, np.float32)) and np.array_equal(b.high, np.ones(7, np.float32))
    assert b.bounded_below.all() and b.bounded_above.all() and b.low_repr == "-1.0" and space.spaces["a_key"].low_repr == "0.0"
    sched = load(sbp.serialized(sbp.float_schedule(3e-4)))
    assert type(sched).__name__ == "FloatSchedule" and type(sched.value_schedule).__name__ == "ConstantSchedule" and sched.value_schedule.val == 3e-4
    assert load(sbp.serialized(sbp.POLICY_CLASS)) is sys.modules["stable_baselines3.common.policies"].MultiInputActorCriticPolicy
    assert load(sbp.serialized(sbp.ROLLOUT_BUFFER_CLASS)) is sys.modules["stable_baselines3.common.buffers"].DictRolloutBuffer


def test_generated_fk_chain_is_current():
    """csrc/kp1_fk_generated.inc (the fp32 handle's FK chain with the robot constants as literals) is what tools/gen_fk_chain.py produces from the
    constants in kp1_env.hip today; kp1_create additionally compares the constants with fold_fk's at run time, bit for bit."""
    import subprocess
    import sys

    from conftest import ROOT

    out = subprocess.run([sys.executable, str(ROOT / "tools" / "gen_fk_chain.py"), "--check"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
