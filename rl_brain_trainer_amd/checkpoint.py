"""Policy checkpoints in the Stable-Baselines3 ``.zip`` layout.

The Gazebo/RViz demo and every evaluator of the reference load policies with ``PPO.load(path)``
(kinematic_phase1/eval/eval_deterministic.py:66-79, v5/phase3a_controlled_sim.py:94-103) and call
``model.predict(obs_dict, deterministic=True)``.  An SB3 archive (pin: stable-baselines3==2.8.0,
final_codes_docker/Dockerfile.demo:32) contains

    data                         JSON; non-JSON values are {":type:", ":serialized:" = base64(cloudpickle)}
    policy.pth                   torch state_dict (log_std, mlp_extractor.{policy,value}_net.{0,2}.{weight,bias},
                                 action_net.*, value_net.*); first-layer columns in alphabetical Dict-key order
    policy.optimizer.pth         torch Adam state_dict
    pytorch_variables.pth        {} for PPO
    _stable_baselines3_version   text
    system_info.txt              text

``save`` writes all six members.  The tensors are exact and loadable by anything that reads SB3 state dicts.
The cloudpickled ``observation_space`` / ``action_space`` / schedules inside ``data`` can only be produced where
gymnasium + SB3 are importable, which is not the case in this image (and SB3 is absent from the reference tree):
those entries are written as plain descriptions and ``tools/finish_sb3_zip.py`` re-saves the archive through SB3's
own ``PPO.save`` on a box that has it.  Until that has been done on real SB3, ``PPO.load`` compatibility is
UNVERIFIED -- this module never claims otherwise.

``load_policy_state_dict`` reads ``policy.pth`` from any SB3 zip (reference-trained 2x64 checkpoints included) with
``torch.load(weights_only=True)``: nothing from the archive is executed.
"""
from __future__ import annotations

import io
import json
import platform
import zipfile
from pathlib import Path
from typing import Any

import torch

from . import config as kcfg
from . import sb3_pickle as sbp

SB3_VERSION_PIN = "2.8.0"


def _policy_data(ppo, env_cfg: kcfg.EnvConfig | None) -> dict[str, Any]:
    cfg = ppo.cfg
    H = cfg.hidden
    layout = kcfg.OBS_LAYOUT
    if getattr(ppo, "obs_dim", kcfg.OBS_DIM) != kcfg.OBS_DIM:   # route wrappers with include_route_keys (route_observation.py:18-28)
        from . import route_config as rcfg

        layout = rcfg.ROUTE_OBS_LAYOUT
    obs_space = {k: {"shape": [n], "low": (0.0 if k in ("task_type", "mode_flag", "progress", "joint_limit_margin", "route_scalar") else -1.0), "high": 1.0,
                     "dtype": "float32"} for k, (_, n) in layout.items()}
    # the six members SB3 restores by unpickling: pickle streams assembled without gymnasium / SB3 (sb3_pickle.py; PARITY UNPINNED, see there)
    obs_pickle = sbp.serialized(sbp.dict_space({k: sbp.box(v["low"], v["high"], v["shape"][0]) for k, v in obs_space.items()}))
    return {
        "policy_class": {":type:": "<class 'abc.ABCMeta'>", "__module__": "stable_baselines3.common.policies",
                         "__name__": "MultiInputActorCriticPolicy", ":serialized:": sbp.serialized(sbp.POLICY_CLASS)},
        "policy_kwargs": {} if H == 64 else {"net_arch": {"pi": [H, H], "vf": [H, H]}},
        "observation_space": {":type:": "<class 'gymnasium.spaces.dict.Dict'>", ":serialized:": obs_pickle, "spaces": obs_space},
        "action_space": {":type:": "<class 'gymnasium.spaces.box.Box'>", ":serialized:": sbp.serialized(sbp.box(-1.0, 1.0, kcfg.NJ)), "shape": [kcfg.NJ],
                         "low": -1.0, "high": 1.0, "dtype": "float32"},
        "n_envs": int(ppo.n_envs * ppo.dist.world_size),
        "num_timesteps": int(ppo.num_timesteps),
        "_total_timesteps": int(cfg.total_timesteps),
        "seed": int(cfg.seed),
        "learning_rate": float(cfg.learning_rate),
        "lr_schedule": {":type:": "<class 'stable_baselines3.common.utils.FloatSchedule'>", ":serialized:": sbp.serialized(sbp.float_schedule(cfg.learning_rate)),
                        "value": float(cfg.learning_rate)},
        "clip_range": {":type:": "<class 'stable_baselines3.common.utils.FloatSchedule'>", ":serialized:": sbp.serialized(sbp.float_schedule(cfg.clip_range)),
                       "value": float(cfg.clip_range)},
        "n_steps": int(cfg.n_steps), "batch_size": int(cfg.batch_size), "n_epochs": int(cfg.n_epochs),
        "gamma": float(cfg.gamma), "gae_lambda": float(cfg.gae_lambda), "ent_coef": float(cfg.ent_coef), "vf_coef": float(cfg.vf_coef),
        "max_grad_norm": float(cfg.max_grad_norm), "normalize_advantage": bool(cfg.normalize_advantage),
        "use_sde": False, "sde_sample_freq": -1, "target_kl": None, "clip_range_vf": None,
        # SB3: self._n_updates += self.n_epochs once per train() call (ppo.py train()); the Adam step count lives in policy.optimizer.pth
        "_n_updates": int(getattr(ppo, "n_train_calls", 0)) * int(cfg.n_epochs),
        # the rest of BaseAlgorithm / OnPolicyAlgorithm.__dict__ as SB3 2.8 saves it (values a freshly constructed model holds)
        "device": "auto", "verbose": 0, "_num_timesteps_at_start": 0, "action_noise": None, "start_time": 0, "tensorboard_log": None,
        "_last_obs": None, "_last_episode_starts": None, "_last_original_obs": None, "_episode_num": 0, "_current_progress_remaining": 1.0,
        "_stats_window_size": 100, "ep_info_buffer": None, "ep_success_buffer": None,
        "rollout_buffer_class": {":type:": "<class 'abc.ABCMeta'>", "__module__": "stable_baselines3.common.buffers", "__name__": "DictRolloutBuffer",
                                 ":serialized:": sbp.serialized(sbp.ROLLOUT_BUFFER_CLASS)},
        "rollout_buffer_kwargs": {},
        "kp1_engine": {"writer": "rl_brain_trainer_amd.checkpoint", "sb3_loadable": False,
                       "needs": "the pickled members are assembled without gymnasium / SB3 (rl_brain_trainer_amd/sb3_pickle.py) and have never been read by "
                                "the real PPO.load: until one has, treat tools/finish_sb3_zip.py on a host with stable-baselines3==" + SB3_VERSION_PIN +
                                " as the route that is correct by construction",
                       "adam_steps": int(ppo.adam_t), "mode": env_cfg.mode_name if env_cfg else None},
    }


# every key stable-baselines3 2.8.0 writes into the `data` member of a PPO("MultiInputPolicy") archive (BaseAlgorithm.save: __dict__ minus the
# excluded attributes); tests/test_host_logic.py checks the writer covers them with JSON-typed values
SB3_PPO_DATA_KEYS = (
    "policy_class", "device", "verbose", "policy_kwargs", "num_timesteps", "_total_timesteps", "_num_timesteps_at_start", "seed", "action_noise",
    "start_time", "learning_rate", "tensorboard_log", "_last_obs", "_last_episode_starts", "_last_original_obs", "_episode_num", "use_sde",
    "sde_sample_freq", "_current_progress_remaining", "_stats_window_size", "ep_info_buffer", "ep_success_buffer", "_n_updates", "observation_space",
    "action_space", "n_envs", "n_steps", "gamma", "gae_lambda", "ent_coef", "vf_coef", "max_grad_norm", "rollout_buffer_class", "rollout_buffer_kwargs",
    "batch_size", "n_epochs", "clip_range", "clip_range_vf", "normalize_advantage", "target_kl", "lr_schedule",
)
PICKLED_MEMBERS = ("policy_class", "observation_space", "action_space", "lr_schedule", "clip_range", "rollout_buffer_class")


def sb3_loadable(path: str | Path) -> bool:
    """True when every entry SB3's loader unpickles (policy class, spaces, schedules) carries its cloudpickle payload, i.e. the archive was
    written by SB3 itself or completed by tools/finish_sb3_zip.py.  Archives straight out of ``save`` here return False: they hold exact
    tensors and plain descriptions, readable by this package's loaders, but ``stable_baselines3.PPO.load`` would stop at the first
    ``":serialized:": null``."""
    data = load_data(path)
    return all(isinstance(data.get(k), dict) and isinstance(data[k].get(":serialized:"), str) for k in PICKLED_MEMBERS if k in data) and \
        not (isinstance(data.get("kp1_engine"), dict) and data["kp1_engine"].get("sb3_loadable") is False)


def require_sb3_loadable(path: str | Path) -> Path:
    """for hand-over points where the consumer is SB3 itself (the Gazebo / RViz demo, final_codes_docker/model_manifest.yaml): refuse an
    archive that still needs tools/finish_sb3_zip.py instead of letting PPO.load fail on the demo box"""
    if not sb3_loadable(path):
        raise ValueError(f"{path}: written by the MI355X engine and not yet completed for stable_baselines3.PPO.load -- run "
                         f"tools/finish_sb3_zip.py on a host with stable-baselines3=={SB3_VERSION_PIN} (SB3 compatibility is unverified in this image)")
    return Path(path)


def _torch_bytes(obj: Any) -> bytes:
    buf = io.BytesIO()
    torch.save(obj, buf)
    return buf.getvalue()


def optimizer_state_dict(ppo) -> dict[str, Any]:
    """torch.optim.Adam.state_dict() of the flat Adam moments, one entry per SB3 parameter (policy.parameters() order)."""
    state, off = {}, 0
    extra = int(getattr(ppo, "actor_extra_steps", 0))   # teacher-anchor steps reach only the actor tensors (torch counts per tensor)
    for i, (name, shape) in enumerate(ppo.policy.spec):
        n = 1
        for d in shape:
            n *= d
        step = torch.tensor(float(ppo.adam_t + (extra if name.startswith(("mlp_extractor.policy_net", "action_net")) else 0)))
        state[i] = {"step": step.clone(), "exp_avg": ppo.adam_m[off:off + n].view(shape).detach().cpu().clone(),
                    "exp_avg_sq": ppo.adam_v[off:off + n].view(shape).detach().cpu().clone()}
        off += n
    group = {"lr": float(ppo.cfg.learning_rate), "betas": (0.9, 0.999), "eps": float(ppo.cfg.adam_eps), "weight_decay": 0, "amsgrad": False,
             "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
             "params": list(range(len(ppo.policy.spec)))}
    return {"state": state, "param_groups": [group]}


def save(path: str | Path, ppo, env_cfg: kcfg.EnvConfig | None = None) -> Path:
    """model.save(path): writes ``<path>.zip`` (SB3 appends the suffix the same way)."""
    path = Path(path)
    if path.suffix != ".zip":
        path = path.with_name(path.name + ".zip")
    path.parent.mkdir(parents=True, exist_ok=True)
    with zipfile.ZipFile(path, "w", compression=zipfile.ZIP_DEFLATED) as z:
        z.writestr("data", json.dumps(_policy_data(ppo, env_cfg), indent=4))
        z.writestr("pytorch_variables.pth", _torch_bytes({}))
        z.writestr("policy.pth", _torch_bytes(ppo.policy.state_dict()))
        z.writestr("policy.optimizer.pth", _torch_bytes(optimizer_state_dict(ppo)))
        z.writestr("_stable_baselines3_version", SB3_VERSION_PIN)
        z.writestr("system_info.txt", f"- OS: {platform.platform()}\n- Python: {platform.python_version()}\n- PyTorch: {torch.__version__}\n"
                                      f"- Writer: rl_brain_trainer_amd (MI355X engine); finish with tools/finish_sb3_zip.py for SB3 {SB3_VERSION_PIN}\n")
    return path


def load_policy_state_dict(path: str | Path) -> dict[str, torch.Tensor]:
    """policy.pth of an SB3 zip (or a bare .pth) as a plain tensor dict; safe loader only."""
    path = Path(path)
    if path.suffix != ".zip" and not path.exists():
        path = path.with_name(path.name + ".zip")
    if path.suffix == ".zip":
        with zipfile.ZipFile(path) as z:
            raw = z.read("policy.pth")
        return torch.load(io.BytesIO(raw), map_location="cpu", weights_only=True)
    return torch.load(path, map_location="cpu", weights_only=True)


def load_optimizer_state_dict(path: str | Path) -> dict[str, Any] | None:
    """policy.optimizer.pth of an SB3 zip (torch.optim.Adam.state_dict()), safe loader only; None when the member is missing."""
    path = Path(path)
    if path.suffix != ".zip":
        path = path.with_name(path.name + ".zip")
    with zipfile.ZipFile(path) as z:
        if "policy.optimizer.pth" not in z.namelist():
            return None
        raw = z.read("policy.optimizer.pth")
    return torch.load(io.BytesIO(raw), map_location="cpu", weights_only=True)


def load_data(path: str | Path) -> dict[str, Any]:
    path = Path(path)
    if path.suffix != ".zip":
        path = path.with_name(path.name + ".zip")
    with zipfile.ZipFile(path) as z:
        return json.loads(z.read("data"))


def hidden_from_state_dict(sd: dict[str, torch.Tensor]) -> int:
    return int(sd["mlp_extractor.policy_net.0.weight"].shape[0])


def hidden_for_run(requested: int, resume_path: str | Path | None) -> int:
    """Width of the actor-critic for a trainer run: the checkpoint's when the run resumes from one (``PPO.load`` rebuilds the policy the zip
    describes -- a reference-trained archive is SB3's default 2x64, train_workspace_expansion.py:187-199), else the requested width."""
    if resume_path and Path(resume_path).exists():
        hidden = hidden_from_state_dict(load_policy_state_dict(resume_path))
        if hidden != int(requested):
            print(f"[checkpoint] {resume_path} holds a 2x{hidden} policy: using that width instead of --hidden {requested}")
        return hidden
    return int(requested)

