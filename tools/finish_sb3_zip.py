#!/usr/bin/env python3
"""Re-save an engine checkpoint through Stable-Baselines3 itself (run where gymnasium + stable-baselines3==2.8.0 exist).

The MI355X engine writes SB3-layout zips whose tensors are exact but whose cloudpickled ``observation_space`` /
``action_space`` / schedule entries cannot be produced without gymnasium + SB3 (absent from the build image).  This tool
builds a real ``PPO("MultiInputPolicy")`` with the same ``net_arch``, loads ``policy.pth`` / ``policy.optimizer.pth``
from the engine zip and calls ``model.save`` -- after which ``PPO.load`` / the Gazebo demo loaders work unchanged.

    python tools/finish_sb3_zip.py engine_model.zip out_model.zip
"""
import io
import json
import sys
import zipfile

import numpy as np
import torch


def main(src: str, dst: str) -> None:
    import gymnasium as gym
    from gymnasium import spaces
    from stable_baselines3 import PPO

    with zipfile.ZipFile(src) as z:
        data = json.loads(z.read("data"))
        policy_sd = torch.load(io.BytesIO(z.read("policy.pth")), map_location="cpu", weights_only=True)
        opt_sd = torch.load(io.BytesIO(z.read("policy.optimizer.pth")), map_location="cpu", weights_only=True)

    def box(lo, n):
        return spaces.Box(low=lo, high=1.0, shape=(n,), dtype=np.float32)

    obs_space = spaces.Dict({k: box(v["low"], v["shape"][0]) for k, v in data["observation_space"]["spaces"].items()})

    class _Stub(gym.Env):
        observation_space = obs_space
        action_space = spaces.Box(low=-1.0, high=1.0, shape=(7,), dtype=np.float32)

        def reset(self, *, seed=None, options=None):
            return {k: np.zeros(s.shape, np.float32) for k, s in obs_space.spaces.items()}, {}

        def step(self, action):
            return self.reset()[0], 0.0, False, True, {}

    kwargs = {k: data[k] for k in ("n_steps", "batch_size", "n_epochs", "gamma", "gae_lambda", "ent_coef", "vf_coef", "max_grad_norm", "learning_rate")}
    model = PPO("MultiInputPolicy", _Stub(), policy_kwargs=data["policy_kwargs"] or None, clip_range=data["clip_range"]["value"], seed=data["seed"], **kwargs)
    model.policy.load_state_dict(policy_sd)
    model.policy.optimizer.load_state_dict(opt_sd)
    model.num_timesteps = data["num_timesteps"]
    model.save(dst)
    print("wrote", dst)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
