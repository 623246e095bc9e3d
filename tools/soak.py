#!/usr/bin/env python3
"""GPU box: long free-running rollout of every env kind with random actions; counts non-finite outputs and episode statistics.
    python tools/soak.py [steps] [envs]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hrl_pybullet_envs_amd import _lib  # noqa: E402
from hrl_pybullet_envs_amd.vec_env import BatchedEnv  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
for kind, name in enumerate(['flat', 'gather', 'maze', 'point', 'maze_mj', 'flagrun']):
    env = BatchedEnv(_lib.default_config(kind, num_envs=n, seed=3, auto_reset=1), 'cuda:0')
    env.reset()
    gen = torch.Generator(device='cuda').manual_seed(kind)
    bad_obs = bad_rew = dones = 0
    ep_len_sum = 0.0
    vmax = 0.0
    for t in range(steps):
        a = torch.rand(n, env.act_dim, device='cuda', generator=gen) * 2 - 1
        obs, rew, done, info = env.step(a)
        bad_obs += int((~torch.isfinite(obs)).any(1).sum()); bad_rew += int((~torch.isfinite(rew)).sum())
        d = done.bool()
        dones += int(d.sum()); ep_len_sum += float(info['episode_length'][d].sum())
        if t % 500 == 0:
            vmax = max(vmax, float(env.state[:, 15:29].abs().max()))
    ok = bool(torch.isfinite(env.state).all())
    print(f'{name:8s} steps {steps} envs {n}: non-finite obs rows {bad_obs}, rewards {bad_rew}, episodes {dones}, '
          f'mean len {ep_len_sum / max(dones, 1):.1f}, max |qvel| sampled {vmax:.1f}, final state finite {ok}', flush=True)
