#!/usr/bin/env python3
"""GPU box: host time of one BatchedEnv.step() call (the launch is asynchronous: the loop below runs ahead of the GPU, so the wall time of the
calls alone, before any synchronisation, is what the interpreter, torch and the ctypes call cost).    python tools/host_overhead.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hrl_pybullet_envs_amd import _lib  # noqa: E402
from hrl_pybullet_envs_amd.vec_env import BatchedEnv  # noqa: E402

env = BatchedEnv(_lib.default_config(1, num_envs=64, seed=0, auto_reset=1), 'cuda:0')
env.reset()
a = torch.rand(64, 8, device='cuda') * 2 - 1
for _ in range(200):
    env.step(a)
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    t0 = time.perf_counter()
    for _ in range(300):      # 300 launches of 40 us queue up without blocking
        env.step(a)
    best = min(best, (time.perf_counter() - t0) * 1e6 / 300)
    torch.cuda.synchronize()
print(f'host time of BatchedEnv.step(): {best:.1f} us per call (64 envs; the kernel lasts ~40 us and runs behind)')
