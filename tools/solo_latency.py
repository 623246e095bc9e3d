#!/usr/bin/env python3
"""GPU box: one env (BASELINE config 1) -- launch time of the step kernel and wall time of step_host() for both workgroup shapes
(model.step_group 0: four env-waves per workgroup with a lane-packed leader; 1: one wave per env doing everything itself).
    python tools/solo_latency.py [kind] [envs]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hrl_pybullet_envs_amd import _lib  # noqa: E402
from hrl_pybullet_envs_amd.vec_env import BatchedEnv  # noqa: E402

kind = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for group in (0, 1):
    env = BatchedEnv(_lib.default_config(kind, num_envs=n, seed=0, auto_reset=1, model_step_group=group), 'cuda:0')
    env.reset()
    acts = torch.rand(512, n, env.act_dim, device='cuda') * 2 - 1
    for t in range(300):
        env.step(acts[t % 512])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(2000):
        env.step(acts[t % 512])
    e1.record(); torch.cuda.synchronize()
    dev_us = e0.elapsed_time(e1) * 1e3 / 2000
    a_np = acts.cpu().numpy()
    for t in range(100):
        env.step_host(a_np[t % 512])
    t0 = time.perf_counter()
    for t in range(2000):
        env.step_host(a_np[t % 512])
    host_us = (time.perf_counter() - t0) * 1e6 / 2000
    print(f'kind {kind} envs {n} step_group {group}: {dev_us:.1f} us per back-to-back launch, step_host {host_us:.1f} us per step ({1e6 / host_us:.0f} steps/s)', flush=True)
    env.close()
