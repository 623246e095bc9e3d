#!/usr/bin/env python3
"""GPU box: a rollout's inner loop -- observation -> policy (a small MLP in torch) -> action -> env step -- eagerly and as ONE captured hipGraph of K
iterations (torch.cuda.graphs).  hrl_step is a plain launch on the caller's stream, so the loop captures as it stands; what the graph removes
is the host time per iteration (a handful of torch launches + the ctypes call), which at small env counts is longer than the GPU work.
    python tools/graph_rollout.py [envs ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hrl_pybullet_envs_amd import _lib  # noqa: E402
from hrl_pybullet_envs_amd.vec_env import BatchedEnv  # noqa: E402

K_ITERS, REPS = 16, 40
sizes = [int(a) for a in sys.argv[1:]] or [256, 1024, 4096]
for n in sizes:
    env = BatchedEnv(_lib.default_config(1, num_envs=n, seed=0, auto_reset=1), 'cuda:0')
    obs = env.reset()
    torch.manual_seed(0)
    policy = torch.nn.Sequential(torch.nn.Linear(env.obs_dim, 256), torch.nn.Tanh(), torch.nn.Linear(256, 256), torch.nn.Tanh(),
                                 torch.nn.Linear(256, env.act_dim), torch.nn.Tanh()).cuda()
    ret = torch.zeros(n, device='cuda')

    def iteration():
        with torch.no_grad():
            a = policy(env.obs)
        _, rew, _, _ = env.step(a)
        ret.add_(rew)

    for _ in range(300):   # settle
        iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K_ITERS * REPS):
        iteration()
    torch.cuda.synchronize()
    eager_us = (time.perf_counter() - t0) * 1e6 / (K_ITERS * REPS)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        iteration()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(K_ITERS):
            iteration()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(REPS):
        g.replay()
    torch.cuda.synchronize()
    graph_us = (time.perf_counter() - t0) * 1e6 / (K_ITERS * REPS)
    print(f'AntGather, {n} envs, policy MLP {env.obs_dim}-256-256-{env.act_dim} + env step: eager {eager_us:.1f} us per iteration ({n / eager_us:.1f} M env-steps/s), '
          f'one hipGraph of {K_ITERS} iterations {graph_us:.1f} us ({n / graph_us:.1f} M env-steps/s)', flush=True)
    del g
    env.close()
