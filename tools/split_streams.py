#!/usr/bin/env python3
"""GPU box: the same 4096 AntGather envs as K sub-shards, each with its own handle and HIP stream, stepped open-loop side by side (the launches of different
sub-shards overlap; a sub-shard's step t + 1 waits only for its own step t).  What hiding one launch's tail behind another's body is worth.
    python tools/split_streams.py [envs] [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hrl_pybullet_envs_amd import _capi as K  # noqa: E402
from hrl_pybullet_envs_amd import _lib  # noqa: E402
from hrl_pybullet_envs_amd.vec_env import BatchedEnv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
dev = torch.device('cuda', 0)
res = {}
for rnd in range(3):
    for k in (1, 2, 4, 8):
        envs = []
        for i in range(k):
            env = BatchedEnv(_lib.default_config(K.HRL_ANT_GATHER, num_envs=n // k, seed=0, auto_reset=1, env_id_offset=i * (n // k)), dev)
            env.reset()
            acts = torch.rand(64, n // k, 8, device=dev) * 2 - 1
            envs.append((env, acts, torch.cuda.Stream(device=dev)))
        def run(t0, cnt):
            for t in range(t0, t0 + cnt):
                for env, acts, st in envs:
                    with torch.cuda.stream(st):
                        env.step(acts[t % 64])
        run(0, 500)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(500, steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res.setdefault(k, []).append(n * steps / dt / 1e6)
        for env, _, _ in envs:
            env.close()
for k, v in res.items():
    print(f'{k} sub-shard(s) of {n // k} envs: {sorted(v)[len(v) // 2]:.1f} M env-steps/s (runs: {", ".join(f"{x:.1f}" for x in v)}); {n * 1e-6 / (sorted(v)[len(v) // 2]) * 1e6:.1f} us per step of all {n} envs')
