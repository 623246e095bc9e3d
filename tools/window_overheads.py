#!/usr/bin/env python3
"""Where the wall clock of a SHORT timed window goes (the driver's `--steps 20`): host time to queue the launches, the GPU's span from the first kernel's
start to the last one's end (HIP events), and what is left -- the dispatch latency of the first launch after an idle queue and the host's wake-up from the
synchronisation.  GPU box: python tools/window_overheads.py [steps] [envs]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hrl_pybullet_envs_amd import _capi as K, _lib  # noqa: E402
from hrl_pybullet_envs_amd.vec_env import BatchedEnv  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    env = BatchedEnv(_lib.default_config(K.HRL_ANT_GATHER, num_envs=n, seed=0, auto_reset=1), 'cuda:0')
    env.reset()
    acts = torch.rand(256, n, 8, device='cuda') * 2 - 1
    for t in range(305):
        env.step(acts[t % 256])
    torch.cuda.synchronize()
    rows = []
    for rep in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        env.step(acts[0])
        e0.record()                      # behind the first launch: stamped when that kernel ends
        t1 = time.perf_counter()
        for t in range(1, steps):
            env.step(acts[t % 256])
        e1.record()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        gpu = e0.elapsed_time(e1) * 1e3  # launches 2..K
        rows.append(((t3 - t0) * 1e6, (t1 - t0) * 1e6, (t2 - t0) * 1e6, gpu))
    rows = rows[2:]
    import statistics as st
    wall, first, queued, gpu = (st.median(r[i] for r in rows) for i in range(4))
    per = gpu / (steps - 1)
    print(f'{steps} launches of {n} envs: wall {wall:7.1f} us = {wall / steps:5.2f} us per step; kernels 2..K on the GPU {gpu:7.1f} us ({per:5.2f} us each)')
    print(f'  host: first launch call returned after {first:5.1f} us, all {steps} queued after {queued:6.1f} us ({queued / steps:4.1f} us per launch call)')
    print(f'  wall - K x kernel = {wall - steps * per:5.1f} us: dispatch latency of the first launch on an idle queue + the host waking up from the synchronisation')


if __name__ == '__main__':
    main()
