#!/usr/bin/env python3
"""The PCIe-inclusive rate: env-steps/s when the caller hands over HOST buffers (numpy actions in, numpy observations / rewards / dones out, the
shape of the reference's `env.step(a)`), beside the device-resident rate bench.py reports.  GPU box:  python tools/host_io_rate.py [kind] [envs ...]

  resident   actions and outputs stay in HBM (bench.py's `value`)
  zero-copy  BatchedEnv.step_host(): the kernel reads the actions from and writes its outputs to PINNED host memory itself; one launch + one
             synchronisation per step
  staged     pinned host action -> async H2D copy -> step -> async D2H copies of obs / reward / done / info into pinned host memory -> one
             synchronisation (what a binding with its own staging buffers does)
  sb3        adapters.SB3VecEnv.step(numpy): pageable numpy in / numpy out and the per-env info dicts SB3 asks for (host-side Python included)
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hrl_pybullet_envs_amd as envs  # noqa: E402
from hrl_pybullet_envs_amd.adapters import SB3VecEnv  # noqa: E402

IDS = {'gather': 'AntGatherBulletEnv-v0', 'point': 'PointGatherBulletEnv-v0', 'maze': 'AntMazeBulletEnv-v0', 'flat': 'AntMjEnv-v0',
       'maze_mj': 'AntMazeMjEnv-v0', 'flagrun': 'AntFlagrunBulletEnv-v0'}


def rate(fn, n, steps):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return n * steps / dt, dt / steps * 1e6


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else 'gather'
    sizes = [int(a) for a in sys.argv[2:]] or [256, 4096, 32768]
    print(f'{IDS[kind]}: env-steps/s (us per step) with device-resident and with host buffers')
    for n in sizes:
        env = envs.make(IDS[kind], num_envs=n, seed=0)
        env.reset()
        be = env._backend()          # the BatchedEnv under the reference-shaped class
        ad, od = be.act_dim, be.obs_dim
        for _ in range(300):   # settled, as bench.py
            env.step(torch.rand(n, ad, device='cuda') * 2 - 1)
        steps = 300 if n <= 4096 else 60
        a_dev = torch.rand(n, ad, device='cuda') * 2 - 1
        a_np = a_dev.cpu().numpy()
        a_pin = a_dev.cpu().pin_memory()
        res = {}
        res['resident'] = rate(lambda: env.step(a_dev), n, steps)
        res['zero-copy'] = rate(lambda: be.step_host(a_np), n, steps)
        out = {k: torch.empty_like(getattr(be, k), device='cpu').pin_memory() for k in ('obs', 'reward', 'done', 'info')}
        a_stage = torch.empty_like(a_dev)

        def staged():
            a_stage.copy_(a_pin, non_blocking=True)
            be.step(a_stage)
            for k, v in out.items():
                v.copy_(getattr(be, k), non_blocking=True)
            torch.cuda.current_stream().synchronize()
        res['staged'] = rate(staged, n, steps)
        sb = SB3VecEnv(env)
        res['sb3'] = rate(lambda: sb.step(a_np), n, max(steps // 3, 10))
        nbytes = n * 4 * (ad + od + 1 + 4) + n
        print(f'  N {n:6d} ({nbytes / 1e3:7.1f} kB over PCIe per step): ' + '; '.join(f'{k} {v[0] / 1e6:6.2f} M ({v[1]:7.1f} us)' for k, v in res.items()))
        env.close()


if __name__ == '__main__':
    main()
