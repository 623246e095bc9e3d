#!/usr/bin/env python3
"""Print the distribution of single-step differences between the HIP kernels and the fp32 CPU oracle
(identical state/items/action each step).  Run on a GPU box: python tests/tools/gpu_parity_report.py [kind] [N] [T]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import orc  # noqa: E402
from hrl_pybullet_envs_amd import _lib  # noqa: E402
from hrl_pybullet_envs_amd.vec_env import BatchedEnv  # noqa: E402


def main():
    kind = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    cfg = _lib.default_config(kind, num_envs=n, seed=3, auto_reset=1)
    g = BatchedEnv(cfg, 'cuda:0'); o = orc.OracleEnv(orc.default_config(kind, num_envs=n, seed=3, auto_reset=1), np.float32)
    g.reset(); o.reset()
    rng = np.random.RandomState(0)
    es, eo, flips = [], [], 0
    for t in range(T):
        a = rng.uniform(-1, 1, (n, o.ad)).astype(np.float32)
        g.state.copy_(torch.from_numpy(o.state)); g.items.copy_(torch.from_numpy(o.items)); g.aux.copy_(torch.from_numpy(o.aux))
        go, gr, gd, gi = g.step(torch.from_numpy(a).cuda()); o.step(a)
        torch.cuda.synchronize()
        flip = (gd.cpu().numpy() != o.done) | (gr.cpu().numpy() != o.rew)
        flips += int(flip.sum())
        es.append(np.abs(g.state.cpu().numpy() - o.state)[~flip].max(axis=1)); eo.append(np.abs(go.cpu().numpy() - o.obs)[~flip].max(axis=1))
    es, eo = np.concatenate(es), np.concatenate(eo)
    pr = [50, 90, 99, 99.9, 100]
    print(f'kind {kind} N {n} T {T}: flips {flips}')
    print('state |d| percentiles', pr, np.percentile(es, pr))
    print('obs   |d| percentiles', pr, np.percentile(eo, pr))
    print('bit-exact state fraction', float((es == 0).mean()))


if __name__ == '__main__':
    main()
