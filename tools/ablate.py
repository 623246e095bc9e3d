#!/usr/bin/env python3
"""Time k_step under config ablations (solver iterations, substeps, contact distance) to see where the time goes.
GPU box: python tools/ablate.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hrl_pybullet_envs_amd import _capi as K  # noqa: E402
from hrl_pybullet_envs_amd import _lib  # noqa: E402
from hrl_pybullet_envs_amd.vec_env import BatchedEnv  # noqa: E402


def timeit(kind, n=4096, steps=300, **kw):
    cfg = _lib.default_config(kind, num_envs=n, seed=0, auto_reset=1, **kw)
    env = BatchedEnv(cfg, 'cuda:0')
    env.reset()
    g = torch.Generator(device='cuda').manual_seed(0)
    acts = torch.rand(64, n, env.act_dim, device='cuda', generator=g) * 2 - 1
    for t in range(50):
        env.step(acts[t % 64])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(steps):
        env.step(acts[t % 64])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


if __name__ == '__main__':
    for name, kind, kw in [
        ('gather default', 1, {}),
        ('gather iters=1', 1, dict(model_solver_iters=1)),
        ('gather iters=10', 1, dict(model_solver_iters=10)),
        ('gather substeps=1', 1, dict(model_frame_skip=1)),
        ('gather substeps=8', 1, dict(model_frame_skip=8)),
        ('gather no contacts/limits rows', 1, dict(model_contact_dist=-1e9, model_limit_margin=-1e9)),
        ('point default', 3, {}),
        ('point iters=1', 3, dict(model_solver_iters=1)),
        ('point substeps=1', 3, dict(model_frame_skip=1)),
        ('point no contacts', 3, dict(model_contact_dist=-1e9)),
        ('gather n=1024', 1, dict()),
    ]:
        n = 1024 if 'n=1024' in name else 4096
        print(f'{name:34s} {timeit(kind, n=n, **kw):8.1f} us/step', flush=True)
