#!/usr/bin/env python3
"""Two populations of 4096 AntGather envs side by side, the bodies' damping off and on (hrl_model.linear_damping / angular_damping), in the settled regime:
launch time, solver rows of every env, the mean over the groups' slowest envs.  What it showed: a launch in which ONE env holds 88 - 92 rows per step instead of a
standing ant's 80 lasts 3 - 4 us longer, damped or not (profiles/EXPERIMENTS.md 8.w).  GPU box: python tools/damping_probe.py"""
import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd())
from hrl_pybullet_envs_amd import _capi as K, _lib
from hrl_pybullet_envs_amd.vec_env import BatchedEnv
n = 4096
acts = torch.rand(256, n, 8, device='cuda') * 2 - 1
envs = {}
for name, d in (('damping 0', 0.0), ('damping 0.04', 0.04)):
    cfg = _lib.default_config(K.HRL_ANT_GATHER, num_envs=n, seed=0, auto_reset=1)
    cfg.model.linear_damping = d; cfg.model.angular_damping = d
    e = BatchedEnv(cfg, 'cuda:0'); e.reset(); envs[name] = e
    for t in range(500):
        e.step(acts[t % 256])
for rnd in range(3):
    for name, e in envs.items():
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(300):
            e.step(acts[t % 256])
        e1.record(); torch.cuda.synchronize()
        rows = e.count_solver_rows()
        e.step(acts[0]); torch.cuda.synchronize()
        r = rows.cpu().numpy().astype(float)
        g = r.reshape(-1, 4).max(axis=1)
        z = e.state[:, 2].cpu().numpy()
        print(f'{name:13s} {e0.elapsed_time(e1) / 300 * 1e3:6.1f} us/step   rows per env-step: mean {r.mean():5.1f} p99 {np.percentile(r, 99):5.0f} max {r.max():4.0f}; mean of group max {g.mean():5.1f}; torso z mean {z.mean():.3f}, share below 0.3: {(z < 0.3).mean():.3f}')
        e.count_solver_rows(False)
