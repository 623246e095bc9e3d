#!/usr/bin/env python3
"""How often is a whole PGS sweep a fixed point?  (if every (c, lambda) bit is unchanged after a sweep, the remaining sweeps are
identical and skipping them is exact.)  Builds an instrumented copy of the CPU oracle (-DORC_SWEEP_STATS), runs a settled rollout of every ant kind
and the point bot with uniform random actions, prints per sweep: run and changed something | run and found fixed | skippable.  CPU only."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from hrl_pybullet_envs_amd import _capi as K  # noqa: E402

so = '/tmp/liborc_sweeps.so'
subprocess.check_call(['gcc', '-O2', '-fPIC', '-fopenmp', '-ffp-contract=off', '-mfma', '-std=c11', '-DORC_SWEEP_STATS', '-shared', '-o', so,
                       os.path.join(ROOT, 'oracle', 'hrl_oracle.c'), '-lm'])
L = C.CDLL(so)
names = {0: 'flat', 1: 'gather', 2: 'maze', 3: 'point', 4: 'maze_mj', 5: 'flagrun'}
for kind in (1, 0, 2, 3):
    cfg = K.hrl_config(); L.orc_default_config(C.c_int32(kind), C.byref(cfg))
    n = 1024; cfg.num_envs = n; cfg.auto_reset = 1; cfg.seed = 0
    od, ad, st = L.orc_obs_dim(C.byref(cfg)), L.orc_act_dim(C.byref(cfg)), L.orc_items_stride(C.byref(cfg))
    f = np.float32
    state, items, aux = np.zeros((n, 32), f), np.zeros((n, st), f), np.zeros((n, 4), np.int32)
    obs, rew, done, info = np.zeros((n, od), f), np.zeros(n, f), np.zeros(n, np.uint8), np.zeros((n, 4), f)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L.orc_reset_batch_f32(C.byref(cfg), p(state), p(items), p(aux), None, p(obs))
    rng = np.random.RandomState(0)
    stats = (C.c_longlong * (64 * 3)).in_dll(L, 'orc_sweep_stats'); rows = (C.c_longlong * (64 * 3)).in_dll(L, 'orc_sweep_rows')
    for t in range(400):
        if t == 200:
            for i in range(64 * 3):
                stats[i] = 0; rows[i] = 0
        a = rng.uniform(-1, 1, (n, ad)).astype(f)
        L.orc_step_batch_f32(C.byref(cfg), p(state), p(items), p(aux), p(a), p(obs), p(rew), p(done), p(info))
    s = np.array(list(stats)).reshape(64, 3)[:5]; r = np.array(list(rows)).reshape(64, 3)[:5]
    tot = s[0].sum()
    print(f'{names[kind]:8s} {tot} solves (substeps with rows) in 200 steps x {n} envs; per sweep: changed | fixed | skippable   (rows in the same classes)')
    for it in range(5):
        print(f'   sweep {it}: {s[it, 0]:9d} {s[it, 1]:9d} {s[it, 2]:9d}    rows {r[it, 0]:10d} {r[it, 1]:10d} {r[it, 2]:10d}')
    print(f'   row-sweeps skippable: {r[:, 2].sum() / max(1, r.sum()):.3%} of {r.sum()}')
