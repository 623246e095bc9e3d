#!/usr/bin/env python3
"""Life time of every env-wave of one k_step launch, from a DIAGNOSTIC build (-DHRL_WGTIME: two s_memtime reads and one
store per wave, nothing else changed).  Answers: how much of the kernel's duration is the tail of its slowest workgroups,
and whether slow waves cluster on particular CUs / XCDs.  GPU box: python tools/wg_times.py [kind] [n_envs]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hrl_pybullet_envs_amd import _lib  # noqa: E402


def main():
    kind = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    lib = os.path.join(ROOT, 'gpurun_out', 'libhrl_envs_wgtime.so')
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O2', '-std=c++17', '-ffp-contract=off', '-fno-slp-vectorize', '-DHRL_WGTIME',
                           '-fPIC', '-shared', '-o', lib, os.path.join(ROOT, 'hrl_pybullet_envs_amd', 'csrc', 'hrl_hip.hip')])
    _lib.LIB_PATH = lib
    _lib._lib = None
    from hrl_pybullet_envs_amd.vec_env import BatchedEnv
    L = _lib.lib()
    env = BatchedEnv(_lib.default_config(kind, num_envs=n, seed=0, auto_reset=1), 'cuda:0')
    env.reset()
    acts = torch.rand(32, n, env.act_dim, device='cuda') * 2 - 1
    for t in range(200):
        env.step(acts[t % 32])
    buf = torch.zeros(4 * n, dtype=torch.int64, device='cuda')
    L.hrl_debug_set_stamps(C.c_void_p(buf.data_ptr()))
    rows, us = [], []
    for t in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.step(acts[t % 32]); e1.record()
        torch.cuda.synchronize()
        us.append(e0.elapsed_time(e1) * 1e3)
        rows.append(buf.cpu().numpy().reshape(n, 4).copy())
    print(f'launch + step wrapper, HIP events: median {np.median(us):.1f} us (this diagnostic build)')
    v = np.concatenate(rows)
    launch = np.repeat(np.arange(20), n)
    t0, t1, hw, xcc = v[:, 0], v[:, 1], v[:, 2], v[:, 3] & 15
    life = (t1 - t0).astype(float)
    # s_memtime counters are not synchronised across CUs: spans are taken inside one (launch, CU)
    cu_all = ((hw >> 8) & 15) | (((hw >> 13) & 7) << 4) | (xcc << 8)
    key = launch * 4096 + cu_all
    order = np.argsort(key, kind='stable')
    ks, st0, st1, sl = key[order], t0[order], t1[order], life[order]
    bounds = np.flatnonzero(np.diff(ks)) + 1
    starts = np.concatenate([[0], bounds]); stops = np.concatenate([bounds, [len(ks)]])
    span_cu = np.array([st1[a:b].max() - st0[a:b].min() for a, b in zip(starts, stops)], dtype=float)
    ramp_cu = np.array([st0[a:b].max() - st0[a:b].min() for a, b in zip(starts, stops)], dtype=float)
    mean_cu = np.array([sl[a:b].mean() for a, b in zip(starts, stops)])
    print(f'kind {kind}, {n} envs, 20 launches, s_memtime ticks')
    print(f'  mean wave life {life.mean():9.0f}; min {life.min():.0f}, p10 {np.percentile(life, 10):.0f}, '
          f'median {np.median(life):.0f}, p90 {np.percentile(life, 90):.0f}, max {life.max():.0f}')
    print(f'  per (launch, CU): first wave start -> last wave end: mean {span_cu.mean():.0f}, p50 {np.median(span_cu):.0f}, p90 {np.percentile(span_cu, 90):.0f}, '
          f'p99 {np.percentile(span_cu, 99):.0f}, max {span_cu.max():.0f}; mean wave life in the CU / that span: {np.mean(mean_cu / span_cu):.3f}; '
          f'last wave start - first wave start: mean {ramp_cu.mean():.0f}')
    per_launch_max = np.array([span_cu[(ks[starts] // 4096) == la].max() for la in range(20)])
    print(f'  slowest CU of a launch (= the launch, but for the dispatch ramp across CUs): mean {per_launch_max.mean():.0f} = {per_launch_max.mean() / span_cu.mean():.3f} x the mean CU, '
          f'{per_launch_max.mean() / life.mean():.3f} x the mean wave life')
    # HW_ID (gfx9): WAVE_ID [3:0], SIMD_ID [5:4], CU_ID [11:8], SE_ID [15:13], TG_ID [19:16]
    for name, key in (('WAVE_ID (slot on its SIMD)', hw & 15), ('SIMD_ID', (hw >> 4) & 3), ('TG_ID (workgroup slot on its CU)', (hw >> 16) & 15),
                      ('wave index in its workgroup', np.tile(np.arange(n) & 3, 20))):
        print(f'  mean life by {name}: ' + ', '.join(f'{int(k)}: {life[key == k].mean():.0f} (n={int((key == k).sum())})' for k in np.unique(key)))
    dbg = v[:, 3] >> 8
    rows, cubes, selfs = (dbg & 0xffff).astype(float), ((dbg >> 16) & 0xff).astype(float), ((dbg >> 24) & 0xff).astype(float)
    if rows.max() > 0:
        # a workgroup lives as long as its slowest env: group quantities = max over the four envs
        g = lambda a: a.reshape(-1, 4).max(axis=1)
        gl, gr, gc, gs = g(life), g(rows), g(cubes), g(selfs)
        print(f'  per step and env: solver rows summed over the substeps mean {rows.mean():.1f} (p10 {np.percentile(rows, 10):.0f}, p90 {np.percentile(rows, 90):.0f}, '
              f'max {rows.max():.0f}); cube passes {cubes.mean():.2f}; substeps with self contacts {selfs.mean():.3f}')
        print(f'  correlation of a workgroup\'s life with (max over its envs of) rows {np.corrcoef(gl, gr)[0, 1]:.3f}, cube passes {np.corrcoef(gl, gc)[0, 1]:.3f}, '
              f'self substeps {np.corrcoef(gl, gs)[0, 1]:.3f}')
        A = np.stack([np.ones_like(gr), gr, gc, gs], axis=1)
        coef, *_ = np.linalg.lstsq(A, gl, rcond=None)
        res = gl - A @ coef
        print(f'  least squares: life = {coef[0]:.0f} + {coef[1]:.1f} * rows + {coef[2]:.0f} * cube passes + {coef[3]:.0f} * self substeps; residual sd {res.std():.0f} of sd {gl.std():.0f}')
    widx, simd = np.tile(np.arange(n) & 3, 20), (hw >> 4) & 3
    print('  waves by (index in workgroup, SIMD_ID): ' + ' | '.join(' '.join(str(int(((widx == i) & (simd == j)).sum())) for j in range(4)) for i in range(4)))
    if rows.max() > 0:
        R = rows.reshape(20, n)
        cc = lambda k: float(np.mean([np.corrcoef(R[t], R[t + k])[0, 1] for t in range(20 - k)]))
        print(f'  persistence of an env\'s row count: correlation with the next step {cc(1):.3f}, 4 steps later {cc(4):.3f}, 8 later {cc(8):.3f}, 16 later {cc(16):.3f}')
        Lf = life.reshape(20, n)
        print(f'  persistence of a wave\'s life: next step {float(np.mean([np.corrcoef(Lf[t], Lf[t + 1])[0, 1] for t in range(19)])):.3f}')
    # which workgroups share a CU?  (last launch) -- the dispatcher's block -> CU map
    last = slice((20 - 1) * n, 20 * n, 4)
    cul = cu_all[last]
    first = {}
    for w, c_ in enumerate(cul):
        first.setdefault(int(c_), []).append(w)
    some = list(first.items())[:6]
    print('  workgroups of the first CUs seen (block indices): ' + '; '.join(f'{k:#x}: {v}' for k, v in some))
    strides = sorted({tuple(np.diff(v)) for v in first.values()})
    print(f'  distinct index-difference patterns of the workgroups sharing a CU: {strides[:8]}{" ..." if len(strides) > 8 else ""} ({len(strides)} patterns)')
    cu = ((hw >> 8) & 15) | (((hw >> 13) & 7) << 4) | (xcc << 8)
    per = np.array([life[cu == c_].mean() for c_ in np.unique(cu)])
    print(f'  {len(per)} CUs; mean life per CU: min {per.min():.0f}, median {np.median(per):.0f}, max {per.max():.0f}')


if __name__ == '__main__':
    main()
