#!/usr/bin/env python3
"""Per-phase cycle shares of k_step from a DIAGNOSTIC build with in-kernel s_memtime stamps (-DHRL_STAMPS).
The stamped library is built to a separate file and never used by the product; read the SHARES, not the totals
(stamps serialise the phases).  GPU box: python tools/stamp_profile.py [kind] [n_envs]"""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hrl_pybullet_envs_amd import _capi as K  # noqa: E402
from hrl_pybullet_envs_amd import _lib  # noqa: E402

NAMES = ['(block entry)', 'K1 kin+ankle', 'K2 hip', 'S+B base', 'V forward/vel', 'C contacts', 'L limits', 'R1 rows J,B',
         'R2 A,w', 'PGS sweeps', 'point: final u', 'point: integrate', 'load/init', '(substeps->obs)', 'obs: pack (+reset copy)', 'reward+store',
         '(group block end)', 'WAIT for leader', 'recon+clamp', 'WAIT for env blocks', 'I integrate', 'obs: calc_state', 'obs: items', '']


def main():
    kind = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    lib = os.path.join(ROOT, 'gpurun_out', 'libhrl_envs_stamps.so')
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O2', '-std=c++17', '-ffp-contract=off', '-fno-slp-vectorize', '-DHRL_STAMPS',
                           '-fPIC', '-shared', '-o', lib, os.path.join(ROOT, 'hrl_pybullet_envs_amd', 'csrc', 'hrl_hip.hip')])
    _lib.LIB_PATH = lib
    _lib._lib = None
    from hrl_pybullet_envs_amd.vec_env import BatchedEnv
    L = _lib.lib()
    env = BatchedEnv(_lib.default_config(kind, num_envs=n, seed=0, auto_reset=1), 'cuda:0')
    env.reset()
    acts = torch.rand(32, n, env.act_dim, device='cuda') * 2 - 1
    for t in range(60):
        env.step(acts[t % 32])
    stamps = torch.zeros(96, dtype=torch.int64, device='cuda')
    L.hrl_debug_set_stamps(C.c_void_p(stamps.data_ptr()))
    steps = 100
    for t in range(steps):
        env.step(acts[t % 32])
    torch.cuda.synchronize()
    w = stamps.cpu().numpy().astype(float).reshape(4, 24)
    grouped = w[1:].sum() > 0
    v = w.sum(axis=0) / (n * steps)
    tot = v.sum()
    print(f'kind {kind}, {n} envs: {tot:.0f} cycles per env-wave per step (stamped build)')
    if grouped:
        lead, rest = w[0] / (n / 4 * steps), w[1:].sum(axis=0) / (3 * n / 4 * steps)
        print('     phase                      leader wave      other waves   (cycles per step)')
        for i, name in enumerate(NAMES):
            if name:
                print(f'  {i:2d} {name:22s} {lead[i]:12.0f}  {rest[i]:15.0f}')
        print(f'     total                  {lead.sum():12.0f}  {rest.sum():15.0f}')
    else:
        for i, name in enumerate(NAMES):
            if name:
                print(f'  {i:2d} {name:22s} {v[i]:9.0f} cyc  {100 * v[i] / tot:5.1f} %')


if __name__ == '__main__':
    main()
