#!/usr/bin/env python3
"""Cost of the optional collision passes at run time (no rebuild): the same kernel with item_collision / self_collision
switched off in the config, and of the optional model parameters switched on.  GPU box: python tools/cfg_ablate.py [n_envs]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hrl_pybullet_envs_amd import _capi as K  # noqa: E402
from hrl_pybullet_envs_amd import _lib  # noqa: E402
from hrl_pybullet_envs_amd.vec_env import BatchedEnv  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    acts = torch.rand(64, n, 8, device='cuda') * 2 - 1
    envs = {}
    for name, kw in (('default', {}), ('no cubes', dict(item_collision=0)), ('no self', dict(self_collision=0)),
                     ('neither', dict(item_collision=0, self_collision=0)),
                     ('no damping', dict(linear_damping=0.0, angular_damping=0.0)),                # (the default has Bullet's per-body damping: 0.04 / 0.04)
                     ('joint damping 1 + armature 1', dict(joint_damping=1.0, joint_armature=1.0)),
                     ('restitution 0.5', dict(restitution=0.5))):
        cfg = _lib.default_config(K.HRL_ANT_GATHER, num_envs=n, seed=0, auto_reset=1)
        for k, v in kw.items():
            setattr(cfg.model, k, v)
        e = BatchedEnv(cfg, 'cuda:0')
        e.reset()
        envs[name] = e
    res = {k: [] for k in envs}
    for rnd in range(6):
        for name, e in envs.items():
            for k in range(20):
                e.step(acts[k % 64])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for k in range(200):
                e.step(acts[k % 64])
            e1.record(); torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / 200 * 1e3)
    for name, v in res.items():
        v = sorted(v)
        print(f'{n} envs  {name:40s} median {v[len(v) // 2]:7.1f} us  min {v[0]:7.1f} us')


if __name__ == '__main__':
    main()
