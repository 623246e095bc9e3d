#!/usr/bin/env python3
"""A/B compiler-flag variants of the HIP library in ONE process, interleaved rounds (cdna_hip_programming.md rule 24).
Variants are built beforehand into build/variants/lib_<name>.so.  GPU box: python tools/variants.py [envs] [kind] [settle steps] [seed]
(HRL_VARIANT_MODEL=field=value,...: hrl_model overrides for every variant)"""
import ctypes as C
import glob
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hrl_pybullet_envs_amd import _capi as K  # noqa: E402
from hrl_pybullet_envs_amd import _lib  # noqa: E402


def bind(path):
    L = C.CDLL(path)
    L.hrl_last_error.restype = C.c_char_p
    L.hrl_create.argtypes = [C.POINTER(K.hrl_config), C.POINTER(C.c_void_p)]
    L.hrl_reset.argtypes = [C.c_void_p, C.POINTER(K.hrl_buffers), C.c_void_p, C.c_void_p]
    L.hrl_step.argtypes = [C.c_void_p, C.POINTER(K.hrl_buffers), C.c_void_p]
    return L


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    kind = int(sys.argv[2]) if len(sys.argv) > 2 else K.HRL_ANT_GATHER
    settle = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # untimed steps of every variant before the first timed round (the settled regime: 500)
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    libs = {'product': bind(_lib.LIB_PATH)}
    for p in sorted(glob.glob(os.path.join(ROOT, 'build', 'variants', 'lib_*.so'))):
        libs[os.path.basename(p)[4:-3]] = bind(p)
    cfg = _lib.default_config(kind, num_envs=n, seed=seed, auto_reset=1)
    for kv in filter(None, os.environ.get('HRL_VARIANT_MODEL', '').split(',')):   # e.g. HRL_VARIANT_MODEL=linear_damping=0,angular_damping=0 (the undamped population)
        k, v = kv.split('=')
        setattr(cfg.model, k, type(getattr(cfg.model, k))(float(v)))
    obs_dim, act_dim = _lib.lib().hrl_obs_dim(C.byref(cfg)), _lib.lib().hrl_act_dim(C.byref(cfg))
    envs = {}
    acts = torch.rand(64, n, act_dim, device='cuda') * 2 - 1
    libs['product+final'] = libs['product']  # the same library with the optional final_obs / truncated outputs of ABI v6 wired up
    for name, L in libs.items():
        h = C.c_void_p()
        assert L.hrl_create(C.byref(cfg), C.byref(h)) == 0
        t = dict(state=torch.zeros(n, 32, device='cuda'), items=torch.zeros(n, 32, device='cuda'), aux=torch.zeros(n, 4, dtype=torch.int32, device='cuda'),
                 obs=torch.zeros(n, obs_dim + 2, device='cuda'), rew=torch.zeros(n, device='cuda'), done=torch.zeros(n, dtype=torch.uint8, device='cuda'),
                 info=torch.zeros(n, 4, device='cuda'))
        b = K.make_buffers(t['state'].data_ptr(), t['items'].data_ptr(), t['aux'].data_ptr(), None, t['obs'].data_ptr(), t['rew'].data_ptr(),
                          t['done'].data_ptr(), t['info'].data_ptr())
        if name.endswith('+final'):
            t['final'] = torch.zeros(n, obs_dim, device='cuda'); t['trunc'] = torch.zeros(n, dtype=torch.uint8, device='cuda')
            b.final_obs = t['final'].data_ptr(); b.truncated = t['trunc'].data_ptr()
        L.hrl_reset(h, C.byref(b), None, None)
        for k in range(settle):
            b.actions = acts[k % 64].data_ptr(); L.hrl_step(h, C.byref(b), None)
        envs[name] = (L, h, b, t)
    res = {k: [] for k in envs}
    for rnd in range(6):
        for name, (L, h, b, t) in envs.items():
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            for k in range(20):
                b.actions = acts[k % 64].data_ptr(); L.hrl_step(h, C.byref(b), st)
            e0.record()
            for k in range(200):
                b.actions = acts[k % 64].data_ptr(); L.hrl_step(h, C.byref(b), st)
            e1.record(); torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / 200 * 1e3)
    ref = envs['product'][3]['state']
    for name, v in res.items():
        v = sorted(v)
        same = bool(torch.equal(envs[name][3]['state'], ref))
        print(f'{name:12s} median {v[len(v) // 2]:7.1f} us  min {v[0]:7.1f} us   state==product: {same}', flush=True)


if __name__ == '__main__':
    main()
