#!/usr/bin/env python3
"""Interleaved A/B of whole source trees in ONE process (cdna_hip_programming.md rule 24) -- for comparisons across an ABI change, where
tools/variants.py (one ctypes mirror, several .so files) cannot be used: every tree brings its own python package + in-tree library.

    python tools/ab_trees.py [--envs N] [--kind K] [--rounds R] name=path/to/tree ...      ('.' = this checkout)

e.g. on the GPU box:  python tools/ab_trees.py r3=build/r3_tree now=.
A tree is a checkout whose hrl_pybullet_envs_amd/libhrl_envs_hip.so has been built (`git worktree add build/r3_tree <commit>` and
`python -m hrl_pybullet_envs_amd.build` inside it, on the build machine).  Prints median / min microseconds per step launch of each tree and
whether the final states agree bit for bit."""
import argparse
import importlib.util
import os
import sys

import torch

KINDS = {'flat': 0, 'gather': 1, 'maze': 2, 'point': 3, 'maze_mj': 4, 'flagrun': 5}


def load_tree(name, path):
    pkg = os.path.join(os.path.abspath(path), 'hrl_pybullet_envs_amd')
    mod = f'hrl_tree_{name}'
    spec = importlib.util.spec_from_file_location(mod, os.path.join(pkg, '__init__.py'), submodule_search_locations=[pkg])
    m = importlib.util.module_from_spec(spec)
    sys.modules[mod] = m
    spec.loader.exec_module(m)
    return importlib.import_module(mod + '._lib'), importlib.import_module(mod + '._capi'), importlib.import_module(mod + '.vec_env')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=4096)
    ap.add_argument('--kind', default='gather')
    ap.add_argument('--rounds', type=int, default=6)
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--settle', type=int, default=200)
    ap.add_argument('trees', nargs='+')
    a = ap.parse_args()
    kind = KINDS[a.kind]
    envs = {}
    acts = None
    for spec in a.trees:
        name, path = spec.split('=', 1)
        lib, K, ve = load_tree(name, path)
        cfg = lib.default_config(kind, num_envs=a.envs, seed=0, auto_reset=1)
        e = ve.BatchedEnv(cfg, 'cuda:0')
        e.reset()
        if acts is None:
            gen = torch.Generator(device='cuda').manual_seed(0)
            acts = torch.rand(64, a.envs, e.act_dim, device='cuda', generator=gen) * 2 - 1
        for k in range(a.settle):
            e.step(acts[k % 64])
        envs[name] = e
    torch.cuda.synchronize()
    res = {k: [] for k in envs}
    for rnd in range(a.rounds):
        for name, e in envs.items():
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for k in range(20):
                e.step(acts[k % 64])
            e0.record()
            for k in range(a.steps):
                e.step(acts[k % 64])
            e1.record()
            torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / a.steps * 1e3)
    first = next(iter(envs.values()))
    for name, v in res.items():
        v = sorted(v)
        same = bool(torch.equal(envs[name].state, first.state))
        print(f'{a.kind} @{a.envs}  {name:10s} median {v[len(v) // 2]:7.2f} us  min {v[0]:7.2f} us   state == first tree: {same}', flush=True)


if __name__ == '__main__':
    main()
