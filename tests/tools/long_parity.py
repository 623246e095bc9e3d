#!/usr/bin/env python3
"""GPU box: HIP kernels and the fp32 CPU oracle run FREE (no state copying) side by side for T steps with random actions and
auto-reset; state / items / counters / reward / done are compared bit for bit at every step, and so are the observations.
    python tests/tools/long_parity.py [T] [N] [kind ...]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import orc  # noqa: E402
from hrl_pybullet_envs_amd import _lib  # noqa: E402
from hrl_pybullet_envs_amd.vec_env import BatchedEnv  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 2100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
kinds = [int(k) for k in sys.argv[3:] if k.isdigit()] or [0, 1, 2, 3, 4, 5]
KW = dict(big=dict(n_food=20, n_poison=12, n_bins=24), huge=dict(n_food=40, n_poison=24, n_bins=64), short=dict(max_episode_steps=50),
          contact=dict(robot_coll_dist=0.0, use_sensor=0, n_food=32, n_poison=32, world_size=(6.0, 6.0), robot_object_spacing=0.5))
kw = {}
for k in sys.argv[3:]:  # named config sets after the kinds: `big` (32 items, 24 bins), `huge` (64 items, 64 bins), `short` (50-step limit), `contact` (pickup by contact, positions in the observation, a crowded 6 x 6 arena)
    kw.update(KW.get(k, {}))
names = ['flat', 'gather', 'maze', 'point', 'maze_mj', 'flagrun']
for kind in kinds:
    g = BatchedEnv(_lib.default_config(kind, num_envs=n, seed=5, auto_reset=1, **kw), 'cuda:0')
    o = orc.OracleEnv(orc.default_config(kind, num_envs=n, seed=5, auto_reset=1, **kw), np.float32)
    g.reset(); o.reset()
    rng = np.random.RandomState(kind)
    t0 = time.time()
    first_bad, obs_max, obs_rows, episodes = None, 0.0, 0, 0
    for t in range(T):
        a = rng.uniform(-1, 1, (n, o.ad)).astype(np.float32)
        go, gr, gd, _ = g.step(torch.from_numpy(a).cuda()); o.step(a)
        same = (np.array_equal(g.state.cpu().numpy(), o.state) and np.array_equal(g.items.cpu().numpy(), o.items)
                and np.array_equal(g.aux.cpu().numpy(), o.aux) and np.array_equal(gr.cpu().numpy(), o.rew, equal_nan=True)
                and np.array_equal(gd.cpu().numpy(), o.done) and np.array_equal(g.truncated.cpu().numpy(), o.truncated)
                and (not o.done.any() or np.array_equal(g.final_obs.cpu().numpy(), o.final_obs, equal_nan=True)))  # terminal observations: when some episode ended
        if not same and first_bad is None:
            first_bad = t
            break
        d = np.abs(go.cpu().numpy() - o.obs)
        d = np.where(np.isfinite(d), d, 0.0)
        obs_max = max(obs_max, float(d.max())); obs_rows += int((d.max(axis=1) > 0).sum()); episodes += int(o.done.sum())
    print(f'{names[kind]:8s} {kw if kw else ""} N {n} T {T}: state/items/aux/reward/done/truncated/final_obs bit-exact for {T if first_bad is None else first_bad} steps'
          f'{"" if first_bad is None else " (FIRST MISMATCH at step %d)" % first_bad}; obs max |d| {obs_max:.2e}, obs rows that differ: {obs_rows} '
          f'of {n * T}; episodes finished {episodes}; {time.time() - t0:.0f} s', flush=True)
