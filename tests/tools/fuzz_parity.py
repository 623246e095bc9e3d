#!/usr/bin/env python3
"""Parity under absurd inputs: the wave phases (the HIP kernels on a GPU box, `gpu`; the lock-step host executor of tests/emu, `emu`) against
the fp32 oracle while NaN, +-inf, 1e20, 3e38, denormals and signed zeros are written into positions, velocities, item coordinates and actions of
running envs -- the values a simulation that blows up can reach (quaternions are unit or NaN, joint angles moderate, joint rates within their clamp).
Every output is compared bit for bit (NaNs as equal) at every step.    python tests/tools/fuzz_parity.py [seeds] [gpu|emu] [matrix]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import orc  # noqa: E402
from hrl_pybullet_envs_amd import _capi as K  # noqa: E402

KINDS = [K.HRL_ANT_FLAT, K.HRL_ANT_GATHER, K.HRL_ANT_MAZE, K.HRL_POINT_GATHER, K.HRL_ANT_MAZE_MJ, K.HRL_ANT_FLAGRUN]
VALS = [np.nan, np.inf, -np.inf, 1e20, -1e20, 3e38, -3e38, 1e-40, -1e-40, -0.0, 0.0, 1e-20, 1e10, 5.0, -5.0, 100.0]


class GpuSide:
    def __init__(self, kind, n, seed, auto_reset, kw):
        import torch
        from hrl_pybullet_envs_amd import _lib
        from hrl_pybullet_envs_amd.vec_env import BatchedEnv
        self.t, self.g = torch, BatchedEnv(_lib.default_config(kind, num_envs=n, seed=seed, auto_reset=auto_reset, **kw), 'cuda:0')

    def reset(self): self.g.reset()

    def step(self, o, a):  # the oracle's (edited) pre-step buffers are pushed, then both step
        t, g = self.t, self.g
        g.state.copy_(t.from_numpy(o.state)); g.items.copy_(t.from_numpy(o.items)); g.aux.copy_(t.from_numpy(o.aux))
        g.step(t.from_numpy(a).cuda())

    def outputs(self):
        g = self.g
        return {k: v.cpu().numpy() for k, v in dict(state=g.state, items=g.items, aux=g.aux, obs=g.obs, rew=g.reward, done=g.done, info=g.info,
                                                     final_obs=g.final_obs, truncated=g.truncated).items()}


class EmuSide:
    def __init__(self, kind, n, seed, auto_reset, kw):
        import emu_env
        self.e = emu_env.EmuEnv(orc.default_config(kind, num_envs=n, seed=seed, auto_reset=auto_reset, **kw))

    def reset(self): self.e.reset()

    def step(self, o, a):
        e = self.e
        e.state[...] = o.state; e.items[...] = o.items; e.aux[...] = o.aux
        e.step(a)

    def outputs(self):
        e = self.e
        return dict(state=e.state, items=e.items, aux=e.aux, obs=e.obs, rew=e.rew, done=e.done, info=e.info, final_obs=e.final_obs, truncated=e.truncated)


def run(Side, kind, seed, auto_reset, n=48, T=10, kw=None):
    kw = kw or {}
    o = orc.OracleEnv(orc.default_config(kind, num_envs=n, seed=seed, auto_reset=auto_reset, **kw), np.float32)
    s = Side(kind, n, seed, auto_reset, kw)
    o.reset(); s.reset()
    rng = np.random.RandomState(seed)
    nq, nv = (7, 6) if kind == K.HRL_POINT_GATHER else (15, 14)
    for t in range(T):
        a = rng.uniform(-1, 1, (n, o.ad)).astype(np.float32)
        if t % 2 == 0:
            for r in rng.permutation(n)[:32]:
                what, v = rng.randint(0, 4), np.float32(VALS[rng.randint(len(VALS))])
                if what == 0:  # a position, a quaternion component, a joint angle
                    idx = rng.randint(0, nq)
                    if 3 <= idx < 7: v = np.float32(np.nan)  # a quaternion is unit or NaN
                    if idx >= 7: v = np.float32(rng.choice([np.nan, 2000.0, -1500.0, 3.0, -0.0]))  # joint rates are clamped: angles stay moderate
                    o.state[r, idx] = v
                elif what == 1:  # a velocity
                    idx = 15 + rng.randint(0, nv)
                    if idx >= 21 and np.isfinite(v) and abs(v) > 100: v = np.float32(100.0 * np.sign(v))  # joint rates leave a step clamped
                    o.state[r, idx] = v
                elif what == 2 and kind in (K.HRL_ANT_GATHER, K.HRL_POINT_GATHER): o.items[r, rng.randint(0, o.items.shape[1])] = v
                else: a[r, rng.randint(0, o.ad)] = v
        pre = o.state.copy()
        s.step(o, a); o.step(a)
        out = s.outputs()
        for name in ('state', 'items', 'aux', 'obs', 'rew', 'done', 'info', 'final_obs', 'truncated'):
            A, B = getattr(o, name).reshape(n, -1), out[name].reshape(n, -1)
            ok = (A == B) | ((A != A) & (B != B))
            if not ok.all():
                e = int(np.where(~ok.all(1))[0][0])
                return f'kind {kind} {kw} seed {seed} auto_reset {auto_reset} step {t}: {name} differs for env {e} at {np.where(~ok[e])[0][:8]}; oracle {A[e][~ok[e]][:6]} other {B[e][~ok[e]][:6]}; pre-step state {pre[e, :29]}'
    return None


MATRIX = [  # non-default branches (the CONFIG_MATRIX of the parity tests), fuzzed with `matrix` as third argument
    (K.HRL_ANT_GATHER, dict(n_bins=7, n_food=5, n_poison=3, sensor_range=9.0, sensor_span=2.0, world_size=(9.0, 11.0))),
    (K.HRL_ANT_GATHER, dict(respawn=0, robot_coll_dist=4.0, dying_cost=-3.0)),
    (K.HRL_ANT_GATHER, dict(robot_coll_dist=0.0)),
    (K.HRL_ANT_GATHER, dict(use_sensor=0)),
    (K.HRL_POINT_GATHER, dict(robot_coll_dist=-1.0, respawn=0)),
    (K.HRL_POINT_GATHER, dict(use_sensor=0)),
    (K.HRL_ANT_MAZE, dict(sense_target=1, n_bins=8)),
    (K.HRL_ANT_MAZE, dict(target_encoding=1, sense_walls=0, tol=3.0, targ_dist_rew=1, max_steps=20, done_at_target=0)),
    (K.HRL_ANT_MAZE_MJ, dict(inner_rew_weight=0.5, n_bins=6)),
    (K.HRL_ANT_FLAGRUN, dict(flag_max_targets=0, flag_max_target_dist=2.5, flag_timeout=6, flag_size=3.0, world_size=(5.0, 5.0), centroid_static_sum=(-2.5, 0.0))),
    (K.HRL_ANT_FLAGRUN, dict(flag_enclosed=0, centroid_n_static=1, centroid_static_sum=(0.0, 0.0), flag_timeout=8, flag_max_targets=5)),
    (K.HRL_ANT_FLAGRUN, dict(flag_switch_on_collision=0, flag_timeout=7, flag_max_targets=4)),
    # beyond the capacity caps of ABI <= 5 (more than 16 / 48 items, observations wider than the wave, many targets) and with a step limit short
    # enough that truncations, terminal observations and in-kernel resets happen inside the 10 fuzzed steps
    (K.HRL_ANT_GATHER, dict(n_food=20, n_poison=12, n_bins=24, max_episode_steps=4)),
    (K.HRL_POINT_GATHER, dict(n_food=40, n_poison=24, n_bins=30, robot_coll_dist=-1.0, max_episode_steps=3)),
    (K.HRL_ANT_GATHER, dict(n_food=40, n_poison=24, n_bins=64, robot_coll_dist=0.0, max_episode_steps=5)),
    (K.HRL_ANT_MAZE_MJ, dict(n_bins=64, max_episode_steps=4)),
    (K.HRL_ANT_MAZE, dict(sense_target=1, n_bins=33, targets=[(-2.0 + 0.5 * i, -4.0 + 0.1 * i) for i in range(12)], max_episode_steps=3)),
    (K.HRL_ANT_FLAGRUN, dict(use_sensor=1, n_bins=40, flag_timeout=2, flag_max_targets=2, max_episode_steps=6)),
]


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    Side = GpuSide if (len(sys.argv) > 2 and sys.argv[2] == 'gpu') else EmuSide
    cases = MATRIX if (len(sys.argv) > 3 and sys.argv[3] == 'matrix') else [(k, {}) for k in KINDS]
    fails = 0
    for kind, kw in cases:
        for seed in range(seeds):
            for ar in (0, 1):
                r = run(Side, kind, seed, ar, kw=kw)
                if r:
                    fails += 1
                    print('FAIL', r, flush=True)
    print(f'{Side.__name__}: {len(cases) * seeds * 2} runs of 10 steps x 48 envs ({"non-default configs" if cases is MATRIX else "default configs"}), {fails} with a difference')
    return 1 if fails else 0


if __name__ == '__main__':
    sys.exit(main())
