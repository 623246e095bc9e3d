#!/usr/bin/env python3
"""Fit the engine parameters of `hrl_model` to recorded reference steps -- the second half of the road to a pin of the rigid-body step.

tools/make_pybullet_golden.py records, where pybullet + gym + the reference are installed, `(qpos, qvel, items, action) -> (qpos', qvel')` of the
reference's own step() into tests/golden/pybullet_<env>.json; tests/test_pybullet_golden.py replays them on the oracle at the DEFAULT model and
reports the deviation.  The model choices nothing in the reference tree decides are parameters of `hrl_model` (density, both ERPs, frictions,
contact distance, limit margin, the bodies' damping, restitution, joint damping and armature, solver sweeps, the contact cap: include/hrl_envs.h), so closing a deviation is a
search over them, not a kernel edit.  This tool does that search on the fp64 CPU oracle (test infrastructure: it runs oracle/liborc.so) and
prints the fitted parameters as `model_*` keyword arguments of `default_config` / fields of `hrl_config.model`:

    python tests/tools/fit_model.py tests/golden/pybullet_AntGatherBulletEnv-v0.json [--params density,contact_erp,...] [--steps 300]

Continuous parameters: Powell's line searches on the median one-step deviation of (qpos', qvel'), started from the defaults; integer ones
(solver_iters, max_contacts) by enumeration, in turns with the continuous ones.  Without fixtures there is nothing to fit; tests/test_fit_model.py checks the
machinery on steps "recorded" from the oracle itself at moved parameters (it must find them again)."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import orc  # noqa: E402
from hrl_pybullet_envs_amd import _capi as K  # noqa: E402

KIND = {'AntGatherBulletEnv': K.HRL_ANT_GATHER, 'AntMazeBulletEnv': K.HRL_ANT_MAZE, 'PointGatherBulletEnv': K.HRL_POINT_GATHER,
        'AntFlagrunBulletEnv': K.HRL_ANT_FLAGRUN, 'AntMjEnv': K.HRL_ANT_FLAT, 'AntMazeMjEnv': K.HRL_ANT_MAZE_MJ}
# (lower, upper) of the search, in the parameter's own unit; positive ones are searched in log space
CONTINUOUS = {'density': (1.0, 5000.0), 'contact_erp': (0.0, 1.0), 'limit_erp': (0.0, 1.0), 'friction_ground': (0.0, 3.0), 'friction_robot': (0.0, 4.0),
              'contact_dist': (0.0, 0.1), 'limit_margin': (0.0, 1.0), 'linear_damping': (0.0, 20.0), 'angular_damping': (0.0, 20.0),
              'restitution': (0.0, 1.0), 'torque_scale': (10.0, 1000.0), 'limit_max_impulse': (0.1, 1e4),
              'joint_damping': (0.0, 50.0), 'joint_armature': (0.0, 10.0)}
INTEGER = {'solver_iters': range(1, 21), 'max_contacts': range(1, 13)}


def load_steps(path, limit):
    g = json.load(open(path))
    kind = KIND[g['env_id'].split('-')[0]]
    steps = [r for ep in g['episodes'] for r in ep['steps']]
    if len(steps) > limit:
        steps = [steps[i] for i in np.linspace(0, len(steps) - 1, limit).astype(int)]
    return kind, steps


class Replay:
    """All recorded steps as ONE oracle batch: a parameter set is evaluated by one batched step from the recorded states."""

    def __init__(self, kind, steps, base_kw=None):
        self.kind, self.n = kind, len(steps)
        self.base_kw = dict(base_kw or {})
        cfg = orc.default_config(kind, num_envs=self.n, seed=0, auto_reset=0, max_episode_steps=0, **self.base_kw)
        self.o = orc.OracleEnv(cfg, np.float64)
        self.o.reset()
        nq = len(steps[0]['qpos']); nv = len(steps[0]['qvel'])
        self.nq, self.nv = nq, nv
        self.state0 = self.o.state.copy(); self.items0 = self.o.items.copy(); self.aux0 = self.o.aux.copy()
        for i, r in enumerate(steps):
            self.state0[i, :nq] = r['qpos']; self.state0[i, 15:15 + nv] = r['qvel']
            if r.get('items') is not None:
                self.items0[i, :2 * len(r['items'])] = np.asarray(r['items'], np.float64).ravel()
        self.act = np.array([r['action'] for r in steps], np.float64)
        self.want_q = np.array([r['qpos_after'] for r in steps], np.float64)
        self.want_v = np.array([r['qvel_after'] for r in steps], np.float64)

    def deviation(self, model):
        """median over the steps of max |qpos' - recorded| + 0.1 max |qvel' - recorded| (a velocity error of 0.1 m/s weighs like 1 cm)"""
        o = self.o
        for k, v in model.items():
            setattr(o.cfg.model, k, int(v) if k in INTEGER else float(v))
        o.state[...] = self.state0; o.items[...] = self.items0; o.aux[...] = self.aux0
        o.step(self.act)
        dq = np.abs(o.state[:, :self.nq] - self.want_q).max(1)
        dv = np.abs(o.state[:, 15:15 + self.nv] - self.want_v).max(1)
        d = dq + 0.1 * dv
        d[~np.isfinite(d)] = 1e3
        return float(np.median(d)), float(np.median(dq)), float(np.median(dv))


def fit(replay, names, verbose=True):
    from scipy.optimize import minimize
    cont = [n for n in names if n in CONTINUOUS]
    ints = [n for n in names if n in INTEGER]
    m = replay.o.cfg.model
    start = {n: float(getattr(m, n)) for n in cont}
    best_int = {n: int(getattr(m, n)) for n in ints}

    def unpack(x):
        out = {}
        for n, xi in zip(cont, x):
            lo, hi = CONTINUOUS[n]
            out[n] = float(np.clip(np.exp(xi) if lo > 0 else xi, lo, hi))
        return out

    def f(x):
        return replay.deviation({**unpack(x), **best_int})[0]

    x0 = np.array([np.log(max(start[n], 1e-9)) if CONTINUOUS[n][0] > 0 else start[n] for n in cont])
    bounds = [(np.log(CONTINUOUS[n][0]), np.log(CONTINUOUS[n][1])) if CONTINUOUS[n][0] > 0 else CONTINUOUS[n] for n in cont]
    before = replay.deviation({**start, **best_int})
    for rnd in range(3):   # continuous <-> integer in turns: the deviation is V-shaped in every parameter (Powell's line searches; a simplex stalls on the creases)
        if cont:
            r = minimize(f, x0, method='Powell', bounds=bounds, options=dict(xtol=1e-7, ftol=1e-14, maxfev=600 * len(x0)))
            x0 = np.asarray(r.x, float).reshape(-1)
        for n in ints:
            vals = {v: replay.deviation({**unpack(x0), **best_int, n: v})[0] for v in INTEGER[n]}
            best_int[n] = min(vals, key=vals.get)
    fitted = {**unpack(x0), **best_int}
    for n in cont:   # a parameter the records do not constrain (no fast approach: restitution; nothing slides: friction) goes back to where it started
        back = {**fitted, n: start[n]}
        if replay.deviation(back)[0] <= replay.deviation(fitted)[0] * (1 + 1e-9) + 1e-15:
            fitted = back
    after = replay.deviation(fitted)
    if verbose:
        print(f'steps {replay.n}; deviation (median: combined, qpos, qvel) before {before}, after {after}')
        print('fitted hrl_model fields:', ', '.join(f'model_{k}={v!r}' for k, v in fitted.items()))
    return fitted, before, after


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('fixture')
    ap.add_argument('--params', default='density,contact_erp,friction_ground,linear_damping,angular_damping,contact_dist,solver_iters')
    ap.add_argument('--steps', type=int, default=300)
    a = ap.parse_args()
    kind, steps = load_steps(a.fixture, a.steps)
    fit(Replay(kind, steps), a.params.split(','))


if __name__ == '__main__':
    main()
