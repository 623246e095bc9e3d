"""Replay of recorded reference steps (tools/make_pybullet_golden.py's JSON) on the fp64 CPU oracle: every record is one `env.step(a)` of the
reference from a known state; the oracle is put into the identical state -- (qpos, qvel, items) and the task bookkeeping the step reads besides
them (`task`: potential, initial_z, the feet-contact flags of the step before, the flagrun goal state) --, stepped once with the recorded action, and
the deviation of qpos', qvel', obs, reward and done is collected per quantity.  Used by tests/test_pybullet_golden.py (fixtures of the real pybullet,
when someone commits them) and by the dry run of the whole road (tests/test_pin_road_dry_run.py).  Test infrastructure only (runs the oracle)."""
import json
import math

import numpy as np

import orc
from hrl_pybullet_envs_amd import _capi as K

KIND = {'AntGatherBulletEnv': K.HRL_ANT_GATHER, 'AntMazeBulletEnv': K.HRL_ANT_MAZE, 'PointGatherBulletEnv': K.HRL_POINT_GATHER,
        'AntFlagrunBulletEnv': K.HRL_ANT_FLAGRUN, 'AntMazeMjEnv': K.HRL_ANT_MAZE_MJ, 'AntMjEnv': K.HRL_ANT_FLAT}
# SURVEY 8d "Parity tolerance to state": one env step from identical inputs, oracle-fp64 against the reference
TOL = {'qpos': 1e-4, 'qvel': 1e-2, 'obs': 1e-4, 'rew': 1e-4, 'walk_target_dist': 1e-4}


def ant_masses(rho):
    """Solid torso sphere r 0.25 + four jointless capsules (r 0.08, length 0.2 sqrt 2) = the torso body; aux capsule (same size); foot capsule
    (length 0.4 sqrt 2): assets/ant.xml:12-58."""
    rt, rc, l1, l2 = 0.25, 0.08, 0.2 * math.sqrt(2), 0.4 * math.sqrt(2)
    cap = lambda L: rho * (math.pi * rc * rc * L + 4.0 / 3.0 * math.pi * rc ** 3)
    return rho * 4.0 / 3.0 * math.pi * rt ** 3 + 4 * cap(l1), cap(l1), cap(l2)


def ant_total(rho):
    m = ant_masses(rho)
    return m[0] + 4 * (m[1] + m[2])


def config_for(kind, r, model=None):
    """the oracle config one record is replayed under: the reference's constructor defaults (what gym.make builds), the record's own target, and for
    flagrun the MANUAL goal mode -- the recorded goals are data there (items record), not draws of the oracle's own stream"""
    kw = {'model_' + k: v for k, v in (model or {}).items()}
    if kind in (K.HRL_ANT_MAZE, K.HRL_ANT_MAZE_MJ) and r.get('target') is not None:
        kw['targets'] = [tuple(r['target'][:2])]
    if kind == K.HRL_ANT_FLAGRUN:
        kw['flag_manual_goals'] = 1
    return orc.default_config(kind, num_envs=1, seed=0, auto_reset=0, max_episode_steps=0, **kw)


def load_record(o, kind, r, t):
    """puts env 0 of OracleEnv `o` into the recorded pre-step state"""
    task = r.get('task') or {}
    nq, nv = len(r['qpos']), len(r['qvel'])
    o.state[0, :nq] = r['qpos']
    o.state[0, 15:15 + nv] = r['qvel']
    if task.get('initial_z') is not None:
        o.state[0, K.HRL_INITZ_OFF] = task['initial_z']
    if task.get('potential') is not None:
        o.state[0, K.HRL_POTENTIAL_OFF] = task['potential']
    if r.get('items') is not None:
        o.items[0, :2 * len(r['items'])] = np.asarray(r['items'], np.float64).ravel()
    o.aux[0, 0] = int(task.get('t', t))
    feet = task.get('feet_contact')
    if feet is not None and kind in (K.HRL_ANT_MAZE, K.HRL_ANT_FLAGRUN):   # robot.feet_contact as the step before left it: bits 28..31 of aux[1]
        bits = sum((1 << l) for l in range(4) if feet[l])
        o.aux[0, 1] = np.uint32((int(o.aux[0, 1]) & 0x0fffffff) | (bits << 28)).astype(np.int32)
    if kind == K.HRL_ANT_FLAGRUN:
        it = o.items[0]
        it[:] = 0
        it[K.HRL_FLAG_GOAL_OFF:K.HRL_FLAG_GOAL_OFF + 2] = r['walk_target']
        it[K.HRL_FLAG_START_OFF:K.HRL_FLAG_START_OFF + 2] = task.get('_goal_start_pos', [0.0, 0.0])
        it[K.HRL_FLAG_SQDIST_OFF] = task.get('_sq_dist_goal', 0.0)
        pending = 1 if task.get('n_goals_pending', 0) > 0 and task.get('next_goal') is not None else 0
        if pending:   # the one goal next_target() would pop (ant_flagrun_env.py:116); an empty list ends the episode there (IndexError, :193-194)
            it[K.HRL_FLAG_PENDING_OFF:K.HRL_FLAG_PENDING_OFF + 2] = task['next_goal']
        a3 = pending | (min(int(task.get('steps_since_goal_change', 0)), 0x7fff) << 16) | ((1 << 31) if task.get('_rewarded') else 0)
        o.aux[0, 3] = np.uint32(a3).astype(np.int32)
    return nq, nv


def replay(files, model=None):
    """-> (report, median deviation per (env, quantity)).  `model`: hrl_model fields to replay under (None: the build's default specification)."""
    report, med = {}, {}
    for path in files:
        g = json.load(open(path))
        name = g['env_id'].split('-')[0]
        kind = KIND[name]
        rep = report.setdefault(name, {})
        rep['versions'] = g.get('versions')
        # ---- which density did Bullet's MJCF importer use?
        if kind != K.HRL_POINT_GATHER:
            total = g['model']['total_mass']
            rep['total_mass'] = total
            rep['density_hypothesis'] = 1000 if abs(total - ant_total(1000.0)) < abs(total - ant_total(5.0)) else 5
            rep['density_estimate'] = 1000.0 * total / ant_total(1000.0)   # the model is linear in the density
            rep['mass_rel_err_vs_build'] = abs(total - ant_total(1000.0)) / ant_total(1000.0)
        eng = g['model']['engine']
        rep['engine'] = {k: eng.get(k) for k in ('fixedTimeStep', 'numSubSteps', 'numSolverIterations', 'erp', 'contactERP', 'frictionERP', 'gravityAccelerationZ')}
        m0 = orc.default_config(kind, num_envs=1).model   # what the build restates from memory against what the engine reports
        rep['engine_equals_build'] = {'fixedTimeStep': _close(eng.get('fixedTimeStep'), m0.timestep * m0.frame_skip), 'numSubSteps': eng.get('numSubSteps') == m0.frame_skip,
                                      'numSolverIterations': eng.get('numSolverIterations') == m0.solver_iters,
                                      'gravityAccelerationZ': _close(eng.get('gravityAccelerationZ'), -m0.gravity), 'contactERP': _close(eng.get('contactERP'), m0.contact_erp)}
        # ---- replay
        dev = {k: [] for k in ('qpos', 'qvel', 'obs', 'rew', 'walk_target_dist')}
        done_flips = steps = pickups = feet_flips = 0
        base = {K.HRL_ANT_GATHER: 26, K.HRL_POINT_GATHER: 8}.get(kind)
        for ep in g['episodes']:
            for t, r in enumerate(ep['steps']):
                o = orc.OracleEnv(config_for(kind, r, model), np.float64)
                o.reset()
                nq, nv = load_record(o, kind, r, t)
                o.step(np.asarray(r['action'], np.float64)[None])
                dev['qpos'].append(np.abs(o.state[0, :nq] - np.asarray(r['qpos_after'])).max())
                dev['qvel'].append(np.abs(o.state[0, 15:15 + nv] - np.asarray(r['qvel_after'])).max())
                ob = np.asarray(r['obs'], np.float64)
                if ob.shape == o.obs[0].shape:
                    # an item picked up in this step was moved by the reference's OWN random stream (MT19937; here Philox): the sensor part of such an
                    # observation shows another item layout and is left out, the robot part is compared as ever
                    moved = base is not None and r.get('items_after') is not None and r['items_after'] != r['items']
                    n = base if moved else ob.shape[0]
                    pickups += int(moved)
                    dev['obs'].append(np.nanmax(np.abs(o.obs[0, :n] - ob[:n])))
                dev['rew'].append(abs(float(o.rew[0]) - r['rew']))
                done_flips += int(bool(o.done[0]) != r['done']); steps += 1
                rob = r.get('robot') or {}
                # upstream's own bookkeeping after the step, where the oracle keeps a counterpart: robot.feet_contact (bits 28..31 of aux[1]) and
                # robot.walk_target_dist (from the PARTS CENTROID; the oracle's potential = -dist / dt) -- the locomotion kinds
                if kind in (K.HRL_ANT_MAZE, K.HRL_ANT_FLAGRUN) and rob.get('feet_contact') is not None:
                    bits = (int(o.aux[0, 1]) >> 28) & 0xf
                    feet_flips += int([(bits >> l) & 1 for l in range(4)] != [int(bool(f)) for f in rob['feet_contact']])
                if kind not in (K.HRL_ANT_GATHER, K.HRL_POINT_GATHER) and rob.get('walk_target_dist') is not None and not r['done']:
                    m = o.cfg.model
                    dev['walk_target_dist'].append(abs(-float(o.state[0, K.HRL_POTENTIAL_OFF]) * m.timestep * m.frame_skip - rob['walk_target_dist']))
        rep['steps'] = steps; rep['done_flips'] = done_flips; rep['pickup_steps'] = pickups; rep['feet_flips'] = feet_flips
        for k, v in dev.items():
            v = np.asarray(v)
            rep[k] = {'max': float(v.max()), 'median': float(np.median(v)), 'p99': float(np.percentile(v, 99))} if len(v) else None
            if len(v):
                med[(name, k)] = float(np.median(v))
    return report, med


def _close(a, b):
    return a is not None and abs(float(a) - float(b)) <= 1e-6 * max(1.0, abs(float(b)))
