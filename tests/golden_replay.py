"""The reference's own numbers (tests/golden/*.json, produced by running its Python: tests/golden/make_golden.py) replayed on the WAVE PHASES of the
product: the state of every fixture case is written into the env's records (pose from xy / rpy, items, target index) and the observation is
recomputed from them -- `hrl_set_state` + `hrl_observe` through the C-ABI on the device (tests/test_gpu_golden.py), the same phases under the
host executor on a box without a GPU (tests/test_emu_golden.py).  What is compared are the slices of the observation the in-tree reference code
computes (sense_walls, get_sensor_readings / get_abs_pos, get_target_vec_obs, PointBot.calc_state), at the fp32 tolerance below; a reading that
sits within rounding of a bin edge / quadrant boundary / sensor range may flip in fp32, such flips are counted and bounded.
Test infrastructure only."""
import json
import math
import os

import numpy as np

from hrl_pybullet_envs_amd import _capi as K

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TOL = 2e-5   # fp32 observation pipeline (atan2 / sin / cos to ~3e-7, quaternion round trip) against the reference's float64 numpy


def load(name):
    with open(os.path.join(GOLD, name + '.json')) as f:
        return json.load(f)


def quat_from_rpy(r, p, y):
    """pybullet getQuaternionFromEuler (x, y, z, w), ZYX"""
    cr, sr, cp, sp, cy, sy = math.cos(r / 2), math.sin(r / 2), math.cos(p / 2), math.sin(p / 2), math.cos(y / 2), math.sin(y / 2)
    return [sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy]


def pose_rows(n, xy, rpy, z=0.75, vel=None):
    qpos = np.zeros((n, 15), np.float32); qvel = np.zeros((n, 14), np.float32)
    for i in range(n):
        qpos[i, 0:2] = xy[i]; qpos[i, 2] = z[i] if np.ndim(z) else z
        qpos[i, 3:7] = quat_from_rpy(*rpy[i])
        if vel is not None:
            qvel[i, 0:3] = vel[i]
    return qpos, qvel


def mismatches(got, want, tol=TOL):
    """readings off by more than tol (count), over arrays of equal shape"""
    return int(np.sum(np.abs(np.asarray(got, np.float64) - np.asarray(want, np.float64)) > tol))


def replay_all(make_side, default_config):
    """make_side(cfg) -> object with .n, .set(qpos, qvel, items=None, aux3=None, initial_z=None) and .observe() -> obs [n, D] numpy.
    Returns {fixture: (cases, readings compared, readings off by more than TOL)}."""
    report = {}

    # ---- sizeable_enclosed_scene.py:63-97 sense_walls: maze (7 lines) through AntMaze, the 15 x 15 arena (4 lines) through AntFlagrun's wall sensor
    g = load('sense_walls')
    groups = {}
    for c in g['cases']:
        groups.setdefault((c['scene'], c['bins'], c['span'], c['range']), []).append(c)
    cases = readings = off = 0
    for (scene, bins, span, rng_), cs in groups.items():
        n = len(cs)
        if scene == 'maze':
            cfg = default_config(K.HRL_ANT_MAZE, num_envs=n, n_bins=bins, sensor_span=span, sensor_range=rng_)
            lo = 28
        else:
            side_len = float(g['arena_bounds'][0][0][0]) * 2
            cfg = default_config(K.HRL_ANT_FLAGRUN, num_envs=n, use_sensor=1, n_bins=bins, sensor_span=span, sensor_range=rng_, world_size=(side_len, side_len))
            lo = 28
        s = make_side(cfg)
        qpos, qvel = pose_rows(n, [c['pos'] for c in cs], [(0.0, 0.0, c['yaw']) for c in cs])
        s.set(qpos, qvel)
        obs = s.observe()
        for i, c in enumerate(cs):
            off += mismatches(obs[i, lo:lo + bins], c['out']); readings += bins
        cases += n
    report['sense_walls'] = (cases, readings, off)

    # ---- ant_gather_env.py:128-196 / gather_base.py:118-187 get_sensor_readings and get_abs_pos
    groups = {}
    for c in load('food_sensor'):
        groups.setdefault((c['cls'], c['n_bins'], c['span'], c['range'], len(c['food']), len(c['poison'])), []).append(c)
    cases = readings = off = 0
    for (cls, nb, span, rng_, nf, npo), cs in groups.items():
        n = len(cs)
        kind, base = (K.HRL_ANT_GATHER, 26) if cls == 'AntGatherBulletEnv' else (K.HRL_POINT_GATHER, 8)
        for use_sensor in (1, 0):
            cfg = default_config(kind, num_envs=n, n_bins=nb, sensor_span=span, sensor_range=rng_, n_food=nf, n_poison=npo, use_sensor=use_sensor)
            s = make_side(cfg)
            qpos, qvel = pose_rows(n, [c['robot_xy'] for c in cs], [(0.0, 0.0, c['yaw']) for c in cs], z=0.75 if kind == K.HRL_ANT_GATHER else 0.5)
            items = np.array([np.array(c['food'] + c['poison'], np.float64).reshape(-1) for c in cs], np.float32)
            s.set(qpos, qvel, items=items)
            obs = s.observe()
            for i, c in enumerate(cs):
                if use_sensor:
                    want = c['food_readings'] + c['poison_readings']
                else:
                    want = c['abs_food'] + c['abs_poison']
                off += mismatches(obs[i, base:base + len(want)], want, TOL if use_sensor else 1e-5 * 8)
                readings += len(want)
        cases += n
    report['food_sensor'] = (cases, readings, off)

    # ---- ant_maze_bullet_env.py:123-133 get_target_vec_obs + the walls of the same step (maze_step.json), and the SURVEY spot values
    ms = load('maze_step')
    tgt_index = {(2, -3): 0, (2, 0): 1, (2, 3): 2, (-2, 4): 3}   # ant_maze_bullet_env.py:13-14
    cases = readings = off = 0
    for enc, st in ((0, False), (1, False), (0, True)):
        cs = [c for c in ms if c['encoding'] == enc and c['sense_target'] == st]
        if not cs:
            continue
        n = len(cs)
        cfg = default_config(K.HRL_ANT_MAZE, num_envs=n, target_encoding=enc, sense_target=int(st))
        s = make_side(cfg)
        qpos, qvel = pose_rows(n, [c['torso_xy'] for c in cs], [c['rpy'] for c in cs], z=0.45)
        s.set(qpos, qvel, aux3=np.array([tgt_index[tuple(c['target'])] for c in cs], np.int32))
        obs = s.observe()
        for i, c in enumerate(cs):
            if not st:
                off += mismatches(obs[i, 26:28], c['target_vec_obs']); readings += 2
                off += mismatches(obs[i, 28:38], c['obs'][28:38]); readings += 10
            else:   # the target sensor's intensity needs upstream's parts-centroid distance, which the fixture drew at random: the walls behind it
                off += mismatches(obs[i, 36:46], c['obs'][36:46]); readings += 10
        cases += n
    spot = load('target_vec_spot')
    for enc, key in ((0, 'normed'), (1, 'angle')):
        cfg = default_config(K.HRL_ANT_MAZE, num_envs=1, target_encoding=enc)
        s = make_side(cfg)
        qpos, qvel = pose_rows(1, [(0.3, -0.2)], [(0.0, 0.0, 0.4)], z=0.5)
        s.set(qpos, qvel, aux3=np.array([3], np.int32))
        off += mismatches(s.observe()[0, 26:28], spot[key]); readings += 2; cases += 1
    report['maze_target_and_walls'] = (cases, readings, off)

    # ---- point_bot.py:48-67 PointBot.calc_state (cases whose rpy is the canonical Euler triple of its rotation: |pitch| < pi / 2)
    cs = [c for c in load('pointbot_state') if abs(c['rpy'][1]) < math.pi / 2 - 1e-3]
    n = len(cs)
    cfg = default_config(K.HRL_POINT_GATHER, num_envs=n)
    s = make_side(cfg)
    qpos, qvel = pose_rows(n, [c['xyz'][:2] for c in cs], [c['rpy'] for c in cs], z=[c['xyz'][2] for c in cs], vel=[c['speed'] for c in cs])
    s.set(qpos, qvel, items=np.full((n, 32), 40.0, np.float32), initial_z=1.0)
    obs = s.observe()
    off = sum(mismatches(obs[i, 0:8], c['out'], 5e-5) for i, c in enumerate(cs))   # 0.3 * |v| up to 2.6: a few fp32 ulps more
    report['pointbot_state'] = (n, 8 * n, off)
    return report


def check(report):
    for name, (cases, readings, off) in report.items():
        print(f'{name}: {cases} cases, {readings} readings, {off} off by more than the tolerance')
    assert report['sense_walls'][0] == 182 and report['sense_walls'][2] <= 2          # rays within rounding of a quadrant / range boundary
    assert report['food_sensor'][0] == 131 and report['food_sensor'][2] <= 2          # items within rounding of a bin edge
    assert report['maze_target_and_walls'][0] == 94 and report['maze_target_and_walls'][2] <= 2
    assert report['pointbot_state'][0] >= 12 and report['pointbot_state'][2] == 0
