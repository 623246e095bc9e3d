"""Oracle task logic vs golden vectors generated from the reference's own Python (tests/golden/make_golden.py).

float64 oracle vs float64 reference outputs: agreement to ~1e-12 (same formulas, same precision).
"""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

import orc
from hrl_pybullet_envs_amd import _capi as K

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
F64 = np.float64


def load(name):
    with open(os.path.join(GOLD, name + '.json')) as f:
        return json.load(f)


def arr(x, dt=F64):
    return np.ascontiguousarray(np.array(x, dtype=dt))


def test_intersection_utils():
    g = load('intersection')
    L = orc.lib()
    for c in g['lines']:
        p = arr(c['p']).reshape(8)
        out = np.zeros(2)
        found = L.orc_inf_intersection_f64(orc.ptr(p), orc.ptr(out))
        assert bool(found) == (c['inf'] is not None)
        if found:
            np.testing.assert_allclose(out, c['inf'], rtol=1e-12, atol=1e-12)
        assert bool(L.orc_segment_intersection_f64(orc.ptr(p))) == c['seg']
    for c in g['quadrant']:
        assert L.orc_quadrant_f64(C.c_double(c['p'][0]), C.c_double(c['p'][1])) == c['q']
    # known answers recorded in SURVEY.md 8c
    p = arr([0, 0, 1, 1, 0, 2, 2, 0]); out = np.zeros(2)
    assert L.orc_inf_intersection_f64(orc.ptr(p), orc.ptr(out)) == 1 and np.allclose(out, [1, 1])
    assert [L.orc_quadrant_f64(C.c_double(x), C.c_double(y)) for x, y in [(1, 1), (1, -1), (-1, 1), (-1, -1), (0, 0)]] \
        == [1, 4, 2, 3, 1]


@pytest.mark.parametrize('dtype,tol', [(np.float64, 1e-11), (np.float32, 2e-5)])
def test_sense_walls(dtype, tol):
    g = load('sense_walls')
    L = orc.lib()
    lines = {'maze': arr(g['maze_bounds'], dtype).reshape(-1, 4), 'arena': arr(g['arena_bounds'], dtype).reshape(-1, 4)}
    assert lines['maze'].shape == (7, 4) and lines['arena'].shape == (4, 4)
    cr = orc.creal(dtype)
    bad = 0
    for c in g['cases']:
        ln = lines[c['scene']]
        out = np.zeros(c['bins'], dtype)
        pos = arr(c['pos'], dtype)
        orc.fn('orc_sense_walls', dtype)(c['bins'], cr(c['span']), cr(c['range']), orc.ptr(pos), cr(c['yaw']), orc.ptr(ln),
                                         ln.shape[0], int(c['span'] == 2 * math.pi), orc.ptr(out))
        if dtype == np.float64:
            np.testing.assert_allclose(out, c['out'], rtol=0, atol=tol)
        else:  # fp32: a ray within rounding of a quadrant/range boundary may flip one reading
            bad += int(np.sum(np.abs(out - np.array(c['out'])) > tol))
    assert bad <= 2


@pytest.mark.parametrize('dtype,tol', [(np.float64, 1e-12), (np.float32, 1e-5)])
def test_food_sensor_and_abs_pos(dtype, tol):
    check_food_sensor(load('food_sensor'), dtype, tol)


def check_food_sensor(g, dtype, tol):
    cr = orc.creal(dtype)
    flips = 0
    for c in g:
        items = arr(c['food'] + c['poison'], dtype)
        nf, npo = len(c['food']), len(c['poison'])
        xy = arr(c['robot_xy'], dtype)
        d2 = np.array([orc.fn('orc_sq_dist', dtype)(orc.ptr(items[i]), orc.ptr(xy)) for i in range(nf + npo)], dtype)
        np.testing.assert_allclose(d2, c['sq_dists'], rtol=1e-5 if dtype == np.float32 else 1e-13)
        fo, po = np.zeros(c['n_bins'], dtype), np.zeros(c['n_bins'], dtype)
        orc.fn('orc_food_sensor', dtype)(c['n_bins'], cr(c['span']), cr(c['range']), orc.ptr(xy), cr(c['yaw']),
                                         orc.ptr(items), nf, npo, orc.ptr(d2), orc.ptr(fo), orc.ptr(po))
        ok = np.allclose(fo, c['food_readings'], atol=tol, rtol=0) and np.allclose(po, c['poison_readings'], atol=tol, rtol=0)
        if dtype == np.float64:
            assert ok
        else:
            flips += int(not ok)
        af, ap = np.zeros(2 * min(nf, c['n_bins']), dtype), np.zeros(2 * min(npo, c['n_bins']), dtype)
        orc.fn('orc_abs_pos', dtype)(c['n_bins'], orc.ptr(items), nf, npo, orc.ptr(d2), orc.ptr(af), orc.ptr(ap))
        np.testing.assert_allclose(af, c['abs_food'], atol=tol)
        np.testing.assert_allclose(ap, c['abs_poison'], atol=tol)
    assert flips <= 1  # fp32 bin-edge flips are measure-zero events


def test_gather_scene_respawn():
    """gather_scene.py:38-62,95-114 replayed with the exact uniform draws the reference consumed."""
    check_gather_scene(load('gather_scene'))


def test_random_constructor_arguments_gather_scene():
    """`random_config.json`: GatherScene with world size, item counts (up to 60), spacing and respawn drawn at random: spawn, restart and
    reward_collision replayed with the uniform draws the reference consumed."""
    check_gather_scene(load('random_config')['gather_scene'])


def check_gather_scene(cases):
    for sc in cases:
        ws = arr(sc['world'])
        draws = sc['restart_draws']
        n_items = sc['n_food'] + sc['n_poison']
        # episode_restart: spawn (avoid (0,0)) for each missing item, then move every item again (avoid (0,0))
        k = 0
        pos_after = []
        for phase in range(2):
            pos_after = []
            for i in range(n_items):
                out = np.zeros(2)
                d = arr(draws[k:])
                used = orc.lib().orc_random_on_plane_f64(orc.ptr(ws), orc.ptr(arr([0, 0])), C.c_double(sc['spacing']),
                                                         orc.ptr(d), len(d), orc.ptr(out))
                assert used > 0
                k += used
                pos_after.append(out.copy())
        assert k == len(draws)
        ref = sc['after_restart']['food'] + sc['after_restart']['poison']
        np.testing.assert_allclose(np.array(pos_after), np.array(ref)[:, :2], atol=1e-12)
        assert all(abs(p[2] - 0.1) < 1e-15 for p in ref)
        for ev in sc['events']:
            if ev['obj_index'] >= n_items:
                assert ev['rew'] == 0
                continue
            assert ev['rew'] == (1 if ev['obj_index'] < sc['n_food'] else -1)
            if sc['respawn']:
                d = arr(ev['draws']); out = np.zeros(2)
                used = orc.lib().orc_random_on_plane_f64(orc.ptr(ws), orc.ptr(arr(ev['agent_xyz'][:2])),
                                                         C.c_double(sc['spacing']), orc.ptr(d), len(d), orc.ptr(out))
                assert used == len(ev['draws'])
                np.testing.assert_allclose(out, ev['new_pos'][:2], atol=1e-12)
                hw = (np.array(sc['world']) - 1) / 2
                assert np.all(np.abs(out) <= hw + 1e-12)
            else:
                assert ev['new_pos'] == [100.0, 0.0, -10.0]


def test_gather_step_task_half():
    """ant_gather_env.py:81-119 / gather_base.py:80-109: pickups, respawn, sensor, alive/done/reward."""
    check_gather_step(load('gather_step'))


def check_gather_step(cases, **kw):
    for c in cases:
        ant = c['cls'] == 'AntGatherBulletEnv'
        cfg = orc.default_config(K.HRL_ANT_GATHER if ant else K.HRL_POINT_GATHER, **kw)
        assert cfg.n_bins == c['n_bins'] and cfg.n_food + cfg.n_poison == len(c['items_before'])
        st = arr(c['state_in']); items = arr(c['items_before']).copy()
        draws = arr(c['respawn_draws']).reshape(-1, 2) if c['respawn_draws'] else np.zeros((1, 2))
        nobs = (26 if ant else 8) + 2 * c['n_bins']
        obs = np.zeros(nobs); rew = C.c_double(); done = C.c_int(); fr = C.c_double(); dr = C.c_double()
        used = orc.lib().orc_gather_task_f64(C.byref(cfg), int(ant), orc.ptr(st), len(st), orc.ptr(arr(c['torso_xyz'])),
                                             C.c_double(c['rpy'][2]), C.c_double(c['initial_z']),
                                             C.c_double(0.26 if ant else -1.0), orc.ptr(items), orc.ptr(draws),
                                             len(c['respawn_draws']), orc.ptr(obs), C.byref(rew), C.byref(done),
                                             C.byref(fr), C.byref(dr))
        assert used == len(c['respawn_draws'])
        np.testing.assert_allclose(items, c['items_after'], atol=1e-12)
        np.testing.assert_allclose(obs, c['obs'], atol=1e-12, equal_nan=True)
        assert rew.value == c['rew'] and bool(done.value) == c['done']
        assert fr.value == c['food_rew'] and dr.value == c['dead_rew']


def test_maze_step_task_half():
    """ant_maze_bullet_env.py:63-97,123-178 with the upstream step result supplied."""
    check_maze_step(load('maze_step'))
    spot = load('target_vec_spot')
    for enc, key in ((0, 'normed'), (1, 'angle')):
        tv = np.zeros(2)
        orc.lib().orc_target_vec_obs_f64(enc, orc.ptr(arr([-2, 4])), orc.ptr(arr([0.3, -0.2])), C.c_double(0.4), orc.ptr(tv))
        np.testing.assert_allclose(tv, spot[key], atol=1e-12)


def check_maze_step(cases, n_bins=10, **kw):
    g = load('sense_walls')
    lines = arr(g['maze_bounds']).reshape(-1, 4)
    for c in cases:
        cfg = orc.default_config(K.HRL_ANT_MAZE, target_encoding=c['encoding'], sense_target=int(c['sense_target']), n_bins=n_bins, **kw)
        nobs = orc.obs_dim(cfg)
        assert nobs == len(c['obs'])
        obs = np.zeros(nobs); rew = C.c_double(); done = C.c_int()
        orc.lib().orc_maze_task_f64(C.byref(cfg), orc.ptr(arr(c['ant_obs'])), C.c_double(c['inner_rew']), 0,
                                    orc.ptr(arr(c['torso_xy'])), C.c_double(c['rpy'][2]), orc.ptr(arr(c['target'])),
                                    C.c_double(c['walk_target_dist']), 7, orc.ptr(lines), 7, 3, orc.ptr(obs),
                                    C.byref(rew), C.byref(done))
        np.testing.assert_allclose(obs, c['obs'], atol=1e-11)
        assert rew.value == pytest.approx(c['rew'], abs=1e-12) and bool(done.value) == c['done']
        tv = np.zeros(2)
        orc.lib().orc_target_vec_obs_f64(c['encoding'], orc.ptr(arr(c['target'])), orc.ptr(arr(c['torso_xy'])),
                                         C.c_double(c['rpy'][2]), orc.ptr(tv))
        np.testing.assert_allclose(tv, c['target_vec_obs'], atol=1e-12)
        ts = np.zeros(n_bins)
        orc.lib().orc_target_sensor_obs_f64(n_bins, C.c_double(2 * math.pi), C.c_double(5.0), orc.ptr(arr(c['target'])),
                                            orc.ptr(arr(c['torso_xy'])), C.c_double(c['rpy'][2]),
                                            C.c_double(c['walk_target_dist']), orc.ptr(lines[4:]), 3, orc.ptr(ts))
        np.testing.assert_allclose(ts, c['target_sensor_obs'], atol=1e-12)


def test_pointbot_calc_state():
    for c in load('pointbot_state'):
        out = np.zeros(8, np.float32)
        orc.lib().orc_pointbot_state_f32(orc.ptr(arr(c['xyz'], np.float32)), orc.ptr(arr(c['rpy'], np.float32)),
                                         orc.ptr(arr(c['speed'], np.float32)), orc.ptr(arr(c['target'], np.float32)),
                                         C.c_float(1.0), orc.ptr(out))
        np.testing.assert_allclose(out, c['out'], atol=3e-6)
        out64 = np.zeros(8)
        orc.lib().orc_pointbot_state_f64(orc.ptr(arr(c['xyz'])), orc.ptr(arr(c['rpy'])), orc.ptr(arr(c['speed'])),
                                         orc.ptr(arr(c['target'])), C.c_double(1.0), orc.ptr(out64))
        np.testing.assert_allclose(out64, c['out'], atol=1e-6)  # reference rounds to float32


def test_antmj_reward():
    for c in load('antmj_step'):
        rew = C.c_double(); done = C.c_int()
        orc.lib().orc_antmj_reward_f64(orc.ptr(arr(c['state'])), C.c_double(c['potential_old']),
                                       C.c_double(c['potential_new']), c['joints_at_limit'],
                                       C.c_double(c['joints_at_limit_cost']), C.byref(rew), C.byref(done))
        assert rew.value == pytest.approx(c['rew'], abs=1e-9) and bool(done.value) == c['done']


def test_maze_mj_step_task_half():
    """ant_maze_mj_env.py:57-78 on top of MjAnt.py:36-97: observation assembly, t/1000, sparse reward."""
    check_maze_mj_step(load('maze_mj_step'))


def test_random_constructor_arguments_maze_mj_step():
    """`random_config.json`: AntMazeMjEnv.step (ant_maze_mj_env.py:57-78) with n_bins, sensor span / range, tol and inner_rew_weight drawn at random."""
    check_maze_mj_step(load('random_config')['maze_mj_step'])


def check_maze_mj_step(cases):
    lines = arr(load('sense_walls')['maze_bounds']).reshape(-1, 4)
    for c in cases:
        nb = c.get('n_bins', 10)
        extra = {k2: c[k1] for k1, k2 in (('span', 'sensor_span'), ('range', 'sensor_range'), ('tol', 'tol')) if k1 in c}   # (the random-argument cases carry them)
        cfg = orc.default_config(K.HRL_ANT_MAZE_MJ, inner_rew_weight=c['inner_rew_weight'], n_bins=nb, **extra)
        assert orc.obs_dim(cfg) == 30 + 3 * nb == len(c['obs'])
        inner = C.c_double(); idone = C.c_int()
        orc.lib().orc_antmj_reward_f64(orc.ptr(arr(c['state'])), C.c_double(c['potential_old']), C.c_double(c['potential_new']),
                                       c['joints_at_limit'], C.c_double(-0.1), C.byref(inner), C.byref(idone))
        obs = np.zeros(30 + 3 * nb); rew = C.c_double(); done = C.c_int()
        orc.lib().orc_maze_mj_task_f64(C.byref(cfg), orc.ptr(arr(c['state'])), C.c_double(c['rpy'][2]), inner, idone,
                                       C.c_double(c['walk_target_dist']), c['t_before'], orc.ptr(lines), 7, orc.ptr(obs),
                                       C.byref(rew), C.byref(done))
        np.testing.assert_allclose(obs, c['obs'], atol=1e-11)
        assert rew.value == pytest.approx(c['rew'], abs=1e-9) and bool(done.value) == c['done']
        assert c['t_after'] == c['t_before'] + 1


def test_flagrun_step_bookkeeping_and_targets():
    """ant_flagrun_env.py:162-204 (goal reward, retarget on reach / timeout, running out of goals) and :71-78."""
    for c in load('flagrun_step'):
        cfg = orc.default_config(K.HRL_ANT_FLAGRUN, tol=c['tol'], flag_timeout=c['timeout'], flag_switch_on_collision=int(c['switch']))
        steps = C.c_int(c['steps_before']); rewarded = C.c_int(int(c['rewarded_before'])); left = C.c_int(c['n_goals'])
        rew = C.c_double(); done = C.c_int(); retarget = C.c_int()
        orc.lib().orc_flagrun_task_f64(C.byref(cfg), C.c_double(c['inner_rew']), int(c['inner_done']), C.c_double(c['walk_target_dist']),
                                       C.byref(steps), C.byref(rewarded), C.byref(left), C.byref(rew), C.byref(done), C.byref(retarget))
        assert rew.value == pytest.approx(c['rew'], abs=1e-9) and bool(done.value) == c['done']
        assert steps.value == c['steps_after'] and bool(rewarded.value) == c['rewarded_after'] and left.value == c['goals_left']
        assert bool(retarget.value) == c['retargeted'] == c['obs_is_new_state']
        if c['retargeted']:
            assert c['target_after'] == c['last_goal']  # goals.pop(): the LAST goal of the list is next
    for c in load('flagrun_create_target'):
        g = np.zeros(2)
        used = orc.lib().orc_flag_create_target_f64(C.c_double(c['size']), orc.ptr(arr([(u + c['size'] / 2) / c['size'] for u in c['draws']])),
                                                    len(c['draws']), orc.ptr(g))
        assert used == len(c['draws'])
        np.testing.assert_allclose(g, c['goal'], atol=1e-12)
        assert np.linalg.norm(g) >= 0.5


def test_flagrun_close_targets():
    """ant_flagrun_env.py:80-89 `create_close_target` and the step bookkeeping in max_target_dist mode (:111-112)."""
    def close_target(c, size, mtd):
        g = np.zeros(2)
        n = len(c['u']) // 2
        used = orc.lib().orc_flag_create_close_target_f64(C.c_double(size), C.c_double(c['tol']), C.c_double(mtd), orc.ptr(arr(c['robot_xy'])),
                                                          orc.ptr(arr(c['u'])), np.asarray(c['b'], np.int32).ctypes.data_as(C.c_void_p), n, orc.ptr(g))
        assert used == n  # accepted exactly on the attempt the reference stopped at
        return g
    rejections = 0
    for c in load('flagrun_create_close_target'):
        g = close_target(c, c['size'], c['max_target_dist'])
        np.testing.assert_allclose(g, c['goal'], atol=1e-12)
        assert np.all(np.abs(g) < c['size'] / 2)
        rejections += len(c['u']) // 2 - 1
    assert rejections > 10  # the edge-hugging cases exercised the redraw loop
    n_retarget = 0
    for c in load('flagrun_close_step'):
        cfg = orc.default_config(K.HRL_ANT_FLAGRUN, tol=c['tol'], flag_timeout=c['timeout'], flag_switch_on_collision=int(c['switch']),
                                 flag_max_targets=0, flag_max_target_dist=3.0)
        steps = C.c_int(c['steps_before']); rewarded = C.c_int(int(c['rewarded_before'])); left = C.c_int(1 << 20)
        rew = C.c_double(); done = C.c_int(); retarget = C.c_int()
        orc.lib().orc_flagrun_task_f64(C.byref(cfg), C.c_double(c['inner_rew']), int(c['inner_done']), C.c_double(c['walk_target_dist']),
                                       C.byref(steps), C.byref(rewarded), C.byref(left), C.byref(rew), C.byref(done), C.byref(retarget))
        assert rew.value == pytest.approx(c['rew'], abs=1e-9) and bool(done.value) == c['done'] == c['inner_done']  # never out of goals
        assert steps.value == c['steps_after'] and bool(rewarded.value) == c['rewarded_after']
        assert bool(retarget.value) == c['retargeted']
        if c['retargeted']:
            n_retarget += 1
            np.testing.assert_allclose(close_target(c, 10.0, 3.0), c['target_after'], atol=1e-12)
    assert n_retarget >= 8



@pytest.mark.parametrize('dtype,tol', [(np.float64, 1e-6), (np.float32, 2e-3)])  # hrl_config holds the weights as float: 0.3 and 0.05 are good to 3e-8 relative
def test_flagrun_class_level_reward_weights(dtype, tol):
    """`flagrun_weights_step.json`: the reference's own step() (ant_flagrun_env.py:162-204) with the class-level weights moved off their defaults
    (:157-160), random path-reward state (:100-103) -- including an env that never received a goal (0 / 0 -> NaN or +-inf rewards) -- and what
    set_target() leaves behind when the step switches goals; info['target'] (:191,199).  Replayed on the oracle's record functions."""
    L, cr = orc.lib(), orc.creal(dtype)
    pr = orc.fn('orc_flag_path_rew', dtype); pr.restype = cr
    n_target = n_nonfinite = 0
    for c in load('flagrun_weights_step'):
        w = c['weights']
        cfg = orc.default_config(K.HRL_ANT_FLAGRUN, tol=c['tol'], flag_timeout=c['timeout'], flag_switch_on_collision=int(c['switch']),
                                 flag_ant_env_rew_weight=w['ant_env_rew_weight'], flag_path_rew_weight=w['path_rew_weight'],
                                 flag_dist_rew_weight=w['dist_rew_weight'], flag_goal_reach_rew=w['goal_reach_rew'])
        xy, goal, start = arr(c['robot_xy'], dtype), arr(c['goal'], dtype), arr(c['goal_start_pos'], dtype)
        with np.errstate(all='ignore'):
            path = pr(orc.ptr(xy), orc.ptr(goal), orc.ptr(start), cr(c['sq_dist_goal']))
        steps = C.c_int(c['steps_before']); rewarded = C.c_int(int(c['rewarded_before'])); left = C.c_int(c['n_goals'])
        rew = cr(); done = C.c_int(); retarget = C.c_int()
        orc.fn('orc_flagrun_task_w', dtype)(C.byref(cfg), cr(c['inner_rew']), int(c['inner_done']), cr(c['walk_target_dist']), 1, cr(path),
                                            C.byref(steps), C.byref(rewarded), C.byref(left), C.byref(rew), C.byref(done), C.byref(retarget))
        if np.isfinite(c['rew']):
            assert rew.value == pytest.approx(c['rew'], abs=tol * max(1.0, abs(c['rew']))), c
        else:
            n_nonfinite += 1
            assert (np.isnan(c['rew']) and np.isnan(rew.value)) or rew.value == c['rew'], (c['rew'], rew.value)
        assert bool(done.value) == c['done'] and steps.value == c['steps_after'] and bool(rewarded.value) == c['rewarded_after'] and left.value == c['goals_left']
        assert bool(retarget.value) == (c['info_target'] is not None)
        st3 = np.array(c['goal_start_pos'] + [c['sq_dist_goal']], dtype)
        if retarget.value:   # next_target() -> set_target(goals.pop()): the LAST goal; `_goal_start_pos` / `_sq_dist_goal` from where the robot stands
            n_target += 1
            new_goal = arr(c['goals'][-1], dtype)
            np.testing.assert_allclose(new_goal, c['info_target'], atol=1e-6)
            orc.fn('orc_flag_set_target_state', dtype)(orc.ptr(new_goal), orc.ptr(xy), orc.ptr(st3))
        np.testing.assert_allclose(st3[:2], c['goal_start_pos_after'], atol=1e-6)
        assert st3[2] == pytest.approx(c['sq_dist_goal_after'], rel=1e-12 if dtype == np.float64 else 1e-5)
    assert n_target >= 20 and n_nonfinite >= 5


def test_flagrun_manual_goal_sequences():
    """manual_goal_creation sequences run on the reference itself (make_golden.py `flagrun_manual_seq`): `env.goals = [...];
    env.next_target()` takes the LAST goal of the list (goals.pop(), ant_flagrun_env.py:116), the list is consumed back to front,
    an empty list ends the episode (IndexError, :193-194) or raises from next_target(); a reset keeps the walk target and
    steps_since_goal_change and drops the list (:132-155); with max_targets < 1 next_target() ignores the list and draws near
    the robot (:113-114).  The oracle's record functions -- the ones orc_env_step_one / orc_env_set_goals_one call -- replay
    every event."""
    g = load('flagrun_manual_seq')
    dt = g['dt']
    L = orc.lib()
    n_pops = n_raise = 0
    for seq in g['list']:
        cfg = orc.default_config(K.HRL_ANT_FLAGRUN, flag_manual_goals=1, flag_timeout=seq['timeout'], flag_switch_on_collision=int(seq['switch']))
        o = orc.OracleEnv(cfg, np.float64)
        st, items, aux = o.state[0], o.items[0], o.aux[0]
        items[0] = 1000.0   # the walk target before the first episode: upstream's default (what a fresh record holds after reset)
        prev = None
        for e in seq['events']:
            if e['op'] == 'reset':
                before = items[0:2].copy(); steps_before = (aux[3] >> 16) & 0x7fff
                o.reset()
                assert np.array_equal(items[0:2], before) and np.all(items[K.HRL_FLAG_PENDING_OFF:] == 0) and (aux[3] & 0xffff) == 0   # target kept, list dropped
                assert ((aux[3] >> 16) & 0x7fff) == steps_before == e['steps'] and not (aux[3] >> 31) & 1
                np.testing.assert_allclose(items[0:2], e['target'], atol=1e-12)
            elif e['op'] in ('set_goals', 'next_target'):
                gl = arr(e['goals']).reshape(-1, 2)
                L.orc_flag_goals_assign_f64(C.byref(cfg), orc.ptr(items), orc.ptr(aux), orc.ptr(gl), len(gl))
                ok = L.orc_flag_next_target_f64(C.byref(cfg), C.c_int64(0), orc.ptr(st), orc.ptr(items), orc.ptr(aux))
                assert bool(ok) == (not e.get('raised', False))
                n_raise += int(not ok)
                np.testing.assert_allclose(items[0:2], e['target'], atol=1e-12)
                if ok:
                    np.testing.assert_allclose(items[0:2], gl[-1], atol=1e-12)   # goals.pop(): the LAST goal
                    assert not (aux[3] >> 31) & 1
                assert (aux[3] & 0xffff) == len(e['goals_left']) and ((aux[3] >> 16) & 0x7fff) == e['steps']
                np.testing.assert_allclose(items[K.HRL_FLAG_PENDING_OFF:K.HRL_FLAG_PENDING_OFF + 2 * len(e['goals_left'])].reshape(-1, 2), arr(e['goals_left']).reshape(-1, 2), atol=1e-12)
            else:
                wtd = float(np.linalg.norm(arr(e['pos'][:2]) - items[0:2]))
                base = 1.0 + (e['potential'] - prev['potential'])
                assert e['potential'] == pytest.approx(-wtd / dt, abs=1e-9)    # measured against the target in effect before the step
                steps = C.c_int((aux[3] >> 16) & 0x7fff); rewarded = C.c_int((aux[3] >> 31) & 1); left = C.c_int(aux[3] & 0xffff)
                rew = C.c_double(); done = C.c_int(); retarget = C.c_int()
                L.orc_flagrun_task_f64(C.byref(cfg), C.c_double(base), 0, C.c_double(wtd), C.byref(steps), C.byref(rewarded), C.byref(left),
                                       C.byref(rew), C.byref(done), C.byref(retarget))
                if retarget.value:   # exactly what orc_env_step_one does next
                    assert L.orc_flag_next_target_f64(C.byref(cfg), C.c_int64(0), orc.ptr(st), orc.ptr(items), orc.ptr(aux)) == 1
                    n_pops += 1
                aux[3] = np.array([(int(aux[3]) & 0xffff) | (steps.value << 16) | (rewarded.value << 31)], np.uint32).view(np.int32)[0]
                assert rew.value == pytest.approx(e['rew'], abs=1e-6) and bool(done.value) == e['done'] and bool(retarget.value) == e['retargeted']
                assert steps.value == e['steps'] and bool(rewarded.value) == e['rewarded'] and (aux[3] & 0xffff) == len(e['goals_left'])
                np.testing.assert_allclose(items[0:2], e['target'], atol=1e-12)
            prev = e
    assert n_pops >= 8 and n_raise >= 8
    n_draws = 0
    for seq in g['close']:
        cfg = orc.default_config(K.HRL_ANT_FLAGRUN, flag_manual_goals=1, flag_max_targets=0, flag_max_target_dist=seq['max_target_dist'],
                                 flag_timeout=seq['timeout'], tol=seq['tol'], flag_size=seq['size'])
        o = orc.OracleEnv(cfg, np.float64)
        st, items, aux = o.state[0], o.items[0], o.aux[0]
        items[0] = 1000.0
        target, prev = [1000.0, 0.0], None
        for e in seq['events']:
            drew = len(e['u']) > 0
            if e['op'] == 'reset':
                o.reset()
                assert not drew and e['target'] == target and np.array_equal(items[0:2], target) and (aux[3] & 0xffff) == 0  # reset draws nothing (:150-153)
            elif e['op'] == 'next_target':
                assert drew and e['list_len_after'] == 1   # the list was not touched: create_close_target (:113-114)
                before = aux[3] & 0xffff
                assert L.orc_flag_next_target_f64(C.byref(cfg), C.c_int64(0), orc.ptr(st), orc.ptr(items), orc.ptr(aux)) == 1
                assert (aux[3] & 0xffff) == before + 1   # the goal counter of the episode keys the draw
                d = np.abs(items[0:2] - st[0:2])
                assert np.all(d >= seq['tol'] - 1e-9) and np.all(d <= seq['max_target_dist'] / 2 + 1e-9) and np.all(np.abs(items[0:2]) < seq['size'] / 2)
            else:
                wtd = float(np.linalg.norm(arr(e['pos'][:2]) - arr(target)))
                base = 1.0 + (e['potential'] - prev['potential'])
                steps = C.c_int(prev['steps']); rewarded = C.c_int(int(prev['rewarded'])); left = C.c_int(1 << 20)
                rew = C.c_double(); done = C.c_int(); retarget = C.c_int()
                L.orc_flagrun_task_f64(C.byref(cfg), C.c_double(base), 0, C.c_double(wtd), C.byref(steps), C.byref(rewarded), C.byref(left),
                                       C.byref(rew), C.byref(done), C.byref(retarget))
                assert rew.value == pytest.approx(e['rew'], abs=1e-6) and not done.value and not e['done'] and bool(retarget.value) == e['retargeted'] == drew
                assert steps.value == e['steps'] and bool(rewarded.value) == e['rewarded']
            if drew:   # the reference's draw, replayed with the uniforms / sign bits it consumed
                n_draws += 1
                gg = np.zeros(2); nat = len(e['u']) // 2
                used = L.orc_flag_create_close_target_f64(C.c_double(seq['size']), C.c_double(seq['tol']), C.c_double(seq['max_target_dist']),
                                                          orc.ptr(arr(e['pos'][:2])), orc.ptr(arr(e['u'])), np.asarray(e['b'], np.int32).ctypes.data_as(C.c_void_p), nat, orc.ptr(gg))
                assert used == nat
                np.testing.assert_allclose(gg, e['target'], atol=1e-12)
            target, prev = e['target'], e
            items[0:2] = target   # the record follows the reference's RandomState draws (the streams are not reproduced)
    assert n_draws >= 20


def test_gather_contact_pickup_step():
    """robot_coll_dist <= 0 (ant_gather_env.py:113-116, gather_base.py:103-106): one reward_collision() per contact point of
    the robot, AFTER the observation was assembled; items touched several times are paid and moved several times."""
    multi = 0
    for c in load('gather_contact_step'):
        ant = c['cls'] == 'AntGatherBulletEnv'
        cfg = orc.default_config(K.HRL_ANT_GATHER if ant else K.HRL_POINT_GATHER, robot_coll_dist=0.0 if ant else -1.0, respawn=int(c['respawn']))
        st = arr(c['state_in']); items = arr(c['items_before']).copy()
        draws = arr(c['respawn_draws']).reshape(-1, 2) if c['respawn_draws'] else np.zeros((1, 2))
        ci = np.asarray(c['contact_items'], np.int32) if c['contact_items'] else np.zeros(1, np.int32)
        nobs = (26 if ant else 8) + 2 * c['n_bins']
        obs = np.zeros(nobs); rew = C.c_double(); done = C.c_int(); fr = C.c_double(); dr = C.c_double()
        used = orc.lib().orc_gather_task_contacts_f64(C.byref(cfg), int(ant), orc.ptr(st), len(st), orc.ptr(arr(c['torso_xyz'])),
                                                      C.c_double(c['rpy'][2]), C.c_double(c['initial_z']),
                                                      C.c_double(0.26 if ant else -1.0), orc.ptr(items), orc.ptr(draws),
                                                      len(c['respawn_draws']), orc.ptr(ci), len(c['contact_items']),
                                                      orc.ptr(obs), C.byref(rew), C.byref(done), C.byref(fr), C.byref(dr))
        assert used == len(c['respawn_draws'])
        np.testing.assert_allclose(items, c['items_after'], atol=1e-12)
        np.testing.assert_allclose(obs, c['obs'], atol=1e-12, equal_nan=True)  # sensor readings of the OLD item positions
        assert rew.value == c['rew'] and bool(done.value) == c['done'] and fr.value == c['food_rew'] and dr.value == c['dead_rew']
        hit = [i for i in c['contact_items'] if i >= 0]
        multi += len(hit) != len(set(hit))
        assert abs(c['food_rew']) <= len(hit)
    assert multi > 5  # the fixture really contains items touched by several contact points


def test_random_constructor_arguments_gather_step():
    """`random_config.json`: the task half of the reference's step() (ant_gather_env.py:76-119, gather_base.py:74-109) with every constructor
    argument drawn at random -- item counts up to 64, 1..64 bins, use_sensor on / off, robot_coll_dist of either sign (distance or contact
    pickup), respawn on / off, world size, spacing, sensor span / range, dying cost: the branches the other fixtures take one at a time,
    combined.  (The combination contact pickup + positions in the observation is where the device once disagreed with the oracle; the
    reference's own output says the oracle was right: the positions are those BEFORE the move.)"""
    combos = set()
    moved_and_shown = 0
    for c in load('random_config')['gather_step']:
        ant = c['cls'] == 'AntGatherBulletEnv'
        cfg = orc.default_config(K.HRL_ANT_GATHER if ant else K.HRL_POINT_GATHER, n_food=c['n_food'], n_poison=c['n_poison'], n_bins=c['n_bins'],
                                 use_sensor=int(c['use_sensor']), robot_coll_dist=c['coll_dist'], respawn=int(c['respawn']),
                                 world_size=tuple(c['world']), robot_object_spacing=c['spacing'], sensor_span=c['span'], sensor_range=c['range'],
                                 dying_cost=c['dying_cost'])
        st = arr(c['state_in']); items = arr(c['items_before']).copy()
        draws = arr(c['respawn_draws']).reshape(-1, 2) if c['respawn_draws'] else np.zeros((1, 2))
        ci = np.asarray(c['contact_items'], np.int32) if c['contact_items'] else np.zeros(1, np.int32)
        assert orc.obs_dim(cfg) == len(c['obs'])
        obs = np.zeros(len(c['obs'])); rew = C.c_double(); done = C.c_int(); fr = C.c_double(); dr = C.c_double()
        head = (C.byref(cfg), int(ant), orc.ptr(st), len(st), orc.ptr(arr(c['torso_xyz'])), C.c_double(c['rpy'][2]), C.c_double(c['initial_z']),
                C.c_double(0.26 if ant else -1.0), orc.ptr(items), orc.ptr(draws), len(c['respawn_draws']))
        tail = (orc.ptr(obs), C.byref(rew), C.byref(done), C.byref(fr), C.byref(dr))
        if c['coll_dist'] > 0:
            used = orc.lib().orc_gather_task_f64(*head, *tail)
        else:
            used = orc.lib().orc_gather_task_contacts_f64(*head, orc.ptr(ci), len(c['contact_items']), *tail)
        assert used == len(c['respawn_draws'])
        np.testing.assert_allclose(items, c['items_after'], atol=1e-12)
        np.testing.assert_allclose(obs, c['obs'], atol=1e-12, equal_nan=True)
        assert rew.value == c['rew'] and bool(done.value) == c['done'] and fr.value == c['food_rew'] and dr.value == c['dead_rew']
        combos.add((c['use_sensor'], c['coll_dist'] > 0, c['respawn']))
        if not c['use_sensor'] and c['coll_dist'] <= 0 and c['items_before'] != c['items_after']:
            moved_and_shown += 1
    assert len(combos) == 8 and moved_and_shown >= 3


def test_random_constructor_arguments_maze_step():
    """`random_config.json`: AntMazeBulletEnv.step (ant_maze_bullet_env.py:63-97) with sense_walls / sense_target / target_encoding /
    done_at_target / max_steps / tol / inner_rew_weight / targ_dist_rew / n_bins / sensor span and range / the target list drawn at random."""
    lines = arr(load('sense_walls')['maze_bounds']).reshape(-1, 4)
    seen = set()
    for c in load('random_config')['maze_step']:
        cfg = orc.default_config(K.HRL_ANT_MAZE, target_encoding=c['encoding'], sense_target=int(c['sense_target']), sense_walls=int(c['sense_walls']),
                                 n_bins=c['n_bins'], sensor_span=c['sensor_span'], sensor_range=c['sensor_range'], done_at_target=int(c['done_at_target']),
                                 max_steps=c['max_steps'], tol=c['tol'], inner_rew_weight=c['inner_rew_weight'], targ_dist_rew=int(c['targ_dist_rew']),
                                 targets=c['targets'])
        assert orc.obs_dim(cfg) == len(c['obs'])
        obs = np.zeros(len(c['obs'])); rew = C.c_double(); done = C.c_int()
        orc.lib().orc_maze_task_f64(C.byref(cfg), orc.ptr(arr(c['ant_obs'])), C.c_double(c['inner_rew']), int(c['inner_done']),
                                    orc.ptr(arr(c['torso_xy'])), C.c_double(c['rpy'][2]), orc.ptr(arr(c['target'])),
                                    C.c_double(c['walk_target_dist']), c['t_before'] + 1, orc.ptr(lines), 7, 3, orc.ptr(obs), C.byref(rew), C.byref(done))
        np.testing.assert_allclose(obs, c['obs'], atol=1e-11)
        assert rew.value == pytest.approx(c['rew'], abs=1e-12) and bool(done.value) == c['done']
        seen.add((c['done'], c['done_at_target'], c['targ_dist_rew'], c['walk_target_dist'] < c['tol']))
    assert len(seen) >= 12


def test_random_constructor_arguments_flagrun_step():
    """`random_config.json`: AntFlagrunBulletEnv.step (ant_flagrun_env.py:162-204) and its observation (:122-130) with tolerance, timeout (0: off),
    switch_flag_on_collision and the wall sensor (bins, span, range, arena of any size) drawn at random; the arena's lines as the oracle derives
    them from `size + 2` (:59-61) equal the reference scene's bounds."""
    seen = set()
    for c in load('random_config')['flagrun_step']:
        cfg = orc.default_config(K.HRL_ANT_FLAGRUN, tol=c['tol'], flag_timeout=c['timeout'], flag_switch_on_collision=int(c['switch']), flag_size=c['size'],
                                 world_size=(c['size'] + 2, c['size'] + 2), use_sensor=int(c['use_sensor']), n_bins=c['n_bins'], sensor_span=c['span'],
                                 sensor_range=c['range'])
        steps = C.c_int(c['steps_before']); rewarded = C.c_int(int(c['rewarded_before'])); left = C.c_int(c['n_goals'])
        rew = C.c_double(); done = C.c_int(); retarget = C.c_int()
        orc.lib().orc_flagrun_task_f64(C.byref(cfg), C.c_double(c['inner_rew']), int(c['inner_done']), C.c_double(c['walk_target_dist']),
                                       C.byref(steps), C.byref(rewarded), C.byref(left), C.byref(rew), C.byref(done), C.byref(retarget))
        assert rew.value == pytest.approx(c['rew'], abs=1e-9) and bool(done.value) == c['done']
        assert steps.value == c['steps_after'] and bool(rewarded.value) == c['rewarded_after'] and left.value == c['goals_left']
        assert bool(retarget.value) == c['retargeted'] == c['state_is_new']
        if c['retargeted']:
            assert c['target_after'] == c['last_goal']
        hx = (c['size'] + 2) / 2
        mine = [[[hx, hx], [-hx, hx]], [[hx, hx], [hx, -hx]], [[-hx, -hx], [-hx, hx]], [[-hx, -hx], [hx, -hx]]]   # orc_impl.h: the four arena lines of the flagrun observation
        assert mine == c['arena_bounds']
        assert len(c['sensor']) == (c['n_bins'] if c['use_sensor'] else 0)
        if c['use_sensor']:
            out = np.zeros(c['n_bins'])
            ln = arr(c['arena_bounds']).reshape(-1, 4)
            orc.lib().orc_sense_walls_f64(c['n_bins'], C.c_double(c['span']), C.c_double(c['range']), orc.ptr(arr(c['pos'])), C.c_double(c['yaw']), orc.ptr(ln), 4,
                                          int(c['span'] == 2 * math.pi), orc.ptr(out))
            np.testing.assert_allclose(out, c['sensor'], rtol=0, atol=1e-12)
        seen.add((c['done'], c['retargeted'], c['timeout'] > 0, c['switch']))
    assert len(seen) >= 8


def test_reset_potential_belongs_to_the_previous_target():
    """Sequence fixture (reference reset()/next_target()/step() run in-tree around a restated upstream bookkeeping, see
    make_golden.py): the potential a reset leaves is the distance to the PREVIOUS target -- from the new pose (flagrun,
    ant_flagrun_env.py:116) or from the robot's home pose (maze, ant_maze_bullet_env.py:111) -- and a retarget inside
    step() leaves the potential of the old target."""
    g = load('reset_potential_seq')
    dt = g['dt']
    pot = orc.lib().orc_reset_potential_f64
    pot.restype = C.c_double
    for name, maze_kind in (('flagrun', 0), ('maze', 1)):
        for events in g[name]:
            prev_target, prev = [1e3, 0.0], None  # upstream's default walk target before the first episode
            for e in events:
                pos = arr(e['pos'][:2])
                if e['op'] == 'reset':
                    v = pot(maze_kind, orc.ptr(arr(prev_target)), orc.ptr(pos), orc.ptr(arr([-2.0, -5.0])), 13, C.c_double(dt))
                    assert v == pytest.approx(e['potential'], abs=1e-9), (name, e)
                else:
                    # step: the new potential is measured against the target in effect BEFORE any retarget of this step
                    assert e['potential'] == pytest.approx(-np.linalg.norm(pos - arr(prev_target)) / dt, abs=1e-9)
                    base = 1.0 + (e['potential'] - prev['potential'])
                    if name == 'flagrun':
                        reached = np.linalg.norm(pos - arr(prev_target)) < 0.5
                        assert e['rew'] == pytest.approx(base + (5000 if reached else 0), abs=1e-6) and e['retargeted'] == bool(reached)
                    else:
                        assert e['rew'] == pytest.approx(base * 1.0, abs=1e-6)  # inner_rew_weight = 1
                prev_target, prev = e['target'], e


def test_configs_beyond_the_default_sizes():
    """tests/golden/big_config.json, generated from the reference like the others: 20 food + 12 poison items with 24 bins (and 40 + 24 with
    64 bins) through get_sensor_readings / get_abs_pos / step / GatherScene, 12 maze targets with 33 bins, AntMazeMj with 16 and 64 bins --
    constructor arguments the reference accepts (ant_gather_env.py:16-29, ant_maze_bullet_env.py:23-25, ant_maze_mj_env.py:50) and ABI <= 5
    rejected."""
    g = load('big_config')
    check_food_sensor(g['food_sensor'], np.float64, 1e-12)
    check_food_sensor(g['food_sensor'], np.float32, 1e-5)
    check_gather_step(g['gather_step'], n_food=20, n_poison=12, n_bins=24)
    check_gather_scene(g['gather_scene'])
    check_maze_step(g['maze_step'], n_bins=33, targets=g['maze_targets'])
    check_maze_mj_step(g['maze_mj_step'])
    assert {len(c['food']) + len(c['poison']) for c in g['food_sensor']} == {32, 64}
    assert any(c['food_rew'] != 0 for c in g['gather_step']) and any(c['done'] for c in g['maze_step'])
