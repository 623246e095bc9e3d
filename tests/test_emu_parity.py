"""Kernel logic on the CPU: the product's wave phases (csrc/step_core.h) run by the lock-step host executor
(tests/emu) against the independent scalar oracle.  Same compiler flags (-ffp-contract=off), same libm, and the
algorithm pins every operation order, so the two implementations must agree BIT FOR BIT.  This is what lets the GPU
tests attribute any difference to device arithmetic rather than to kernel logic."""
import ctypes as C

import numpy as np
import pytest

import emu_env
import orc
from hrl_pybullet_envs_amd import _capi as K

KINDS = [K.HRL_ANT_FLAT, K.HRL_ANT_GATHER, K.HRL_ANT_MAZE, K.HRL_POINT_GATHER, K.HRL_ANT_MAZE_MJ, K.HRL_ANT_FLAGRUN]


@pytest.mark.parametrize('kind', KINDS)
def test_reset_bit_exact(kind):
    cfg = orc.default_config(kind, num_envs=64, seed=7)
    o, e = orc.OracleEnv(cfg, np.float32), emu_env.EmuEnv(cfg)
    o.reset(); e.reset()
    assert np.array_equal(o.state, e.state) and np.array_equal(o.items, e.items) and np.array_equal(o.aux, e.aux)
    assert np.array_equal(o.obs, e.obs)
    # masked reset touches only the selected envs
    mask = np.zeros(64, np.uint8); mask[::3] = 1
    s0 = e.state.copy()
    e.reset(mask); o.reset(mask)
    assert np.array_equal(e.state[mask == 0], s0[mask == 0]) and np.array_equal(o.state, e.state)
    assert np.all(e.aux[mask == 1, 2] == 2) and np.all(e.aux[mask == 0, 2] == 1)


@pytest.mark.parametrize('kind', KINDS)
def test_free_running_bit_exact(kind):
    """Both implementations run free (no state copying) for 120 steps with auto-reset and a short time limit."""
    n = 24
    cfg = orc.default_config(kind, num_envs=n, seed=3, auto_reset=1, max_episode_steps=50)
    o, e, er = orc.OracleEnv(cfg, np.float32), emu_env.EmuEnv(cfg), emu_env.EmuEnv(cfg, reverse=True)
    o.reset(); e.reset(); er.reset()
    rng = np.random.RandomState(kind)
    for t in range(120):
        a = rng.uniform(-1, 1, (n, o.ad)).astype(np.float32)
        if kind == K.HRL_POINT_GATHER and t == 7:
            a[0] = 0  # point_bot.py:29 divides by |a| -> NaN -> done -> auto-reset
        o.step(a); e.step(a); er.step(a)
        for name in ('state', 'items', 'aux', 'obs', 'rew', 'done', 'info'):
            x, y, z = getattr(o, name), getattr(e, name), getattr(er, name)
            assert np.array_equal(x, y, equal_nan=True), (t, name)
            assert np.array_equal(y, z, equal_nan=True), (t, name, 'lane-order dependence')
    assert o.aux[:, 2].min() >= 3  # every env went through at least two auto-resets


def test_gather_pickups_happen_and_match():
    """Drive the ants onto food: pickup, respawn draws and rewards must match exactly."""
    n = 32
    cfg = orc.default_config(K.HRL_ANT_GATHER, num_envs=n, seed=11, auto_reset=1)
    o, e = orc.OracleEnv(cfg, np.float32), emu_env.EmuEnv(cfg)
    o.reset(); e.reset()
    rng = np.random.RandomState(5)
    picked = 0
    for t in range(40):
        # teleport every torso next to one of its items (identical in both copies)
        k = rng.randint(0, 16, n)
        xy = o.items.reshape(n, 16, 2)[np.arange(n), k] + rng.uniform(-0.6, 0.6, (n, 2)).astype(np.float32)
        o.state[:, 0:2] = xy; e.state[:, 0:2] = xy
        a = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
        o.step(a); e.step(a)
        picked += int((o.info[:, 0] != 0).sum())
        assert np.array_equal(o.items, e.items) and np.array_equal(o.rew, e.rew) and np.array_equal(o.obs, e.obs)
    assert picked > 50


def test_maze_reaches_targets_and_matches():
    n = 32
    cfg = orc.default_config(K.HRL_ANT_MAZE, num_envs=n, seed=2, auto_reset=1)
    o, e = orc.OracleEnv(cfg, np.float32), emu_env.EmuEnv(cfg)
    o.reset(); e.reset()
    rng = np.random.RandomState(9)
    hits = 0
    for t in range(20):
        tgt = np.array([[2, -3], [2, 0], [2, 3], [-2, 4]], np.float32)[o.aux[:, 3]]
        xy = tgt + rng.uniform(-2.5, 2.5, (n, 2)).astype(np.float32)
        xy[:, 0] = np.clip(xy[:, 0], 1.6, 4.5)  # stay out of the box
        o.state[:, 0:2] = xy; e.state[:, 0:2] = xy
        o.state[:, 2] = 0.6; e.state[:, 2] = 0.6
        a = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
        o.step(a); e.step(a)
        hits += int((o.rew > 0).sum())
        assert np.array_equal(o.rew, e.rew) and np.array_equal(o.done, e.done) and np.array_equal(o.state, e.state)
    assert hits > 20


@pytest.mark.parametrize('kind,n_bins,nf,npo', [(K.HRL_ANT_GATHER, 10, 8, 8), (K.HRL_POINT_GATHER, 5, 8, 8), (K.HRL_ANT_GATHER, 3, 5, 2)])
def test_abs_pos_observation_variant(kind, n_bins, nf, npo):
    """use_sensor=False (ant_gather_env.py:179-196): nearest-first item coordinates instead of the bin sensor."""
    cfg = orc.default_config(kind, num_envs=16, seed=3, auto_reset=1, use_sensor=0, n_bins=n_bins, n_food=nf, n_poison=npo)
    o, e = orc.OracleEnv(cfg, np.float32), emu_env.EmuEnv(cfg)
    assert o.od == (26 if kind == K.HRL_ANT_GATHER else 8) + 2 * (min(nf, n_bins) + min(npo, n_bins))
    o.reset(); e.reset()
    assert np.array_equal(o.obs, e.obs)
    rng = np.random.RandomState(0)
    for t in range(40):
        a = rng.uniform(-1, 1, (16, o.ad)).astype(np.float32)
        o.step(a); e.step(a)
        assert np.array_equal(o.obs, e.obs) and np.array_equal(o.state, e.state) and np.array_equal(o.items, e.items)
    # nearest-first: consecutive squared distances of the reported food items are non-decreasing
    nb = 26 if kind == K.HRL_ANT_GATHER else 8
    mf = min(nf, n_bins)
    fxy = o.obs[:, nb:nb + 2 * mf].reshape(16, mf, 2)
    d2 = ((fxy - o.state[:, None, 0:2]) ** 2).sum(-1)
    assert np.all(np.diff(d2, axis=1) >= -1e-4)


def test_flagrun_goals_rewards_and_exhaustion():
    """Teleport the ants onto their goals: +5000 and a new goal each time, `done` once the 4 goals are used up."""
    import ctypes as C
    n = 16
    cfg = orc.default_config(K.HRL_ANT_FLAGRUN, num_envs=n, seed=6, auto_reset=1, flag_max_targets=4, use_sensor=1)
    o, e = orc.OracleEnv(cfg, np.float32), emu_env.EmuEnv(cfg)
    o.reset(); e.reset()
    rng = np.random.RandomState(3)
    hits = dones = 0
    for t in range(24):
        if t % 2 == 1:
            for i in range(n):
                g = np.zeros(2, np.float32)
                orc.lib().orc_flag_goal_f32(C.byref(cfg), int(o.aux[i, 2]), int(o.aux[i, 3] & 0xffff), orc.ptr(g))
                # walk_target_dist is measured from the parts centroid (13 links + floor + wall at (-6, 0)): invert it
                o.state[i, 0:2] = np.clip([(15 * g[0] + 6) / 13, 15 * g[1] / 13], -5.5, 5.5); o.state[i, 2] = 0.6
            e.state[...] = o.state
        a = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
        o.step(a); e.step(a)
        hits += int((o.rew > 1000).sum()); dones += int(o.done.sum())
        for name in ('state', 'aux', 'obs', 'rew', 'done'):
            assert np.array_equal(getattr(o, name), getattr(e, name)), (t, name)
    assert o.obs.shape == (n, 36) and hits >= 4 * n and dones >= n


def test_flagrun_close_goal_mode():
    """max_target_dist mode (ant_flagrun_env.py:80-89,111-112): goals drawn around the robot, kept in items[0..1], never run out."""
    n = 16
    cfg = orc.default_config(K.HRL_ANT_FLAGRUN, num_envs=n, seed=11, auto_reset=1, flag_max_targets=0, flag_max_target_dist=3.0,
                             flag_timeout=7)
    o, e = orc.OracleEnv(cfg, np.float32), emu_env.EmuEnv(cfg)
    o.reset(); e.reset()
    assert np.array_equal(o.items, e.items) and np.array_equal(o.obs, e.obs)
    g0 = o.items[:, 0:2].copy()
    # first goal: per axis tol <= |offset from the start pose (0, 0)| <= max_target_dist / 2, inside the arena
    assert np.all(np.abs(g0) >= 0.5 - 1e-6) and np.all(np.abs(g0) <= 1.5 + 1e-6) and len({tuple(r) for r in g0.tolist()}) > 1
    rng = np.random.RandomState(5)
    hits = retargets = 0
    for t in range(40):
        if t % 5 == 3:  # teleport onto the goal (walk_target_dist is measured from the parts centroid: 13 links + floor + wall at (-6, 0))
            g = o.items[:, 0:2]
            o.state[:, 0] = np.clip((15 * g[:, 0] + 6) / 13, -4.2, 4.2); o.state[:, 1] = np.clip(15 * g[:, 1] / 13, -4.2, 4.2)
            o.state[:, 2] = 0.75; o.state[:, 3:7] = (0, 0, 0, 1); o.state[:, 7:29] = 0  # a clean drop pose, clear of ground and walls
            e.state[...] = o.state
        before = o.items[:, 0:2].copy()
        a = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
        o.step(a); e.step(a)
        for name in ('state', 'items', 'aux', 'obs', 'rew', 'done'):
            assert np.array_equal(getattr(o, name), getattr(e, name)), (t, name)
        moved = np.any(o.items[:, 0:2] != before, axis=1) & ~o.done.astype(bool)
        retargets += int(moved.sum()); hits += int((o.rew > 1000).sum())
        # a fresh goal lies within max_target_dist / 2 per axis of the robot and strictly inside the arena
        d = np.abs(o.items[moved, 0:2] - o.state[moved, 0:2])
        assert np.all(d <= 1.5 + 1e-5) and np.all(d >= 0.5 - 1e-5) and np.all(np.abs(o.items[moved, 0:2]) < 5.0)
    assert hits >= n and retargets >= 4 * n  # reached goals + 7-step timeouts; the episode never ends for lack of goals
    bad = orc.default_config(K.HRL_ANT_FLAGRUN, flag_max_targets=5, flag_max_target_dist=2.0)
    import ctypes as C
    assert b'exactly one' in emu_env.lib().emu_validate(C.byref(bad))


def test_validation_errors():
    import ctypes as C
    L = emu_env.lib()
    bad = orc.default_config(K.HRL_ANT_GATHER, n_food=40, n_poison=25)
    assert b'n_food' in L.emu_validate(C.byref(bad))
    assert L.emu_validate(C.byref(orc.default_config(K.HRL_ANT_GATHER, n_food=40, n_poison=24, n_bins=64))) == b''   # 64 items, obs 154
    assert b'n_bins' in L.emu_validate(C.byref(orc.default_config(K.HRL_ANT_GATHER, n_bins=65)))
    assert b'flag_goal_capacity' in L.emu_validate(C.byref(orc.default_config(K.HRL_ANT_FLAGRUN, flag_manual_goals=1, flag_goal_capacity=64)))
    bad = orc.default_config(K.HRL_ANT_GATHER, robot_coll_dist=0, model_item_collision=0)
    assert b'robot_coll_dist' in L.emu_validate(C.byref(bad))  # contact pickup needs the cubes as colliders
    assert L.emu_validate(C.byref(orc.default_config(K.HRL_ANT_GATHER, robot_coll_dist=0))) == b''
    bad = orc.default_config(K.HRL_ANT_MAZE, n_bins=1, sensor_span=3.0)
    assert b'n_bins' in L.emu_validate(C.byref(bad))  # the wall sensor would divide by n_bins - 1
    bad = orc.default_config(K.HRL_ANT_MAZE, sense_walls=0, sense_target=1, sensor_span=0.0)
    assert b'sensor_span' in L.emu_validate(C.byref(bad))
    bad = orc.default_config(K.HRL_ANT_MAZE, n_targets=0)
    assert b'n_targets' in L.emu_validate(C.byref(bad))
    assert L.emu_validate(C.byref(orc.default_config(K.HRL_ANT_MAZE))) == b''
    assert L.emu_lds_bytes() <= 10240  # 16 waves per CU x 10 KB <= 160 KB LDS


def test_product_defaults_equal_oracle_defaults():
    import ctypes as C
    for kind in KINDS:
        a = K.hrl_config(); emu_env.lib().emu_default_config(kind, C.byref(a))
        assert bytes(a) == bytes(orc.default_config(kind))
        assert emu_env.lib().emu_obs_dim(C.byref(a)) == orc.obs_dim(a) == {0: 29, 1: 46, 2: 38, 3: 18, 4: 60, 5: 28}[kind]


@pytest.mark.parametrize('kind,n,kw', [
    (K.HRL_ANT_GATHER, 5, dict(n_bins=7, n_food=5, n_poison=3, sensor_range=9.0, sensor_span=2.0, world_size=(9.0, 11.0))),
    (K.HRL_ANT_GATHER, 4, dict(respawn=0, robot_coll_dist=4.0, dying_cost=-3.0)),
    (K.HRL_ANT_MAZE, 6, dict(sense_target=1, n_bins=8)),
    (K.HRL_ANT_MAZE, 5, dict(target_encoding=1, sense_walls=0, tol=3.0, targ_dist_rew=1, max_steps=20, done_at_target=0)),
    (K.HRL_ANT_MAZE_MJ, 4, dict(inner_rew_weight=0.5, n_bins=6)),
    (K.HRL_ANT_GATHER, 3, dict(model_solver_iters=2, model_frame_skip=2, model_limit_margin=0.1)),
    (K.HRL_ANT_GATHER, 6, dict(model_self_collision=0, model_item_collision=0)),
    # hrl_model of ABI v7, all on at once: Bullet's per-body damping (pybullet's 0.04 and a strong one), restitution, a tight contact cap, joint damping + armature
    (K.HRL_ANT_GATHER, 6, dict(model_linear_damping=0.04, model_angular_damping=0.04, model_restitution=0.3, model_max_contacts=6, model_joint_damping=1.0, model_joint_armature=1.0)),
    (K.HRL_ANT_MAZE, 5, dict(model_linear_damping=3.0, model_angular_damping=8.0, model_restitution_threshold=0.0, model_restitution=0.8)),
    (K.HRL_POINT_GATHER, 6, dict(model_linear_damping=0.04, model_angular_damping=2.0, model_restitution=0.5, model_max_contacts=3)),
    (K.HRL_ANT_GATHER, 6, dict(robot_coll_dist=0.0)),
    (K.HRL_POINT_GATHER, 6, dict(robot_coll_dist=-1.0, respawn=0)),
    (K.HRL_ANT_MAZE, 5, dict(inner_rew_weight=1.0)),
    (K.HRL_ANT_FLAGRUN, 7, dict(flag_max_targets=0, flag_max_target_dist=2.5, flag_timeout=6, flag_size=3.0, world_size=(5.0, 5.0), centroid_static_sum=(-2.5, 0.0))),
    (K.HRL_ANT_FLAGRUN, 6, dict(flag_enclosed=0, centroid_n_static=1, centroid_static_sum=(0.0, 0.0), flag_timeout=8, flag_max_targets=5)),
    (K.HRL_ANT_FLAGRUN, 6, dict(flag_switch_on_collision=0, flag_timeout=7, flag_max_targets=4)),
    # constructor arguments beyond the caps of ABI <= 5 (ant_gather_env.py:16-29 takes any n_food / n_poison / n_bins, ant_maze_bullet_env.py:23
    # any targets): more than 16 items (longer items record, 16-item slices in the packed contact phase, 6-bit item field of the respawn key),
    # observations wider than the wave (several packing / store passes), more than 8 targets
    (K.HRL_ANT_GATHER, 6, dict(n_food=20, n_poison=12, n_bins=24, world_size=(9.0, 9.0))),
    (K.HRL_POINT_GATHER, 6, dict(n_food=20, n_poison=12, n_bins=24, world_size=(9.0, 9.0))),
    (K.HRL_ANT_GATHER, 5, dict(n_food=40, n_poison=24, n_bins=64, robot_coll_dist=0.0, world_size=(8.0, 8.0))),
    (K.HRL_POINT_GATHER, 5, dict(n_food=33, n_poison=31, n_bins=40, robot_coll_dist=-1.0, world_size=(8.0, 8.0))),
    (K.HRL_ANT_GATHER, 5, dict(n_food=20, n_poison=12, n_bins=24, use_sensor=0, world_size=(9.0, 9.0))),
    (K.HRL_ANT_MAZE_MJ, 4, dict(n_bins=16)),
    (K.HRL_ANT_MAZE_MJ, 4, dict(n_bins=64)),
    (K.HRL_ANT_MAZE, 6, dict(sense_target=1, n_bins=33, targets=[(-2.0 + 0.5 * i, -4.0 + 0.1 * i) for i in range(12)], tol=0.7)),
    (K.HRL_ANT_FLAGRUN, 4, dict(use_sensor=1, n_bins=40, flag_timeout=9)),
])
def test_non_default_configs_bit_exact(kind, n, kw):
    cfg = orc.default_config(kind, num_envs=n, seed=17, auto_reset=1, max_episode_steps=25, **kw)
    o, e = orc.OracleEnv(cfg, np.float32), emu_env.EmuEnv(cfg)
    o.reset(); e.reset()
    rng = np.random.RandomState(4)
    for t in range(60):
        a = rng.uniform(-1, 1, (n, o.ad)).astype(np.float32)
        o.step(a); e.step(a)
        for name in ('state', 'items', 'aux', 'obs', 'rew', 'done', 'info', 'final_obs', 'truncated'):
            assert np.array_equal(getattr(o, name), getattr(e, name), equal_nan=True), (t, name)


def both(cfg, reverse=False):
    return orc.OracleEnv(cfg, np.float32), emu_env.EmuEnv(cfg, reverse=reverse)


def same(o, e, tag=''):
    for name in ('state', 'items', 'aux', 'obs', 'rew', 'done', 'info', 'final_obs', 'truncated'):
        assert np.array_equal(getattr(o, name), getattr(e, name), equal_nan=True), (tag, name)


@pytest.mark.parametrize('kind', [K.HRL_ANT_GATHER, K.HRL_POINT_GATHER])
def test_item_cubes_collide_and_contact_pickup(kind):
    """The food / poison cubes are static boxes (food.xml:12,19): robots parked on top of / next to cubes are pushed by
    them, and with robot_coll_dist <= 0 every contact point with a cube pays +-1 and moves it (ant_gather_env.py:113-116).
    Both lane orders of the wave phases equal the oracle bit for bit."""
    n = 32
    for kw in (dict(), dict(robot_coll_dist=0.0), dict(robot_coll_dist=0.0, use_sensor=0)):
        cfg = orc.default_config(kind, num_envs=n, seed=13, auto_reset=1, **kw)
        (o, e), er = both(cfg), emu_env.EmuEnv(cfg, reverse=True)
        o.reset(); e.reset(); er.reset()
        rng = np.random.RandomState(2)
        touched = paid = 0
        for t in range(30):
            k = rng.randint(0, 16, n)
            off = rng.uniform(-1.0, 1.0, (n, 2)).astype(np.float32) * (1.4 if kind == K.HRL_ANT_GATHER else 0.45)
            if kw:  # contact mode: nothing is picked up by distance, so stand right next to / on the cube
                xy = o.items.reshape(n, 16, 2)[np.arange(n), k] + off
                for env in (o, e, er):
                    env.state[:, 0:2] = xy
                if kind == K.HRL_POINT_GATHER:
                    # a player teleported INTO a cube is thrown out within a substep and touches nothing at the step's last collision
                    # pass: park it at rest, unturned, with a face against the cube (gap -4 .. 12 mm) at any offset along that face
                    side = rng.randint(0, 4, n); d = np.array([[1, 0], [-1, 0], [0, 1], [0, -1]], np.float32)[side]
                    lat = rng.uniform(-0.42, 0.42, n).astype(np.float32); gap = rng.uniform(-0.004, 0.012, n).astype(np.float32)
                    xy = o.items.reshape(n, 16, 2)[np.arange(n), k] - d * (np.float32(0.475) + gap)[:, None] + d[:, ::-1] * lat[:, None]
                    for env in (o, e, er):
                        env.state[:, 0:2] = xy; env.state[:, 2] = 0.35; env.state[:, 3:7] = [0, 0, 0, 1]; env.state[:, 7:13] = 0
            a = rng.uniform(-1, 1, (n, o.ad)).astype(np.float32)
            it0 = o.items.copy()
            o.step(a); e.step(a); er.step(a)
            same(o, e, t); same(e, er, (t, 'lane order'))
            paid += int((o.info[:, 0] != 0).sum()); touched += int(np.any(o.items != it0, axis=1).sum())
            if 'use_sensor' in kw:  # get_food_obs (ant_gather_env.py:95-96) ran before reward_collision (:113-116) moved the cubes: old positions
                live = np.isfinite(o.obs).all(axis=1) & (o.done == 0)
                m, nb = min(8, cfg.n_bins), o.od - 4 * min(8, cfg.n_bins)
                for i in np.nonzero(live)[0]:
                    old = it0[i, :16].reshape(8, 2)
                    assert all(any(np.array_equal(f, q) for q in old) for f in e.obs[i, nb:nb + 2 * m].reshape(m, 2)), (t, i)
        if kw:
            assert paid > 20 and touched > 20, (paid, touched)
    # the cubes matter to the physics: the same rollout without them diverges
    c1 = orc.default_config(kind, num_envs=n, seed=13, robot_coll_dist=0.0)
    c0 = orc.default_config(kind, num_envs=n, seed=13, robot_coll_dist=4.0, respawn=0, model_item_collision=0)
    o1, o0 = orc.OracleEnv(c1, np.float32), orc.OracleEnv(c0, np.float32)
    o1.reset(); o0.reset()
    # ant: the torso over the cube; point bot: one of its bottom corners over the cube (its contact points are the 8 corners)
    xy = o1.items.reshape(n, 16, 2)[:, 3] + np.float32(0.05 if kind == K.HRL_ANT_GATHER else 0.33)
    o1.state[:, 0:2] = xy; o0.state[:, 0:2] = xy; o0.items[...] = o1.items
    if kind == K.HRL_ANT_GATHER:
        o1.state[:, 2] = 0.4; o0.state[:, 2] = 0.4  # torso low enough to sit on the 0.225 m high cube
    a = np.zeros((n, o1.ad), np.float32) + np.float32(0.3)
    o1.step(a); o0.step(a)
    assert np.abs(o1.state[:, :15] - o0.state[:, :15]).max() > 1e-3


def test_pointbot_cube_corners_against_the_turned_player_box_bit_exact():
    """The second half of the player-vs-cube contacts (the cube's corners against the player's oriented box, orc_impl.h
    orc_point_substep): players turned by any yaw and slightly tipped, parked so that cubes sit under the middle of a face, under
    the body or beside an edge; both lane orders of the wave phases equal the oracle bit for bit, and such contacts do occur."""
    import ctypes as C
    n = 48
    cfg = orc.default_config(K.HRL_POINT_GATHER, num_envs=n, seed=21, auto_reset=1, robot_coll_dist=0.0)
    (o, e), er = both(cfg), emu_env.EmuEnv(cfg, reverse=True)
    o.reset(); e.reset(); er.reset()
    rng = np.random.RandomState(6)
    face_only = 0
    for t in range(40):
        if t % 2 == 0:
            k = rng.randint(0, 16, n)
            ang = rng.uniform(-np.pi, np.pi, n); dist = rng.uniform(0.0, 0.62, n)
            xy = o.items.reshape(n, 16, 2)[np.arange(n), k] + (np.stack([np.cos(ang), np.sin(ang)], 1) * dist[:, None]).astype(np.float32)
            yaw = rng.uniform(-np.pi, np.pi, n); tip = rng.uniform(-0.05, 0.05, (n, 2))
            quat = np.stack([tip[:, 0], tip[:, 1], np.sin(yaw / 2), np.cos(yaw / 2)], 1); quat /= np.linalg.norm(quat, axis=1, keepdims=True)
            for env in (o, e, er):
                env.state[:, 0:2] = xy; env.state[:, 3:7] = quat.astype(np.float32)
            # contacts that only the cube's corners can make: the oracle sees contacts with this cube, yet no player corner is near it
            for i in range(0, n, 4):
                q = o.state[i, :7].astype(np.float64); info = np.zeros(3, np.int32)
                c0 = orc.default_config(K.HRL_POINT_GATHER, num_envs=1, seed=0)
                orc.lib().orc_point_substeps_items_f64(C.byref(c0), orc.ptr(q.copy()), orc.ptr(np.zeros(6)), orc.ptr(np.zeros(3)), 1,
                                                      orc.ptr(o.items[i].astype(np.float64)), 16, orc.ptr(info))
                R = np.array([[np.cos(yaw[i]), -np.sin(yaw[i])], [np.sin(yaw[i]), np.cos(yaw[i])]])
                corners = q[:2] + (R @ (np.array([[1, 1], [1, -1], [-1, 1], [-1, -1]]).T * 0.35)).T
                far = np.all(np.abs(corners - o.items.reshape(n, 16, 2)[i, k[i]]).max(1) > 0.125 + 0.06)
                face_only += int(info[1] > 0 and far)
        a = rng.uniform(-1, 1, (n, o.ad)).astype(np.float32)
        o.step(a); e.step(a); er.step(a)
        same(o, e, t); same(e, er, (t, 'lane order'))
    assert face_only > 10, face_only


def test_self_collision_rows_bit_exact_and_keep_legs_apart():
    """Hips forced far beyond their +-40 degree range so that capsules of different legs meet (within the range they
    cannot: step_core.h broad phase): the two-body rows of the wave phases equal the oracle's bit for bit in both lane
    orders, and the contacts push the legs apart again (with self-collision off they stay interpenetrated longer)."""
    import ctypes as C
    n = 48
    cfg = orc.default_config(K.HRL_ANT_FLAT, num_envs=n, seed=5, auto_reset=0)
    (o, e), er = both(cfg), emu_env.EmuEnv(cfg, reverse=True)
    o.reset(); e.reset(); er.reset()
    rng = np.random.RandomState(1)
    seen = 0
    for t in range(25):
        if t % 5 == 0:
            hips = rng.uniform(-1.5, 1.5, (n, 4)).astype(np.float32)
            ank = rng.uniform(-1.8, 1.8, (n, 4)).astype(np.float32)
            for env in (o, e, er):
                env.state[:, 2] = 1.5; env.state[:, 7:15:2] = hips; env.state[:, 8:15:2] = ank; env.state[:, 15:29] = 0
        # how many self contacts does the oracle see in these poses?
        for i in range(0, n, 8):
            q = o.state[i, :15].astype(np.float64); u = np.zeros(14); info = np.zeros(3, np.int32); dbg = np.zeros(13, np.int32)
            orc.lib().orc_ant_substeps_items_f64(C.byref(cfg), orc.ptr(q), orc.ptr(u), orc.ptr(np.zeros(8)), 1, None, 0, orc.ptr(info), orc.ptr(dbg), None)
            seen += int((dbg[1:] >= 64).sum())
        a = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
        o.step(a); e.step(a); er.step(a)
        same(o, e, t); same(e, er, (t, 'lane order'))
    assert seen >= 10, seen  # sampled every 8th env: the rollouts really contain self contacts


def test_self_collision_broad_phase_is_exact():
    """The wave phases skip the 48 capsule pairs while every |hip angle| <= 0.75 rad and |ankle angle| <= 2 rad; the oracle
    always tests them.  On random poses inside those bounds (any torso orientation) the oracle never finds a pair within
    reach; just outside the ankle bound it does (a foot folded back over the torso)."""
    import ctypes as C
    cfg = orc.default_config(K.HRL_ANT_FLAT)
    rng = np.random.RandomState(8)
    for i in range(4000):
        q = np.zeros(15); q[2] = 5.0
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax); ang = rng.uniform(-3.1, 3.1)
        q[3:6], q[6] = ax * np.sin(ang / 2), np.cos(ang / 2)
        q[7::2] = rng.choice([-0.75, 0.75, 0.0], 4) if i % 3 == 0 else rng.uniform(-0.75, 0.75, 4)
        q[8::2] = rng.choice([-2.0, 2.0], 4) if i % 5 == 0 else rng.uniform(-2.0, 2.0, 4)
        info = np.zeros(3, np.int32); dbg = np.zeros(13, np.int32)
        orc.lib().orc_ant_substeps_items_f64(C.byref(cfg), orc.ptr(q), orc.ptr(np.zeros(14)), orc.ptr(np.zeros(8)), 1, None, 0, orc.ptr(info), orc.ptr(dbg), None)
        assert dbg[0] == 0 and info[2] == 0, (i, q[7:], dbg)
    found = 0
    for i in range(300):  # outside the ankle bound a foot can fold back over the torso onto another leg
        q = np.zeros(15); q[2] = 5.0; q[6] = 1
        q[7::2] = rng.uniform(-0.75, 0.75, 4); q[8::2] = rng.uniform(2.2, 3.1, 4) * rng.choice([-1, 1], 4)
        orc.lib().orc_ant_substeps_items_f64(C.byref(cfg), orc.ptr(q), orc.ptr(np.zeros(14)), orc.ptr(np.zeros(8)), 1, None, 0, orc.ptr(info), orc.ptr(dbg), None)
        found += int(dbg[0] > 0)
    assert found > 0


def test_flagrun_manual_goals():
    """manual_goal_creation (ant_flagrun_env.py:27,150-153): reset draws no goal; `env.goals = [...]; env.next_target()` comes
    through hrl_set_goals (emu_set_goals here): as in the reference the list is consumed from its BACK (`goals.pop()`, :116) and
    the episode ends when it runs out; next_target() alone (hrl_next_target) pops one more, IndexError -> ok = 0."""
    import ctypes as C
    n, G = 8, 3
    cfg = orc.default_config(K.HRL_ANT_FLAGRUN, num_envs=n, seed=6, auto_reset=0, flag_manual_goals=1, flag_timeout=0)
    o, e = both(cfg)
    o.reset(); e.reset()
    same(o, e, 'reset')
    assert np.all(o.items[:, 0] == 1000) and np.all(o.items[:, 1:] == 0) and np.all(o.aux[:, 3] == 0)  # upstream default walk target
    goals = np.random.RandomState(0).uniform(-4, 4, (n, G, 2)).astype(np.float32)
    orc.lib().orc_set_goals_batch_f32(C.byref(cfg), orc.ptr(o.state), orc.ptr(o.items), orc.ptr(o.aux), orc.ptr(goals), G, None, orc.ptr(o.obs))
    b = e._bufs()
    assert emu_env.lib().emu_set_goals(C.byref(cfg), C.byref(b), orc.ptr(goals), G, None, 0) == 0
    same(o, e, 'set_goals')
    assert np.array_equal(o.items[:, 0:2], goals[:, G - 1]) and np.all((o.aux[:, 3] & 0xffff) == G - 1)  # the LAST goal first
    P0 = K.HRL_FLAG_PENDING_OFF
    assert np.array_equal(o.items[:, P0:P0 + 2 * (G - 1)], goals[:, :G - 1].reshape(n, -1))                # the rest, in list order
    rng = np.random.RandomState(1)
    visited = np.zeros(n, int); done_at = np.full(n, -1)
    for t in range(12):
        cur = o.items[:, 0:2].copy()
        # walk_target_dist is measured from the parts centroid, which holds 13 robot parts and 2 static bodies
        xy = ((15 * cur - np.array([-6.0, 0.0], np.float32)) / 13).astype(np.float32)
        for env in (o, e):
            env.state[:, 0:2] = xy; env.state[:, 2] = 0.5
        a = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
        o.step(a); e.step(a)
        same(o, e, t)
        for i in range(n):
            if done_at[i] < 0:
                if o.rew[i] > 1000:
                    visited[i] += 1
                if visited[i] <= G - 1 and o.rew[i] > 1000 and not o.done[i]:
                    assert np.array_equal(o.items[i, 0:2], goals[i, G - 1 - visited[i]])  # back to front
                if o.done[i]:
                    done_at[i] = t
    assert np.all(visited >= G) and np.all(done_at >= 0)  # all goals reached, then the episode ends for lack of goals
    # next_target() alone: one more goal assigned as plain data (env.goals = [g]), popped by hrl_next_target; then IndexError
    extra = np.random.RandomState(3).uniform(-4, 4, (n, 2)).astype(np.float32)
    ok_o = np.full(n, 7, np.uint8); ok_e = np.full(n, 7, np.uint8)
    mask = np.ones(n, np.uint8); mask[5] = 0
    for env in (o, e):
        env.items[:, K.HRL_FLAG_PENDING_OFF:K.HRL_FLAG_PENDING_OFF + 2] = extra; env.aux[:, 3] = (env.aux[:, 3] & ~0xffff) | 1
    for rep, want in ((0, 1), (1, 0)):
        orc.lib().orc_next_target_batch_f32(C.byref(cfg), orc.ptr(o.state), orc.ptr(o.items), orc.ptr(o.aux), orc.ptr(mask), orc.ptr(o.obs), orc.ptr(ok_o))
        b = e._bufs()
        assert emu_env.lib().emu_next_target(C.byref(cfg), C.byref(b), orc.ptr(mask), orc.ptr(ok_e), rep) == 0  # (second call: lanes in reverse order)
        same(o, e, ('next_target', rep))
        assert np.array_equal(ok_o, ok_e) and np.all(ok_o[mask == 1] == want) and ok_o[5] == 7
        assert np.array_equal(o.items[mask == 1, 0:2], extra[mask == 1]) and np.all((o.aux[mask == 1, 3] & 0xffff) == 0)
    assert not np.array_equal(o.items[5, 0:2], extra[5])  # the masked-out env kept its target


def test_flagrun_manual_close_targets():
    """manual_goal_creation with max_targets < 1 (ant_flagrun_env.py:113-114): next_target() -- from step() on reaching the goal /
    timing out, or from outside -- draws a goal near the robot whatever env.goals holds; reset() draws nothing (:150-153) and
    the episode never runs out of goals.  hrl_set_goals is refused there (the list would never be read)."""
    import ctypes as C
    n = 8
    cfg = orc.default_config(K.HRL_ANT_FLAGRUN, num_envs=n, seed=9, auto_reset=0, flag_manual_goals=1, flag_max_targets=0,
                             flag_max_target_dist=3.0, flag_timeout=4)
    o, e = both(cfg)
    o.reset(); e.reset()
    same(o, e, 'reset')
    assert np.all(o.items[:, 0] == 1000) and np.all(o.aux[:, 3] == 0)
    ok_o = np.zeros(n, np.uint8); ok_e = np.zeros(n, np.uint8)
    orc.lib().orc_next_target_batch_f32(C.byref(cfg), orc.ptr(o.state), orc.ptr(o.items), orc.ptr(o.aux), None, orc.ptr(o.obs), orc.ptr(ok_o))
    b = e._bufs()
    assert emu_env.lib().emu_next_target(C.byref(cfg), C.byref(b), None, orc.ptr(ok_e), 0) == 0
    same(o, e, 'next_target')
    assert np.all(ok_o == 1) and np.all(ok_e == 1) and np.all((o.aux[:, 3] & 0xffff) == 1)
    d = np.abs(o.items[:, 0:2] - o.state[:, 0:2])
    assert np.all(d >= 0.5 - 1e-6) and np.all(d <= 1.5 + 1e-6) and np.all(np.abs(o.items[:, 0:2]) < 5)   # +-U(tol, mtd / 2) per axis, inside the arena
    goals = np.zeros((n, 2, 2), np.float32)
    assert emu_env.lib().emu_set_goals(C.byref(cfg), C.byref(b), orc.ptr(goals), 2, None, 0) != 0
    rng = np.random.RandomState(2)
    seen = [set() for _ in range(n)]
    for t in range(14):  # the 4-step timeout retargets three times; nobody runs out of goals
        a = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
        o.step(a); e.step(a)
        same(o, e, t)
        for i in range(n):
            seen[i].add(tuple(o.items[i, 0:2]))
    assert not o.done.any() and all(len(sx) >= 3 for sx in seen) and np.all((o.aux[:, 3] & 0xffff) >= 4)


def test_flagrun_open_field_and_no_switch_on_collision():
    """ant_flagrun_env.py:59-64 `enclosed=False` (and no sensor): upstream's stadium scene, no walls -- an ant beyond where the
    arena's walls would stand meets nothing lateral; :183-194 `switch_flag_on_collision=False`: reaching the goal pays the +5000
    once and keeps the goal until the timeout moves it."""
    n = 8
    cfg = orc.default_config(K.HRL_ANT_FLAGRUN, num_envs=n, seed=3, auto_reset=0, flag_enclosed=0, centroid_n_static=1, centroid_static_sum=(0.0, 0.0),
                             flag_switch_on_collision=0, flag_timeout=6, flag_max_targets=3)
    walled = orc.default_config(K.HRL_ANT_FLAGRUN, num_envs=n, seed=3, auto_reset=0, flag_switch_on_collision=0, flag_timeout=6, flag_max_targets=3)
    o, e = both(cfg)
    ow = orc.OracleEnv(walled, np.float32)
    o.reset(); e.reset(); ow.reset()
    same(o, e, 'reset')
    for env in (o, e, ow):   # astride the line x = 6 where the enclosed arena's wall stands (world 12 x 12), feet on the ground
        env.state[:, 0] = 6.0; env.state[:, 2] = 0.3
    rng = np.random.RandomState(0)
    for t in range(5):
        a = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
        o.step(a); e.step(a); ow.step(a)
        same(o, e, ('open', t))
    assert np.isfinite(o.state).all() and np.all(np.abs(o.state[:, 0] - 6.0) < 0.5)   # nothing pushed it away
    assert np.abs(ow.state[:, 0] - o.state[:, 0]).max() > 0.05                         # the walled arena did
    o.reset(); e.reset()
    paid = np.zeros(n, int); goals_seen = [set() for _ in range(n)]
    for t in range(14):
        g = np.zeros((n, 2), np.float32)
        for i in range(n):
            orc.lib().orc_flag_goal_f32(C.byref(cfg), int(o.aux[i, 2]), int(o.aux[i, 3] & 0xffff), orc.ptr(g[i:i + 1]))
            goals_seen[i].add(tuple(g[i]))
        xy = ((14 * g) / 13).astype(np.float32)   # centroid = (13 robot parts + the stadium's floor at the origin) / 14
        for env in (o, e):
            env.state[:, 0:2] = xy; env.state[:, 2] = 0.5
        a = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
        o.step(a); e.step(a)
        same(o, e, ('noswitch', t))
        paid += (o.rew > 1000).astype(int)
    # on the goal every step: paid once per goal, the goal only moves with the 6-step timeout (14 steps -> 3 goals), never `done` for it
    assert np.all(paid == np.array([len(sx) for sx in goals_seen])) and all(len(sx) == 3 for sx in goals_seen), (paid, goals_seen)


def test_reset_potential_is_taken_before_the_target_switch():
    """ant_flagrun_env.py:116 / ant_maze_bullet_env.py:111: the potential a reset leaves behind belongs to the PREVIOUS
    target (first episode: upstream's default walk target (1e3, 0)), so the first step's `progress` carries the jump."""
    dt = np.float32(0.0165 / 4) * 4
    for kind, kw in ((K.HRL_ANT_FLAGRUN, dict()), (K.HRL_ANT_MAZE, dict(inner_rew_weight=1.0)), (K.HRL_ANT_MAZE_MJ, dict(inner_rew_weight=1.0))):
        cfg = orc.default_config(kind, num_envs=6, seed=2, auto_reset=0, **kw)
        o, e = both(cfg)
        o.reset(); e.reset()
        same(o, e, 'reset 1')
        pot1 = o.state[:, 31].copy()
        assert np.all(np.abs(pot1 * dt + 1000) < 8)   # distance to (1e3, 0) from the start area
        a = np.zeros((6, 8), np.float32)
        o.step(a); e.step(a)
        same(o, e, 'step 1')
        assert np.all(o.rew > 5e4)                    # the reference's first-step jump (~ +6e4)
        o.reset(); e.reset()
        same(o, e, 'reset 2')
        assert np.all(np.abs(o.state[:, 31] * dt) < 20)  # second episode: against the previous episode's goal / target


def test_specified_transcendentals_accuracy_and_agreement():
    """sin / cos / atan2 / asin of DESIGN.md 3.7: the product's and the oracle's implementations agree bit for bit, and both are
    within a few 1e-7 of the exact functions (what the fp32 observations inherit)."""
    import ctypes as C
    rng = np.random.RandomState(0)
    n = 400000
    def run(which, a, b=None):
        a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(a if b is None else b, np.float32)
        o1 = np.zeros(len(a), np.float32); o2 = np.zeros(len(a), np.float32)
        orc.lib().orc_spec_math_f32(which, orc.ptr(a), orc.ptr(b), orc.ptr(o1), len(a))
        emu_env.lib().emu_spec_math(which, orc.ptr(a), orc.ptr(b), orc.ptr(o2), len(a))
        assert np.array_equal(o1, o2, equal_nan=True)
        return o1.astype(np.float64)
    x = rng.uniform(-20, 20, n)
    assert np.abs(run(0, x) - np.sin(x.astype(np.float32).astype(np.float64))).max() < 2e-7
    assert np.abs(run(1, x) - np.cos(x.astype(np.float32).astype(np.float64))).max() < 2e-7
    yy = np.concatenate([rng.normal(size=n), 10.0 ** rng.uniform(-6, 6, n) * rng.choice([-1, 1], n), [0, 0, 1, -1, 0.0, 3, -3]]).astype(np.float32)
    xx = np.concatenate([rng.normal(size=n), 10.0 ** rng.uniform(-6, 6, n) * rng.choice([-1, 1], n), [0, 1, 0, 0, -1.0, 3, -3]]).astype(np.float32)
    assert np.abs(run(2, yy, xx) - np.arctan2(yy.astype(np.float64), xx.astype(np.float64))).max() < 3.5e-7
    s = rng.uniform(-0.99999, 0.99999, n).astype(np.float32)
    assert np.abs(run(3, s) - np.arcsin(s.astype(np.float64))).max() < 3e-7
    run(2, np.array([np.nan, 1, np.inf, -np.inf, np.inf], np.float32), np.array([1, np.nan, 1, np.inf, np.inf], np.float32))  # same bits, whatever they are


@pytest.mark.parametrize('auto_reset', [1, 0])
@pytest.mark.parametrize('kind', KINDS)
def test_envs_that_blow_up_match_too(kind, auto_reset):
    """Non-finite and absurd states -- velocities of 1e20 and inf, a NaN coordinate, a torso 1e19 m away, robots inside a wall, joint angles far
    out of range -- go through the same arithmetic in the wave phases and in the oracle: NaN, inf, done, the dying cost and the auto-reset that
    follows come out identical (NaNs compared as equal)."""
    n = 64
    cfg = orc.default_config(kind, num_envs=n, seed=31, auto_reset=auto_reset)
    o, e = both(cfg)
    o.reset(); e.reset()
    rng = np.random.RandomState(9)
    nq = 7 if kind == K.HRL_POINT_GATHER else 15
    for t in range(15):
        a = rng.uniform(-1, 1, (n, o.ad)).astype(np.float32)
        if t % 3 == 0:
            rows = rng.permutation(n)[:48]
            for env in (o, e):
                r2 = np.random.RandomState(100 + t)
                env.state[rows[0:8], 15 + r2.randint(0, 6, 8)] = 1e20
                env.state[rows[8:16], 15 + r2.randint(0, 6, 8)] = np.inf
                env.state[rows[16:24], 15 + r2.randint(0, 6, 8)] = -3e38
                env.state[rows[24:28], 0] = 1e19
                env.state[rows[28:32], 2] = -1e19
                env.state[rows[32:36], r2.randint(0, nq, 4)] = np.nan
                env.state[rows[36:40], 0:2] = [-2.0, 0.0] if kind in (K.HRL_ANT_MAZE, K.HRL_ANT_MAZE_MJ) else [7.6, 7.6]
                if kind != K.HRL_POINT_GATHER:
                    env.state[rows[40:48], 7 + r2.randint(0, 8, 8)] = r2.choice([40.0, -1e6, 3e30], 8)
                if kind in (K.HRL_ANT_GATHER, K.HRL_POINT_GATHER):  # item coordinates too
                    env.items[rows[0:4], r2.randint(0, 32, 4)] = np.nan
                    env.items[rows[4:8], r2.randint(0, 32, 4)] = np.inf
                    env.items[rows[24:26], r2.randint(0, 32, 2)] = 1e30
                env.state[rows[20:24], 15 + r2.randint(0, 6, 4)] = np.nan
                env.state[rows[26:28], 1] = -np.inf
        o.step(a); e.step(a)
        same(o, e, t)


@pytest.mark.parametrize('kind', KINDS)
def test_fuzzed_absurd_values_stay_bit_exact(kind):
    """tests/tools/fuzz_parity.py: NaN, +-inf, 1e20, 3e38, denormals and signed zeros written into positions, velocities, item coordinates and actions
    of running envs; every output of the wave phases equals the oracle's at every step (this found a cube at a NaN place being everywhere for the
    oracle and nowhere for the phases)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('fuzz_parity', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools', 'fuzz_parity.py'))
    fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
    for seed in (0, 1):
        for ar in (0, 1):
            assert fz.run(fz.EmuSide, kind, seed, ar) is None


@pytest.mark.parametrize('kind', KINDS)
def test_terminal_observation_and_truncation_flag(kind):
    """ABI v6: `final_obs` = the observation of the step that ended an episode (what the reference's step() returns there,
    ant_gather_env.py:96,118-119, ant_maze_bullet_env.py:82,97) and `truncated` = ended by the step limit alone (gym TimeLimit:
    info['TimeLimit.truncated'] = not done).  Property, on the oracle: an auto-resetting env with a step limit against a twin without reset
    and without limit stepped from the same pre-step records -- the twin's observation IS the terminal one, its done the env's own.
    Then the wave phases (both lane orders) equal the oracle bit for bit, final_obs / truncated included (same())."""
    n, limit = 24, 9
    kw = dict(flag_timeout=4, flag_max_targets=3) if kind == K.HRL_ANT_FLAGRUN else {}
    cfg = orc.default_config(kind, num_envs=n, seed=5, auto_reset=1, max_episode_steps=limit, **kw)
    twin = orc.OracleEnv(orc.default_config(kind, num_envs=n, seed=5, auto_reset=0, max_episode_steps=0, **kw), np.float32)
    (o, e), er = both(cfg), emu_env.EmuEnv(cfg, reverse=True)
    o.reset(); e.reset(); er.reset()
    rng = np.random.RandomState(8)
    n_trunc = n_term = n_both = 0
    for t in range(40):
        if t % 4 == 3 or t % 9 == 8:   # some episodes end on their own -- a numerical failure ends any kind's (ant_gather_env.py:101-103,
            rows = rng.permutation(n)[:4]   # gather_base.py:91-93), an ant held under 0.26 m dies --, some of them exactly at the step limit
            for env in (o, e, er):
                env.state[rows[:2], 15] = np.nan
                if kind != K.HRL_POINT_GATHER:
                    env.state[rows[2:], 2] = 0.05; env.state[rows[2:], 17] = -3.0
        twin.state[...] = o.state; twin.items[...] = o.items; twin.aux[...] = o.aux
        a = rng.uniform(-1, 1, (n, o.ad)).astype(np.float32)
        keep = o.final_obs.copy()
        o.step(a); e.step(a); er.step(a); twin.step(a)
        same(o, e, t); same(e, er, (t, 'lane order'))
        d = o.done.astype(bool)
        hit_limit = twin.aux[:, 0] >= limit
        assert np.array_equal(d, twin.done.astype(bool) | hit_limit)
        assert np.array_equal(o.truncated.astype(bool), hit_limit & ~twin.done.astype(bool))
        assert np.array_equal(o.final_obs[d], twin.obs[d], equal_nan=True)         # the terminal observation
        assert np.array_equal(o.final_obs[~d], keep[~d], equal_nan=True)            # rows of live envs are left alone
        assert np.array_equal(o.rew, twin.rew, equal_nan=True)
        assert not np.array_equal(o.final_obs[d], o.obs[d], equal_nan=True) or not d.any()   # obs itself is already the next episode's first
        n_trunc += int(o.truncated.sum()); n_term += int((d & ~o.truncated.astype(bool)).sum()); n_both += int((hit_limit & twin.done.astype(bool)).sum())
    assert n_trunc >= 20 and n_term >= 5 and n_both >= 1, (n_trunc, n_term, n_both)   # n_both: ended on its own AT the limit -> not truncated


@pytest.mark.parametrize('kind,kw', [
    (K.HRL_ANT_GATHER, dict(n_food=20, n_poison=12, n_bins=24)),
    (K.HRL_POINT_GATHER, dict(n_food=20, n_poison=12, n_bins=24)),
    (K.HRL_ANT_GATHER, dict(n_food=40, n_poison=24, n_bins=64, robot_coll_dist=0.0)),
    (K.HRL_POINT_GATHER, dict(n_food=40, n_poison=24, n_bins=30, robot_coll_dist=-1.0)),
])
def test_more_than_16_items(kind, kw):
    """n_food + n_poison > 16 (ant_gather_env.py:16-17 takes any counts): the longer items record, the 16-item slices of the packed
    contact phase, the wider item field of the respawn key and the item codes beyond the capsule pairs.  Robots are parked at every slot in
    turn so that the items past the 16th (and past the 48th) are picked up, touched and sensed; both lane orders equal the oracle."""
    n = 32
    n_items = kw['n_food'] + kw['n_poison']
    cfg = orc.default_config(kind, num_envs=n, seed=19, auto_reset=1, **kw)
    assert orc.items_stride(cfg) == (64 if n_items == 32 else 128)
    (o, e), er = both(cfg), emu_env.EmuEnv(cfg, reverse=True)
    o.reset(); e.reset(); er.reset()
    same(o, e, 'reset')
    assert np.all(o.items[:, 2 * n_items:] == 0) and np.all(np.abs(o.items[:, :2 * n_items]) <= 7.0) and np.all(o.items[:, 2 * n_items - 2:2 * n_items] != 0)
    contact = 'robot_coll_dist' in kw
    rng = np.random.RandomState(3)
    paid_hi = moved_hi = 0
    for t in range(36):
        k = (np.arange(n) * 2 + t * 5) % n_items        # every slot gets its turn
        it = o.items[:, :2 * n_items].reshape(n, n_items, 2)[np.arange(n), k]
        if contact and kind == K.HRL_POINT_GATHER:
            side = rng.randint(0, 4, n); d = np.array([[1, 0], [-1, 0], [0, 1], [0, -1]], np.float32)[side]
            lat = rng.uniform(-0.3, 0.3, n).astype(np.float32); gap = rng.uniform(-0.004, 0.01, n).astype(np.float32)
            xy = it - d * (np.float32(0.475) + gap)[:, None] + d[:, ::-1] * lat[:, None]
            for env in (o, e, er):
                env.state[:, 0:2] = xy; env.state[:, 2] = 0.35; env.state[:, 3:7] = [0, 0, 0, 1]; env.state[:, 7:13] = 0
        else:
            off = rng.uniform(-1.0, 1.0, (n, 2)).astype(np.float32) * np.float32(0.5 if not contact else 1.2)
            for env in (o, e, er):
                env.state[:, 0:2] = it + off
        a = rng.uniform(-1, 1, (n, o.ad)).astype(np.float32)
        it0 = o.items.copy()
        o.step(a); e.step(a); er.step(a)
        same(o, e, t); same(e, er, (t, 'lane order'))
        moved = np.any((o.items != it0).reshape(n, -1, 2), axis=2) & ~o.done.astype(bool)[:, None]
        moved_hi += int(moved[:, 16:n_items].sum()) if n_items <= 48 else int(moved[:, 48:n_items].sum())
        paid_hi += int((o.info[:, 0] != 0).sum())
    assert moved_hi >= 10 and paid_hi >= 20, (moved_hi, paid_hi)
    # the sensor sees the items past the 16th: with only those in range, readings are non-zero
    if not contact:
        o2 = orc.OracleEnv(orc.default_config(kind, num_envs=4, seed=1, **kw), np.float32)
        e2 = emu_env.EmuEnv(o2.cfg)
        o2.reset(); e2.reset()
        for env in (o2, e2):
            env.items[:, :32] = 90.0       # the first 16 far out of range
            env.items[:, 32:2 * n_items:2] = env.state[:, 0:1] + 3.0; env.items[:, 33:2 * n_items:2] = env.state[:, 1:2] + np.linspace(-2, 2, n_items - 16, dtype=np.float32)
        a = np.zeros((4, o2.ad), np.float32) + np.float32(0.1)
        o2.step(a); e2.step(a)
        same(o2, e2, 'sensor')
        nb = 26 if kind == K.HRL_ANT_GATHER else 8
        assert (o2.obs[:, nb:] > 0).any(axis=1).all()


def test_manual_goal_lists_longer_than_15():
    """manual_goal_creation with flag_goal_capacity = 40 (`env.goals = [...]` takes any list, ant_flagrun_env.py:45): the pending list lives in a
    longer items record; 40 goals are consumed back to front, one per 3-step timeout."""
    import ctypes as C
    n, G = 6, 40
    cfg = orc.default_config(K.HRL_ANT_FLAGRUN, num_envs=n, seed=2, auto_reset=0, flag_manual_goals=1, flag_goal_capacity=G, flag_timeout=3, max_episode_steps=0)
    assert orc.items_stride(cfg) == 96
    (o, e), er = both(cfg), emu_env.EmuEnv(cfg, reverse=True)
    o.reset(); e.reset(); er.reset()
    goals = np.random.RandomState(0).uniform(-4, 4, (n, G, 2)).astype(np.float32)
    orc.lib().orc_set_goals_batch_f32(C.byref(cfg), orc.ptr(o.state), orc.ptr(o.items), orc.ptr(o.aux), orc.ptr(goals), G, None, orc.ptr(o.obs))
    for env in (e, er):
        b = env._bufs()
        assert emu_env.lib().emu_set_goals(C.byref(cfg), C.byref(b), orc.ptr(goals), G, None, env.reverse) == 0
    same(o, e, 'set_goals'); same(e, er, 'set_goals lane order')
    assert np.array_equal(o.items[:, 0:2], goals[:, G - 1]) and np.all((o.aux[:, 3] & 0xffff) == G - 1)
    assert emu_env.lib().emu_set_goals(C.byref(cfg), C.byref(e._bufs()), orc.ptr(goals), G + 1, None, 0) != 0   # beyond the capacity
    rng = np.random.RandomState(1)
    for t in range(3 * G + 2):
        a = rng.uniform(-0.3, 0.3, (n, 8)).astype(np.float32)
        o.step(a); e.step(a); er.step(a)
        same(o, e, t); same(e, er, (t, 'lane order'))
        left, cur = G - 1 - (t + 1) // 3, o.aux[:, 3] & 0xffff   # one goal per timeout, sooner where a goal happens to be reached
        live = ~o.done.astype(bool)
        assert np.all(cur[live] <= max(left, 0)) and np.array_equal(o.items[live, 0:2], goals[np.arange(n)[live], cur[live]]), t
    assert o.done.all()    # the list ran out (IndexError in the reference, :193-194)


def test_next_target_pops_the_shared_list_of_a_non_manual_env():
    """`env.next_target()` called from outside on a NON-manual env (ant_flagrun_env.py:112-116): `self.goals.pop()` of the list reset() made --
    here: the goal counter of the episode advances --, ok = 0 (IndexError in the reference) once max_targets goals are used up."""
    import ctypes as C
    n = 9
    cfg = orc.default_config(K.HRL_ANT_FLAGRUN, num_envs=n, seed=4, auto_reset=0, flag_max_targets=3)
    o, e = both(cfg)
    o.reset(); e.reset()
    mask = np.ones(n, np.uint8); mask[::4] = 0
    for k in range(4):
        ok_o, ok_e = np.full(n, 7, np.uint8), np.full(n, 7, np.uint8)
        orc.lib().orc_next_target_batch_f32(C.byref(cfg), orc.ptr(o.state), orc.ptr(o.items), orc.ptr(o.aux), orc.ptr(mask), orc.ptr(o.obs), orc.ptr(ok_o))
        assert emu_env.lib().emu_next_target(C.byref(cfg), C.byref(e._bufs()), orc.ptr(mask), orc.ptr(ok_e), 0) == 0
        same(o, e, k)
        assert np.array_equal(ok_o, ok_e) and np.all(ok_o[mask == 0] == 7) and np.all(ok_o[mask == 1] == (1 if k < 2 else 0))
        assert np.all((o.aux[mask == 1, 3] & 0xffff) == min(2 + k, 3)) and np.all((o.aux[mask == 0, 3] & 0xffff) == 1)
    a = np.zeros((n, 8), np.float32)
    o.step(a); e.step(a)
    same(o, e, 'step')


def test_random_legal_configs():
    """tests/tools/fuzz_configs.py, a slice of it: every constructor argument and engine parameter drawn at random (biased to the edges of the capacity
    ranges), teleports / masked resets / manual goals between the steps; the wave phases equal the oracle bit for bit after every step.
    (This fuzzer is what found that with robot_coll_dist <= 0 and use_sensor=0 the observation held the positions of touched items AFTER their
    move; the hand-picked matrix had both branches, never together.)"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools'))
    import fuzz_configs as F
    ended = 0
    for seed in range(5000, 5016):
        for kind in F.KINDS:
            r, e = F.run(F.EmuSide, kind, seed * 16 + kind, 25)
            assert r is None, r
            ended += e
    assert ended > 1000


def _item_boxes(o, i):
    ni = o.cfg.n_food + o.cfg.n_poison
    it = o.items[i, :2 * ni].reshape(ni, 2).astype(np.float64)
    return {(16 + k if k < 48 else 64 + k): (np.r_[it[k] - 0.125, -0.025], np.r_[it[k] + 0.125, 0.225]) for k in range(ni)}


@pytest.mark.parametrize('group', [0, 1])
def test_capsule_mid_sections_against_cubes_and_the_maze_box(group):
    """assets/ant.xml:16-55 capsules against assets/food.xml:12 cubes and the assets/box.xml:12 maze box: ants let down onto cubes with the
    MIDDLE of their feet (contact pickup, ant_gather_env.py:113-116: the touch is paid) and feet laid across the vertical edges of the maze
    box -- contacts no end-point sphere sees; wave phases == oracle bit for bit, both launch shapes."""
    import capsule_cases as cc
    n = 32
    cfg = orc.default_config(K.HRL_ANT_GATHER, num_envs=n, seed=4, auto_reset=1, robot_coll_dist=0.0, model_step_group=group)
    o, e = orc.OracleEnv(cfg, np.float32), emu_env.EmuEnv(cfg)
    o.reset(); e.reset()
    rng = np.random.RandomState(8)
    paid = mid = 0
    for t in range(6):
        cc.cubes_under_the_feet(o, rng)
        mid += cc.count_mid_section_contacts(o, range(0, n, 4), lambda i: _item_boxes(o, i))
        e.state[...] = o.state; e.items[...] = o.items; e.aux[...] = o.aux
        for k in range(2):
            a = (rng.uniform(-1, 1, (n, 8)) * (0.0 if k == 0 else 0.3)).astype(np.float32)
            o.step(a); e.step(a)
            for name in ('state', 'items', 'aux', 'obs', 'rew', 'done', 'info'):
                assert np.array_equal(getattr(o, name), getattr(e, name), equal_nan=True), (t, k, name)
            paid += int((o.info[:, 0] != 0).sum())
    assert mid >= 20 and paid >= 60, (mid, paid)
    cfg = orc.default_config(K.HRL_ANT_MAZE, num_envs=n, seed=4, auto_reset=1, model_step_group=group)
    o, e = orc.OracleEnv(cfg, np.float32), emu_env.EmuEnv(cfg)
    o.reset(); e.reset()
    box = {8: (np.array([-5., -2, 0]), np.array([1., 2, 2]))}
    mid = 0
    for t in range(6):
        cc.foot_across_the_maze_corner(o, rng)
        mid += cc.count_mid_section_contacts(o, range(0, n, 2), lambda i: box)
        e.state[...] = o.state; e.aux[...] = o.aux
        for k in range(2):
            a = (rng.uniform(-1, 1, (n, 8)) * 0.3).astype(np.float32)
            o.step(a); e.step(a)
            for name in ('state', 'aux', 'obs', 'rew', 'done', 'info'):
                assert np.array_equal(getattr(o, name), getattr(e, name), equal_nan=True), (t, k, name)
    assert mid >= 25, mid


def test_feet_contact_flags_show_the_previous_step():
    """Upstream WalkerBaseBulletEnv.step (recalled): `state = robot.calc_state()` comes BEFORE the loop that refreshes
    `robot.feet_contact` from the step's contacts, so obs[22:26] of AntMaze / obs[24:28] of AntFlagrun are the flags of the step before
    (zeros in the first step after a reset: robot_specific_reset); AntGather never refreshes them (ant_gather_env.py:105-111).  The flags of
    a step ride in bits 28..31 of aux[1]."""
    for kind, lo in ((K.HRL_ANT_MAZE, 22), (K.HRL_ANT_FLAGRUN, 24)):
        cfg = orc.default_config(kind, num_envs=6, seed=2, auto_reset=1, max_episode_steps=30)
        o, e = orc.OracleEnv(cfg, np.float32), emu_env.EmuEnv(cfg)
        o.reset(); e.reset()
        assert np.all(o.obs[:, lo:lo + 4] == 0) and np.all((o.aux[:, 1].view(np.uint32) >> 28) == 0)
        seen = 0
        for t in range(70):
            before = (o.aux[:, 1].view(np.uint32) >> 28) & 0xf
            a = np.zeros((6, 8), np.float32)
            o.step(a); e.step(a)
            assert np.array_equal(o.obs, e.obs, equal_nan=True) and np.array_equal(o.aux, e.aux)
            cont = o.done == 0   # an env that ended was reset in place: its row is the next episode's first observation, flags 0
            want = ((before[:, None] >> np.arange(4)) & 1).astype(np.float32)
            assert np.array_equal(o.obs[cont, lo:lo + 4], want[cont]) and np.all(o.obs[~cont, lo:lo + 4] == 0)
            assert np.all(((o.aux[~cont, 1].view(np.uint32) >> 28) & 0xf) == 0)
            assert np.all((o.aux[:, 1] & 0x0fffffff) == t + 1)   # the lifetime count underneath
            seen += int(want.sum())
        assert seen > 200   # standing ants: feet on the floor most of the time
    cfg = orc.default_config(K.HRL_ANT_GATHER, num_envs=4, seed=2, auto_reset=1)
    o = orc.OracleEnv(cfg, np.float32); o.reset()
    for t in range(20):
        o.step(np.zeros((4, 8), np.float32))
    assert np.all(o.obs[:, 22:26] == 0) and np.all(o.aux[:, 1] == 20)


def test_flagrun_path_reward_switched_on_for_a_live_env():
    """`AntFlagrunBulletEnv.path_rew_weight = 0.5` while an env runs (ant_flagrun_env.py:158,174-176): set_target() has kept `_goal_start_pos` and
    `_sq_dist_goal` all along (:98-103), so the very next step pays a finite path reward measured from where the robot stood when it got its
    CURRENT goal.  Every flagrun env keeps both in its items record whatever the weight is -- with the bookkeeping tied to the weight the record
    of a default env was all zeros and the first steps after the switch paid +-inf (x / 0) until the next goal or reset."""
    n = 12
    cfg = orc.default_config(K.HRL_ANT_FLAGRUN, num_envs=n, seed=9, auto_reset=1, flag_timeout=25, max_episode_steps=2000)
    o, e = orc.OracleEnv(cfg, np.float32), emu_env.EmuEnv(cfg)
    o.reset(); e.reset()
    assert np.all(o.items[:, K.HRL_FLAG_SQDIST_OFF] >= 0.25) and np.all(o.items[:, K.HRL_FLAG_START_OFF:K.HRL_FLAG_START_OFF + 2] == 0)   # goals >= 0.5 from the start (:71-78)
    rng = np.random.RandomState(1)
    checked = 0
    for t in range(90):
        if t == 33:   # mid-episode, between two goal switches
            cfg.flag_path_rew_weight = 0.5
        if t == 70:   # off and on again: the start position is the CURRENT goal's, not a stale one
            cfg.flag_path_rew_weight = 0.0
        if t == 75:
            cfg.flag_path_rew_weight = 0.5
        xy0, start, sqd = o.state[:, 0:2].copy(), o.items[:, 2:4].copy(), o.items[:, 4].copy()
        goal = np.zeros((n, 2), np.float32)
        for i in range(n):
            orc.lib().orc_flag_goal_f32(C.byref(cfg), int(o.aux[i, 2]), int(o.aux[i, 3] & 0xffff), orc.ptr(goal[i]))
        a = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
        o.step(a); e.step(a)
        for name in ('state', 'items', 'aux', 'obs', 'rew', 'done', 'info', 'goal'):
            assert np.array_equal(getattr(o, name), getattr(e, name), equal_nan=True), (t, name)
        assert np.all(np.isfinite(o.rew)), (t, o.rew)
        keep = (o.done == 0) & (o.goal[:, 2] == 0) & (o.rew < 1000)   # the rows whose record was not rewritten by this step (no switch, no reset, no goal reward)
        xy = o.state[:, 0:2].astype(np.float64)
        path = ((xy - start) * (goal.astype(np.float64) - start)).sum(1) / sqd
        want = (o.info[:, 0].astype(np.float64) + o.info[:, 1]) + cfg.flag_path_rew_weight * path
        assert np.allclose(o.rew[keep], want[keep], rtol=1e-5, atol=1e-4), (t, o.rew[keep], want[keep])
        if cfg.flag_path_rew_weight:
            checked += int(keep.sum())
            assert not keep.any() or np.abs(path[keep]).max() < 50
    assert checked > 300


def test_straggler_threshold_follows_the_model():
    """DevCfg::hot_rows (host_cfg.h::standing_rows): the solver rows of an ant that STANDS under the config -- its feet's contacts x 3 as far as the
    contact cap lets them in, one limit row per joint at its stop, two where the margin spans the joint's whole range -- is what ant_env_block compares an
    env's rows with (scheduling only).  20 at the defaults, the literal of round 5."""
    L = emu_env.lib()
    for kind in (K.HRL_ANT_FLAT, K.HRL_ANT_GATHER, K.HRL_ANT_MAZE, K.HRL_ANT_MAZE_MJ, K.HRL_ANT_FLAGRUN):
        assert L.emu_hot_rows(C.byref(orc.default_config(kind, num_envs=4))) == 20
    cfg = orc.default_config(K.HRL_ANT_GATHER, num_envs=4, model_max_contacts=2)
    assert L.emu_hot_rows(C.byref(cfg)) == 6 + 8
    cfg = orc.default_config(K.HRL_ANT_GATHER, num_envs=4, model_limit_margin=1.3)   # spans the ankles' 70 degrees, not the hips' 80
    assert L.emu_hot_rows(C.byref(cfg)) == 12 + 4 + 8
    cfg = orc.default_config(K.HRL_ANT_GATHER, num_envs=4, model_limit_margin=1.5)
    assert L.emu_hot_rows(C.byref(cfg)) == 12 + 16
    # ... and that is what an ant standing on its four feet holds AT MOST (all eight joints at their stops): the settled population under random
    # torques reaches it and does not pass it -- an env above it has more than its feet on something
    cfg = orc.default_config(K.HRL_ANT_GATHER, num_envs=128, seed=1, auto_reset=1)
    o = orc.OracleEnv(cfg, np.float32); o.reset()
    rng = np.random.RandomState(0)
    for t in range(200):
        o.step(rng.uniform(-1, 1, (128, 8)).astype(np.float32))
    rows = []
    for t in range(20):
        o.solver_rows[:] = 0
        o.step(rng.uniform(-1, 1, (128, 8)).astype(np.float32))
        rows.append(o.solver_rows / 4.0)
    rows = np.concatenate(rows)
    assert np.percentile(rows, 95) <= 20.0 and np.percentile(rows, 90) >= 19.0 and (rows > 20).mean() < 0.02, np.percentile(rows, [50, 90, 95, 99])


@pytest.mark.parametrize('group', [0, 1])
def test_second_support_points_of_capsules_lying_flat_on_a_box_face(group):
    """A capsule that rests flat on a face of the maze box (assets/box.xml:12) or on the top of an item cube (assets/food.xml:12) gets a SECOND support
    point (Bullet keeps a manifold there; with one point the capsule rocks): feet hanging alongside the box's vertical faces, legs stretched out level
    over cubes -- states full of such contacts, counted; wave phases == oracle bit for bit in both launch shapes, the contact cap included."""
    import capsule_cases as cc
    n = 32
    rng = np.random.RandomState(12)
    for cap in (12, 5):
        cfg = orc.default_config(K.HRL_ANT_MAZE, num_envs=n, seed=4, auto_reset=1, model_step_group=group, model_max_contacts=cap)
        o, e = orc.OracleEnv(cfg, np.float32), emu_env.EmuEnv(cfg)
        o.reset(); e.reset()
        seconds = 0
        for t in range(4):
            cc.feet_flat_against_the_maze_box(o, rng)
            seconds += cc.count_second_points(o, range(n))
            e.state[...] = o.state; e.aux[...] = o.aux
            for k in range(2):
                a = (rng.uniform(-1, 1, (n, 8)) * 0.3).astype(np.float32)
                o.step(a); e.step(a)
                for name in ('state', 'aux', 'obs', 'rew', 'done', 'info', 'solver_rows'):
                    assert np.array_equal(getattr(o, name), getattr(e, name), equal_nan=True), (cap, t, k, name)
        assert seconds >= (60 if cap == 12 else 10), (cap, seconds)
    cfg = orc.default_config(K.HRL_ANT_GATHER, num_envs=n, seed=4, auto_reset=1, robot_coll_dist=0.0, model_step_group=group)
    o, e = orc.OracleEnv(cfg, np.float32), emu_env.EmuEnv(cfg)
    o.reset(); e.reset()
    seconds = paid = 0
    for t in range(4):
        cc.feet_flat_on_cubes(o, rng)
        seconds += cc.count_second_points(o, range(n))
        e.state[...] = o.state; e.items[...] = o.items; e.aux[...] = o.aux
        for k in range(2):
            a = (rng.uniform(-1, 1, (n, 8)) * 0.2).astype(np.float32)
            o.step(a); e.step(a)
            for name in ('state', 'items', 'aux', 'obs', 'rew', 'done', 'info', 'solver_rows'):
                assert np.array_equal(getattr(o, name), getattr(e, name), equal_nan=True), (t, k, name)
            paid += int((o.info[:, 0] != 0).sum())
    assert seconds >= 100 and paid >= 60, (seconds, paid)


def test_observe_shows_the_feet_flags_the_last_step_left():
    """hrl_observe (include/hrl_envs.h): the feet-contact entries of an AntMaze / AntFlagrun observation are what robot.feet_contact holds in the reference at
    that point -- the flags the LAST step left (bits 28..31 of aux[1]), zeros right after a reset; a replay that wants zeros clears the bits with the
    state it writes.  Wave phases == oracle, and both == the bits."""
    for kind, lo in ((K.HRL_ANT_MAZE, 22), (K.HRL_ANT_FLAGRUN, 24)):
        cfg = orc.default_config(kind, num_envs=8, seed=3, auto_reset=0, max_episode_steps=0)
        o, e = orc.OracleEnv(cfg, np.float32), emu_env.EmuEnv(cfg)
        o.reset(); e.reset()
        o.observe(); e.observe()
        assert np.array_equal(o.obs, e.obs) and np.all(o.obs[:, lo:lo + 4] == 0)
        for t in range(30):
            o.step(np.zeros((8, 8), np.float32)); e.step(np.zeros((8, 8), np.float32))
        bits = (o.aux[:, 1].view(np.uint32) >> 28) & 0xf
        want = ((bits[:, None] >> np.arange(4)) & 1).astype(np.float32)
        assert want.sum() >= 8   # standing ants: feet on the floor
        o.obs[...] = -7; e.obs[...] = -7
        o.observe(); e.observe()
        assert np.array_equal(o.obs, e.obs) and np.array_equal(o.obs[:, lo:lo + 4], want)
        o.aux[:, 1] &= 0x0fffffff; e.aux[...] = o.aux   # the replay's way to zeros
        o.observe(); e.observe()
        assert np.array_equal(o.obs, e.obs) and np.all(o.obs[:, lo:lo + 4] == 0)
