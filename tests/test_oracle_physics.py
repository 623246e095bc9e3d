"""Self-consistency known-answer tests for the oracle's rigid-body half (PARITY UNPINNED vs pybullet: the
reference ships no tests/fixtures for it and pybullet is absent, so the build's own spec is checked against
physics identities instead)."""
import ctypes as C

import numpy as np
import pytest

import orc
from hrl_pybullet_envs_amd import _capi as K

LO = np.deg2rad([-40, 30, -40, -100, -40, -100, -40, 30])
HI = np.deg2rad([40, 100, 40, -30, 40, -30, 40, 100])


def philox_raw(ctr, key):
    c = (C.c_uint32 * 4)(*ctr); k = (C.c_uint32 * 2)(*key); o = (C.c_uint32 * 4)()
    orc.lib().orc_philox4x32_raw(c, k, o)
    return list(o)


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    assert philox_raw([0] * 4, [0] * 2) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert philox_raw([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert philox_raw([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def rand_q(rng):
    q = np.zeros(15)
    q[:3] = rng.uniform(-1, 1, 3)
    qq = rng.normal(size=4)
    q[3:7] = qq / np.linalg.norm(qq)
    q[7:] = rng.uniform(LO, HI)
    return q


def energy_momentum(model, q, u):
    out = np.zeros(8)
    orc.lib().orc_ant_energy_momentum_f64(C.byref(model), orc.ptr(q), orc.ptr(u), orc.ptr(out))
    return out


def mass_matrix_from_T(model, q):
    E = np.eye(14)
    Td = [energy_momentum(model, q, E[i])[0] for i in range(14)]
    M = np.zeros((14, 14))
    for i in range(14):
        for j in range(14):
            M[i, j] = 2 * Td[i] if i == j else energy_momentum(model, q, E[i] + E[j])[0] - Td[i] - Td[j]
    return M


def test_total_mass_and_model():
    cfg = orc.default_config(K.HRL_ANT_GATHER)
    M = mass_matrix_from_T(cfg.model, rand_q(np.random.RandomState(0)))
    # rho=1000: sphere r=.25 (65.45) + 8 short capsules (7.832) + 4 foot capsules (13.518)  
    assert M[3, 3] == pytest.approx(65.4498 + 8 * 7.83156 + 4 * 13.51846, rel=1e-5)
    assert M[3, 3] == pytest.approx(M[4, 4]) and M[3, 3] == pytest.approx(M[5, 5])


def test_aba_impulse_response_is_inverse_mass_matrix():
    """M^-1 assembled from articulated-body impulse responses == inverse of the kinetic-energy Hessian."""
    cfg = orc.default_config(K.HRL_ANT_GATHER)
    rng = np.random.RandomState(3)
    for _ in range(5):
        q = rand_q(rng)
        M = mass_matrix_from_T(cfg.model, q)
        Minv = np.zeros(196)
        orc.lib().orc_ant_minv_f64(C.byref(cfg.model), orc.ptr(q), orc.ptr(Minv))
        Minv = Minv.reshape(14, 14)
        assert np.abs(Minv - Minv.T).max() < 1e-13
        assert np.abs(Minv @ M - np.eye(14)).max() < 1e-11
        assert np.all(np.linalg.eigvalsh(M) > 0)


def _displace(q, d, eps):
    q2 = q.copy()
    if d < 3:
        a = np.zeros(4); a[d] = np.sin(eps / 2); a[3] = np.cos(eps / 2)
        x, y, z, w = q[3:7]
        q2[3] = a[3] * x + a[0] * w + a[1] * z - a[2] * y
        q2[4] = a[3] * y - a[0] * z + a[1] * w + a[2] * x
        q2[5] = a[3] * z + a[0] * y - a[1] * x + a[2] * w
        q2[6] = a[3] * w - a[0] * x - a[1] * y - a[2] * z
    elif d < 6:
        q2[d - 3] += eps
    else:
        q2[7 + d - 6] += eps
    return q2


def test_aba_forward_dynamics_at_rest():
    """qdd(q, 0, tau) == M^-1 (tau - dV/dq): ABA against a numerically differentiated potential."""
    cfg = orc.default_config(K.HRL_ANT_GATHER)
    rng = np.random.RandomState(4)
    for _ in range(3):
        q = rand_q(rng); tau = rng.uniform(-250, 250, 8); u = np.zeros(14); acc = np.zeros(14)
        orc.lib().orc_ant_accel_f64(C.byref(cfg.model), orc.ptr(q), orc.ptr(u), orc.ptr(tau), orc.ptr(acc))
        V = lambda qq: energy_momentum(cfg.model, qq, u)[1]
        G = np.array([-(V(_displace(q, d, 1e-6)) - V(_displace(q, d, -1e-6))) / 2e-6 for d in range(14)])
        G[6:] += tau
        Minv = np.zeros(196)
        orc.lib().orc_ant_minv_f64(C.byref(cfg.model), orc.ptr(q), orc.ptr(Minv))
        assert np.abs(Minv.reshape(14, 14) @ G - acc).max() < 1e-5 * max(1, np.abs(acc).max())


def _free_flight(h, gravity, T=0.2):
    cfg = orc.default_config(K.HRL_ANT_FLAT, model_ground_z=-1000.0, model_limit_margin=-1e9, model_timestep=h,
                             model_max_joint_vel=1e9, model_gravity=gravity, model_linear_damping=0.0, model_angular_damping=0.0)   # conservative dynamics only
    q = np.zeros(15); q[2] = 0.75; q[6] = 1; q[7:] = np.deg2rad([0, 60, 0, -60, 0, -60, 0, 60])
    u = np.zeros(14); u[:3] = [1.0, -2.0, 0.5]; u[3:6] = [0.3, 0.2, 1.0]; u[6:] = [1, -2, 0.5, 1.5, -1, 2, 0.7, -0.3]
    e0 = energy_momentum(cfg.model, q, u)
    n = int(round(T / h))
    orc.lib().orc_ant_substeps_f64(C.byref(cfg), orc.ptr(q), orc.ptr(u), orc.ptr(np.zeros(8)), n, None)
    return e0, energy_momentum(cfg.model, q, u), n * h


def test_energy_and_momentum_conservation_first_order():
    """Velocity-product (Coriolis/gyroscopic) terms: drift of E, P, L is O(h) and halves with h."""
    drift = []
    for h in (1e-3, 5e-4):
        e0, e1, _ = _free_flight(h, 0.0)
        drift.append([abs(e1[0] - e0[0]), np.abs(e1[2:5] - e0[2:5]).max(), np.abs(e1[5:8] - e0[5:8]).max()])
    drift = np.array(drift)
    assert np.all(drift[0] < [0.1, 0.1, 0.1])             # small at h = 1 ms (E0 ~ 200 J, |L| ~ 60)
    assert np.all(np.abs(drift[1] / drift[0] - 0.5) < 0.05)  # first order


def test_free_fall_matches_closed_form():
    e0, e1, T = _free_flight(5e-4, 9.8)
    m = 65.4498 + 8 * 7.83156 + 4 * 13.51846
    assert e1[4] - e0[4] == pytest.approx(-m * 9.8 * T, rel=2e-3)  # dPz = -m g t
    assert abs((e1[0] + e1[1]) - (e0[0] + e0[1])) < 2e-3 * abs(e0[0] + e0[1])


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_standing_ant_contact_and_limits(dtype):
    cfg = orc.default_config(K.HRL_ANT_GATHER, num_envs=8, seed=1)
    env = orc.OracleEnv(cfg, dtype)
    env.reset()
    assert np.allclose(env.qpos[:, 2], 0.75) and np.all(np.abs(env.qpos[:, 7:]) <= 0.1)
    for t in range(150):
        env.step(np.zeros((8, 8)))
    assert np.isfinite(env.state).all()
    # resting: torso well above the ground, joints inside their ranges (ERP push-out), tiny residual velocity of the base
    assert np.all(env.qpos[:, 2] > 0.4) and np.all(env.qpos[:, 2] < 0.8)
    assert np.all(env.qpos[:, 7:] > LO - 0.05) and np.all(env.qpos[:, 7:] < HI + 0.05)
    assert np.abs(env.qvel[:, 2]).max() < 0.05
    rng = np.random.RandomState(0)
    for t in range(300):
        env.step(rng.uniform(-1, 1, (8, 8)))
        assert np.all(env.qpos[:, 7:] > LO - 0.2) and np.all(env.qpos[:, 7:] < HI + 0.2)
    assert np.isfinite(env.state).all() and np.all(env.qpos[:, 2] > 0.2)


def test_config1_plumbing_and_determinism():
    """BASELINE config 1: 1 env, 1000 random-action steps (README.md:29-34): API shape + determinism."""
    outs = []
    for rep in range(2):
        cfg = orc.default_config(K.HRL_ANT_GATHER, num_envs=1, seed=0)
        env = orc.OracleEnv(cfg, np.float32)
        o = env.reset()
        assert o.shape == (1, 46)
        rng = np.random.RandomState(0)
        tot = 0.0
        for t in range(1000):
            o, r, d, info = env.step(rng.uniform(-1, 1, (1, 8)))
            tot += float(r[0])
            assert info[0, 0] + info[0, 1] == r[0]
            if d[0]:
                break
        outs.append((t, tot, env.state.copy(), env.items.copy()))
    assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1]
    assert np.array_equal(outs[0][2], outs[1][2]) and np.array_equal(outs[0][3], outs[1][3])


def test_sharding_invariance():
    """RNG streams are keyed by GLOBAL env id: a shard [4,8) of 8 envs == envs 4..7 of the full batch."""
    full = orc.OracleEnv(orc.default_config(K.HRL_ANT_GATHER, num_envs=8, seed=5, auto_reset=1), np.float32)
    part = orc.OracleEnv(orc.default_config(K.HRL_ANT_GATHER, num_envs=4, seed=5, auto_reset=1, env_id_offset=4), np.float32)
    full.reset(); part.reset()
    assert np.array_equal(full.state[4:], part.state) and np.array_equal(full.items[4:], part.items)
    a = np.random.RandomState(2).uniform(-1, 1, (30, 8, 8))
    for t in range(30):
        full.step(a[t]); part.step(a[t, 4:])
    assert np.array_equal(full.state[4:], part.state) and np.array_equal(full.obs[4:], part.obs)


def test_reset_distribution_and_items():
    cfg = orc.default_config(K.HRL_ANT_GATHER, num_envs=256, seed=11)
    env = orc.OracleEnv(cfg, np.float32); env.reset()
    it = env.items.reshape(256, 16, 2)
    assert np.all(np.abs(it) <= 7.0) and np.all(np.linalg.norm(it, axis=2) >= 2.0 - 1e-6)  # gather_scene.py:52-62
    j = env.qpos[:, 7:]
    assert np.all(np.abs(j) <= 0.1) and abs(j.mean()) < 0.01 and j.std() == pytest.approx(0.2 / np.sqrt(12), rel=0.1)
    mz = orc.OracleEnv(orc.default_config(K.HRL_ANT_MAZE, num_envs=512, seed=3), np.float32); mz.reset()
    assert np.allclose(mz.qpos[:, :3], [-2, -5, 0.25]) and set(np.unique(mz.aux[:, 3])) == {0, 1, 2, 3}
    assert mz.obs.shape == (512, 38)


@pytest.mark.parametrize('kind,od,ad', [(K.HRL_ANT_FLAT, 29, 8), (K.HRL_ANT_MAZE, 38, 8), (K.HRL_POINT_GATHER, 18, 2)])
def test_other_kinds_run(kind, od, ad):
    cfg = orc.default_config(kind, num_envs=16, seed=2, auto_reset=1)
    env = orc.OracleEnv(cfg, np.float32)
    assert env.reset().shape == (16, od)
    rng = np.random.RandomState(1)
    for t in range(200):
        o, r, d, i = env.step(rng.uniform(-1, 1, (16, ad)))
    assert np.isfinite(env.state).all() and o.shape == (16, od)


def test_pointbot_zero_action_is_nan_done():
    """point_bot.py:29 divides by |a|: zero action -> NaN force -> non-finite obs -> done (SURVEY C-11)."""
    env = orc.OracleEnv(orc.default_config(K.HRL_POINT_GATHER, num_envs=2, seed=0), np.float32)
    env.reset()
    a = np.array([[0, 0], [1, 0]], np.float32)
    o, r, d, i = env.step(a)
    assert d[0] == 1 and d[1] == 0 and not np.isfinite(o[0]).all() and np.isfinite(o[1]).all()


def test_pointbot_is_stopped_by_a_cube_under_the_middle_of_a_face():
    """The player cube (half extent 0.35) pushed face-on into an item cube (half extent 0.125) that sits opposite the MIDDLE of
    its +x face: none of the player's 8 corners comes near the cube, it is the cube's own corners against the player's box that
    stop it.  Same for a player turned 30 degrees about z, and a cube beside the path changes nothing."""
    import ctypes as C
    cfg = orc.default_config(K.HRL_POINT_GATHER, num_envs=1, seed=0)
    f = np.array([40.0, 0.0, 0.0])  # 4 m/s^2: would carry it 0.5 m in half a second

    def run(yaw, items, n=240):
        q = np.array([0, 0, 0.35, 0, 0, np.sin(yaw / 2), np.cos(yaw / 2)], np.float64); u = np.zeros(6); info = np.zeros(3, np.int32)
        it = np.asarray(items, np.float64)
        orc.lib().orc_point_substeps_items_f64(C.byref(cfg), orc.ptr(q), orc.ptr(u), orc.ptr(f), n, orc.ptr(it), len(it) // 2, orc.ptr(info))
        return q, u, info

    gap = 0.1
    q, u, info = run(0.0, [0.35 + gap + 0.125, 0.0])
    assert abs(q[0] - gap) < 0.01 and abs(u[3]) < 0.02 and abs(q[1]) < 5e-3 and info[2] >= 6 and info[1] >= 2  # 4 ground corners + the cube's near corners
    assert abs(q[2] - 0.35) < 0.01 and np.abs(q[3:6]).max() < 0.02  # and it neither climbs nor tips
    qf, uf, _ = run(0.0, [0.35 + gap + 0.125, 3.0])  # the same cube out of the way: the player sails on
    assert qf[0] > 0.4 and uf[3] > 1.0
    # turned by 30 degrees the leading edge is a corner line; a cube at 20 degrees off it meets the oblique +x face
    yaw = np.pi / 6
    nx, ny = np.cos(yaw), np.sin(yaw)
    c = np.array([nx, ny]) * (0.35 + gap + 0.125 * (abs(nx) + abs(ny)))  # its nearest corner `gap` in front of the face
    f[:] = [40.0 * nx, 40.0 * ny, 0.0]
    q, u, info = run(yaw, c)
    assert abs(q[0] * nx + q[1] * ny - gap) < 0.015 and abs(u[3] * nx + u[4] * ny) < 0.03 and info[1] >= 1


def test_upright_player_cube_needs_no_edge_edge_test_against_item_cubes():
    """the PointBot's contacts with an item cube are corner-in-box tests both ways (the player's 8 corners against the cube, the cube's 8 against
    the player's oriented box); box-box EDGE-EDGE crossings are not generated.  While the player cube is upright the two footprints are squares of sides
    0.7 (assets/player_cube.xml:8) and 0.25 (assets/food.xml:12): two squares whose sides differ by more than sqrt 2 cannot overlap without a corner of
    one inside the other, so the corner tests are complete.  Checked, not just argued: random yaws and offsets around touching -- whenever the footprints
    overlap by a millimetre or more the collision pass reports a contact with the cube, and whenever they are further apart than the contact distance
    (+ rounding) it reports none."""
    cfg = orc.default_config(K.HRL_POINT_GATHER, num_envs=1, seed=0)
    rng = np.random.RandomState(4)

    def square(c, half, yaw):
        R = np.array([[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]])
        return c + (R @ (np.array([[1, 1], [-1, 1], [-1, -1], [1, -1]]).T * half)).T

    def signed_gap(A, B):
        """separating-axis distance of two convex polygons: > 0 apart (a lower bound of the distance, exact when a face separates), < 0: overlap depth"""
        best = -1e9
        for P, Q in ((A, B), (B, A)):
            for i in range(len(P)):
                e = P[(i + 1) % len(P)] - P[i]
                n = np.array([e[1], -e[0]]) / np.linalg.norm(e)
                best = max(best, (Q @ n).min() - (P @ n).max())
        return best
    overlaps = apart = 0
    for trial in range(3000):
        yaw = rng.uniform(-np.pi, np.pi)
        ang = rng.uniform(-np.pi, np.pi)
        d = rng.uniform(0.30, 0.72)   # centre distance: from deep overlap to clear of each other (0.35 sqrt 2 + 0.125 sqrt 2 = 0.67)
        cube = np.array([d * np.cos(ang), d * np.sin(ang)])
        gap = signed_gap(square(np.zeros(2), 0.35, yaw), square(cube, 0.125, 0.0))
        q = np.array([0, 0, 0.355, 0, 0, np.sin(yaw / 2), np.cos(yaw / 2)])   # resting on the floor: z spans [0.005, 0.705], the cube's [-0.025, 0.225]
        items = np.full((16, 2), 50.0); items[0] = cube
        info = np.zeros(3, np.int32)
        orc.lib().orc_point_substeps_items_f64(C.byref(cfg), orc.ptr(q.copy()), orc.ptr(np.zeros(6)), orc.ptr(np.zeros(3)), 1, orc.ptr(items.reshape(-1).copy()), 16, orc.ptr(info))
        if gap <= -1e-3:
            overlaps += 1
            assert info[1] >= 1, (trial, yaw, cube, gap, info)
        elif gap >= cfg.model.contact_dist + 1e-6:
            apart += 1
            assert info[1] == 0, (trial, yaw, cube, gap, info)
    assert overlaps > 800 and apart > 300, (overlaps, apart)
