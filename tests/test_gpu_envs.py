"""GPU: the reference-shaped env classes (README.md:19-37 usage) and the native library actually being used."""
import numpy as np
import pytest
import torch

import orc
from hrl_pybullet_envs_amd import _capi as K

pytestmark = pytest.mark.gpu


def test_readme_loop_single_env():
    """BASELINE config 1 shape: env.seed(0); reset; random actions until done (README.md:24-34)."""
    import hrl_pybullet_envs_amd as H
    env = H.make('AntGatherBulletEnv-v0')
    env.seed(0)
    ob = env.reset()
    assert isinstance(ob, np.ndarray) and ob.shape == (46,) and ob.dtype == np.float64
    o = orc.OracleEnv(orc.default_config(K.HRL_ANT_GATHER, num_envs=1, seed=0), np.float32)
    o.reset()
    assert np.array_equal(ob.astype(np.float32), o.obs[0])
    rng = np.random.RandomState(0)
    for t in range(300):
        a = rng.uniform(-1, 1, 8)
        ob, rew, done, info = env.step(a)
        o.step(a.astype(np.float32)[None])
        assert isinstance(rew, float) and isinstance(done, bool) and set(info) >= {'food_rew', 'dead_rew'}
        assert rew == float(o.rew[0]) and done == bool(o.done[0]) and np.array_equal(ob.astype(np.float32), o.obs[0])
        if done:
            break
    env.close()


def test_time_limit_truncation_flag():
    import hrl_pybullet_envs_amd as H
    env = H.PointGatherBulletEnv(seed=1)
    env.max_episode_steps = 5
    env.reset()
    for t in range(5):
        ob, rew, done, info = env.step(np.array([1.0, 0.0]))
    assert done and info.get('TimeLimit.truncated') is True and ob.shape == (18,)


def test_step_limit_is_gym_makes_part():
    """In the reference the classes have no step limit; gym.make wraps them in TimeLimit(2000) (__init__.py:15).  Here: make() switches the kernel's
    limit on, a directly constructed single env has none -- so a real gym.make, which wraps the object in gym's own TimeLimit, keeps its truncation
    flag (gym 0.21: `info['TimeLimit.truncated'] = not done` AFTER the env's step: an inner limit firing at the same step would turn it False) --,
    and a limit set on a running env carries the simulation over."""
    import hrl_pybullet_envs_amd as H

    class GymTimeLimit:   # gym/wrappers/time_limit.py of gym 0.21, the version the reference needs (seeding)
        def __init__(self, env, max_episode_steps):
            self.env, self._max, self._elapsed = env, max_episode_steps, None

        def reset(self):
            self._elapsed = 0
            return self.env.reset()

        def step(self, action):
            observation, reward, done, info = self.env.step(action)
            self._elapsed += 1
            if self._elapsed >= self._max:
                info['TimeLimit.truncated'] = not done
                done = True
            return observation, reward, done, info

    acts = np.random.RandomState(0).uniform(-1, 1, (5, 2))
    wrapped = GymTimeLimit(H.PointGatherBulletEnv(seed=1), 5)              # what a real gym.make builds
    ours = H.make('PointGatherBulletEnv-v0', seed=1)                        # what make() builds
    assert wrapped.env.max_episode_steps == 0 and ours.max_episode_steps == 2000
    ours.max_episode_steps = 5
    wrapped.reset(); ours.reset()
    for t in range(5):
        ow, rw, dw, iw = wrapped.step(acts[t]); oo, ro, do, io = ours.step(acts[t])
        assert np.array_equal(ow, oo) and rw == ro and dw == do and iw.get('TimeLimit.truncated') == io.get('TimeLimit.truncated'), t
    assert dw is True and iw['TimeLimit.truncated'] is True
    # the limit switched on in the middle of an episode: the same episode goes on
    late = H.PointGatherBulletEnv(seed=1)
    late.reset()
    for t in range(3):
        late.step(acts[t])
    late.max_episode_steps = 5
    for t in range(3, 5):
        ol, rl, dl, il = late.step(acts[t])
    assert np.array_equal(ol, oo) and rl == ro and dl is True and il['TimeLimit.truncated'] is True
    for e in (wrapped.env, ours, late):
        e.close()


def test_batched_classes_and_maze_flat():
    import hrl_pybullet_envs_amd as H
    for cls, od, ad in ((H.AntMazeBulletEnv, 38, 8), (H.AntMjEnv, 29, 8), (H.PointGatherBulletEnv, 18, 2), (H.AntGatherBulletEnv, 46, 8)):
        env = cls(num_envs=256, seed=3)
        ob = env.reset()
        assert ob.shape == (256, od) and ob.is_cuda
        for t in range(20):
            ob, rew, done, info = env.step(torch.rand(256, ad, device='cuda') * 2 - 1)
        assert rew.shape == (256,) and done.dtype == torch.uint8 and bool(torch.isfinite(ob).all())
        if cls is H.AntMjEnv:   # MjAnt.py:10-28: env.robot is the MjAnt, its calc_state() the qpos | qvel the observation is made of
            from hrl_pybullet_envs_amd.envs.MjAnt import MjAnt
            assert isinstance(env.robot, MjAnt) and env.robot.power == 2.5
            live = ~done.bool().cpu().numpy()   # (an env that ended shows its next episode's first observation)
            assert np.array_equal(env.robot.calc_state()[live].astype(np.float32), ob.cpu().numpy()[live])
            assert np.array_equal(env.robot.alive_bonus(env.robot.body_real_xyz[:, 2], 0.0) < 0, env.robot.body_real_xyz[:, 2] - 0.75 <= 0.26)
        env.close()


def test_native_library_is_loaded_not_a_fallback():
    from hrl_pybullet_envs_amd import _lib
    assert _lib.lib().hrl_backend() == b'hip-gfx950'
    maps = open('/proc/self/maps').read()
    assert 'libhrl_envs_hip.so' in maps


def test_c_abi_error_paths():
    """The boundary fails loudly and names the cause: null buffers, a missing items record, bad configs."""
    import ctypes as C
    from hrl_pybullet_envs_amd import _lib
    L = _lib.lib()
    n = 8
    for kind, kw in ((K.HRL_ANT_GATHER, {}), (K.HRL_ANT_FLAGRUN, dict(flag_max_targets=0, flag_max_target_dist=3.0))):
        cfg = _lib.default_config(kind, num_envs=n, seed=0, **kw)
        h = C.c_void_p()
        assert L.hrl_create(C.byref(cfg), C.byref(h)) == K.HRL_OK
        od, ad = L.hrl_obs_dim(C.byref(cfg)), L.hrl_act_dim(C.byref(cfg))
        st = torch.zeros(n, 32, device='cuda'); it = torch.zeros(n, 32, device='cuda'); aux = torch.zeros(n, 4, dtype=torch.int32, device='cuda')
        obs = torch.zeros(n, od, device='cuda'); act = torch.zeros(n, ad, device='cuda'); rew = torch.zeros(n, device='cuda')
        done = torch.zeros(n, dtype=torch.uint8, device='cuda'); info = torch.zeros(n, 4, device='cuda')
        no_items = K.make_buffers(st.data_ptr(), None, aux.data_ptr(), act.data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr(), info.data_ptr())
        assert L.hrl_reset(h, C.byref(no_items), None, None) == K.HRL_ERR_BAD_ARG and b'items' in L.hrl_last_error()
        assert L.hrl_step(h, C.byref(no_items), None) == K.HRL_ERR_BAD_ARG and b'items' in L.hrl_last_error()
        ok = K.make_buffers(st.data_ptr(), it.data_ptr(), aux.data_ptr(), None, obs.data_ptr(), rew.data_ptr(), done.data_ptr(), info.data_ptr())
        assert L.hrl_reset(h, C.byref(ok), None, None) == K.HRL_OK
        assert L.hrl_step(h, C.byref(ok), None) == K.HRL_ERR_BAD_ARG  # no actions
        torch.cuda.synchronize()
        assert L.hrl_destroy(h) == K.HRL_OK
    bad = _lib.default_config(K.HRL_ANT_FLAGRUN, flag_max_targets=5, flag_max_target_dist=2.0)
    h = C.c_void_p()
    assert L.hrl_create(C.byref(bad), C.byref(h)) == K.HRL_ERR_BAD_ARG and b'exactly one' in L.hrl_last_error()
    # a buffer record that was never initialised (ABI v7: hrl_buffers.struct_size): refused at every entry point, before any launch
    cfg = _lib.default_config(K.HRL_ANT_FLAT, num_envs=n, seed=0)
    assert L.hrl_create(C.byref(cfg), C.byref(h)) == K.HRL_OK
    junk = K.hrl_buffers()
    C.memset(C.byref(junk), 0x5a, C.sizeof(junk))
    zero = K.hrl_buffers(); zero.struct_size = 0
    for rec in (junk, zero):
        assert L.hrl_reset(h, C.byref(rec), None, None) == K.HRL_ERR_BAD_ARG and b'hrl_buffers_init' in L.hrl_last_error()
        assert L.hrl_step(h, C.byref(rec), None) == K.HRL_ERR_BAD_ARG and b'struct_size' in L.hrl_last_error()
        assert L.hrl_observe(h, C.byref(rec), None, None) == K.HRL_ERR_BAD_ARG
        assert L.hrl_get_state(h, C.byref(rec), st.data_ptr(), st.data_ptr(), None) == K.HRL_ERR_BAD_ARG
    fresh = K.hrl_buffers()
    C.memset(C.byref(fresh), 0x5a, C.sizeof(fresh))
    assert L.hrl_buffers_init(C.byref(fresh)) == K.HRL_OK and fresh.struct_size == C.sizeof(K.hrl_buffers) and fresh.goal is None and fresh.state is None
    # a caller compiled against a shorter record of this ABI version (no `goal` field): accepted, what lies beyond its struct_size is not read
    od = L.hrl_obs_dim(C.byref(cfg))
    obs = torch.zeros(n, od, device='cuda'); act = torch.zeros(n, 8, device='cuda')
    short = K.make_buffers(st.data_ptr(), None, aux.data_ptr(), act.data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr(), info.data_ptr())
    short.goal = 0x5a5a5a5a5a5a          # garbage beyond the declared size
    short.struct_size = K.hrl_buffers.goal.offset
    assert L.hrl_reset(h, C.byref(short), None, None) == K.HRL_OK and L.hrl_step(h, C.byref(short), None) == K.HRL_OK
    torch.cuda.synchronize()
    assert L.hrl_destroy(h) == K.HRL_OK


def test_update_config_changes_a_live_env_in_place():
    """hrl_update_config: the step limit, reward parameters and engine parameters of a RUNNING env -- `env.max_episode_steps = n`, the flagrun
    class weights -- change with the next launch; the handle, every tensor handed out and the pinned host buffers of step_host() stay the same
    objects; what the buffers' shapes depend on is refused.  Afterwards the env equals an oracle that was given the new config at that step."""
    import ctypes as C
    import hrl_pybullet_envs_amd as H
    from hrl_pybullet_envs_amd import _lib
    n = 64
    env = H.AntGatherBulletEnv(num_envs=n, seed=3)
    o = orc.OracleEnv(orc.default_config(K.HRL_ANT_GATHER, num_envs=n, seed=3, auto_reset=1, max_episode_steps=2000), np.float32)
    ob = env.reset(); o.reset()
    handle, state_ptr = env._backend()._h.value, env._backend().state.data_ptr()
    rng = np.random.RandomState(1)
    for t in range(30):
        if t == 10:
            env.max_episode_steps = 15          # takes effect with the next step: every env is truncated at its 15th
            o.cfg.max_episode_steps = 15
        if t == 20:
            cfg = env._cfg
            cfg.dying_cost = -3.0; cfg.model.solver_iters = 3; cfg.model.contact_erp = 0.5
            env._backend().update_config(cfg)
            o.cfg.dying_cost = -3.0; o.cfg.model.solver_iters = 3; o.cfg.model.contact_erp = 0.5
        a = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
        ob, r, d, info = env.step(torch.from_numpy(a).cuda()); o.step(a)
        assert np.array_equal(env._backend().state.cpu().numpy(), o.state) and np.array_equal(r.cpu().numpy(), o.rew) and np.array_equal(d.cpu().numpy(), o.done), t
        if t == 14:
            assert d.all() and info['TimeLimit.truncated'].all()
    assert env._backend()._h.value == handle and env._backend().state.data_ptr() == state_ptr
    bad = env._cfg.copy(); bad.n_bins = 11      # another observation width
    assert _lib.lib().hrl_update_config(env._backend()._h, C.byref(bad), None) == K.HRL_ERR_BAD_ARG and b'cannot change' in _lib.lib().hrl_last_error()
    env.close()
    one = H.AntGatherBulletEnv(seed=3)          # one env: the pinned host buffers of step_host() survive the change
    one.reset(); one.step(np.zeros(8))
    host = one._backend()._host['obs'].data_ptr()
    one.max_episode_steps = 3
    dn = [one.step(np.zeros(8))[2] for _ in range(2)]
    assert dn == [False, True] and one._backend()._host['obs'].data_ptr() == host
    one.close()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs in one process')
def test_c_abi_refuses_a_launch_from_another_device():
    """A handle belongs to the device that was current at hrl_create(): called with another device current, every entry point returns
    HRL_ERR_BAD_ARG and says which device to set (a launch there would hand the kernel a constants pointer of another GPU)."""
    import ctypes as C
    from hrl_pybullet_envs_amd import _lib
    from hrl_pybullet_envs_amd.vec_env import BatchedEnv
    g = BatchedEnv(_lib.default_config(K.HRL_ANT_FLAT, num_envs=8, seed=0), 'cuda:0')
    g.reset()
    with torch.cuda.device(1):
        assert _lib.lib().hrl_reset(g._h, C.byref(g._bufs), None, None) == K.HRL_ERR_BAD_ARG
        assert b'hipSetDevice(0)' in _lib.lib().hrl_last_error()
    g.step(torch.zeros(8, 8, device='cuda:0'))   # BatchedEnv switches to its device itself
    g.close()


def test_flagrun_info_target_and_class_level_reward_weights():
    """ant_flagrun_env.py:191,199: `info['target'] = self.goal` on exactly the steps in which next_target() ran; :157-178: the class-level reward
    weights, read off the class when the env is built.  One env (the reference's object) and a batch; rewards against the oracle configured alike."""
    import hrl_pybullet_envs_amd as H
    import orc
    env = H.AntFlagrunBulletEnv(timeout=7, num_envs=1, seed=4)
    env.reset()
    g0, switched = env.goal, []
    for t in range(30):
        ob, r, d, info = env.step(np.zeros(8))
        if 'target' in info:
            switched.append(t)
            assert info['target'] == pytest.approx(env.goal) and info['target'] != pytest.approx(g0)
            g0 = env.goal
        else:
            assert env.goal == pytest.approx(g0)
    assert len(switched) >= 4 and switched[0] <= 6 and all(b - a <= 7 for a, b in zip(switched, switched[1:]))   # the 7-step timeout at the latest
    env.close()
    cls = H.AntFlagrunBulletEnv
    from hrl_pybullet_envs_amd.envs.upstream import WalkerBaseBulletEnv
    saved = (cls.ant_env_rew_weight, cls.path_rew_weight, cls.dist_rew_weight, cls.goal_reach_rew)
    saved_up = (WalkerBaseBulletEnv.electricity_cost, WalkerBaseBulletEnv.stall_torque_cost, WalkerBaseBulletEnv.joints_at_limit_cost)
    try:
        _flagrun_weights_body(H, cls, WalkerBaseBulletEnv)
    finally:
        cls.ant_env_rew_weight, cls.path_rew_weight, cls.dist_rew_weight, cls.goal_reach_rew = saved
        WalkerBaseBulletEnv.electricity_cost, WalkerBaseBulletEnv.stall_torque_cost, WalkerBaseBulletEnv.joints_at_limit_cost = saved_up


def _flagrun_weights_body(H, cls, WalkerBaseBulletEnv):
    import orc
    WalkerBaseBulletEnv.electricity_cost, WalkerBaseBulletEnv.stall_torque_cost, WalkerBaseBulletEnv.joints_at_limit_cost = -2.0, -0.1, -0.1   # (the env above zeroed them)
    cls.ant_env_rew_weight, cls.path_rew_weight, cls.dist_rew_weight, cls.goal_reach_rew = 0.5, 2.0, 0.25, 77
    n = 128
    env = H.AntFlagrunBulletEnv(timeout=9, tolerance=1.0, num_envs=n, seed=6)
    assert env.reward_weights == dict(ant_env_rew_weight=0.5, path_rew_weight=2.0, dist_rew_weight=0.25, goal_reach_rew=77.0)
    ocfg = orc.default_config(K.HRL_ANT_FLAGRUN, num_envs=n, seed=6, auto_reset=1, flag_timeout=9, tol=1.0, max_episode_steps=2000,
                              flag_ant_env_rew_weight=0.5, flag_path_rew_weight=2.0, flag_dist_rew_weight=0.25, flag_goal_reach_rew=77.0)
    o = orc.OracleEnv(ocfg, np.float32)
    assert env._cfg.walker_electricity_cost == -2.0                     # upstream's class attribute as it was when the env was built ...
    ob = env.reset(); o.reset()
    assert WalkerBaseBulletEnv.electricity_cost == 0 and WalkerBaseBulletEnv.joints_at_limit_cost == 0   # ... and reset() zeroes it ON THE CLASS (ant_flagrun_env.py:133-135)
    assert bytes(ocfg) == bytes(env._cfg)
    assert np.array_equal(ob.cpu().numpy(), o.obs) and np.array_equal(env._backend().items.cpu().numpy(), o.items)
    rng = np.random.RandomState(0)
    n_sw = 0
    for t in range(40):
        if t == 20:   # some ants put next to their goals: the goal reward (77) and a switch on reaching it
            gl = env.goal
            o.state[::2, 0:2] = gl[::2].astype(np.float32) + np.float32(0.2)
            env._backend().state.copy_(torch.from_numpy(o.state))
        a = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
        ob, r, d, info = env.step(torch.from_numpy(a).cuda()); o.step(a)
        assert np.array_equal(r.cpu().numpy(), o.rew, equal_nan=True) and np.array_equal(ob.cpu().numpy(), o.obs, equal_nan=True), t
        assert np.array_equal(env._backend().goal.cpu().numpy(), o.goal) and np.array_equal(env._backend().items.cpu().numpy(), o.items)
        sw = info['retargeted'].cpu().numpy() != 0
        n_sw += int(sw.sum())
        assert np.allclose(info['target'].cpu().numpy()[~o.done.astype(bool)], env.goal[~o.done.astype(bool)], atol=1e-6)
        if t == 20:
            assert (o.rew[::2] > 60).sum() > 40   # reached: + goal_reach_rew
    assert n_sw > 4 * n   # every 9 steps at the latest
    from hrl_pybullet_envs_amd.adapters import GymnasiumVectorEnv, SB3VecEnv
    seen = 0
    for wrap in (SB3VecEnv(env), GymnasiumVectorEnv(env)):   # the vector-env views hand the switch over per env
        for t in range(10):
            a = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
            o.step(a)
            if isinstance(wrap, SB3VecEnv):
                _, _, _, infos = wrap.step(a)
                sw = o.goal[:, 2] != 0
                assert [('target' in i) for i in infos] == sw.tolist()
                assert all(infos[i]['target'] == pytest.approx(tuple(o.goal[i, :2])) for i in np.nonzero(sw)[0])
                seen += int(sw.sum())
            else:
                _, _, _, _, info = wrap.step(torch.from_numpy(a).cuda())
                assert np.array_equal(info['_target'].cpu().numpy(), o.goal[:, 2] != 0) and np.array_equal(info['target'].cpu().numpy(), o.goal[:, :2])
    assert seen > n // 2
    assert np.all(np.isfinite(env._sq_dist_goal.cpu().numpy())) and env._goal_start_pos.shape == (n, 2)
    env.set_reward_weights(path_rew_weight=0.0, goal_reach_rew=5000)   # this env alone, in place
    o.cfg.flag_path_rew_weight = 0.0; o.cfg.flag_goal_reach_rew = 5000.0
    a = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
    ob, r, d, info = env.step(torch.from_numpy(a).cuda()); o.step(a)
    assert np.array_equal(r.cpu().numpy(), o.rew, equal_nan=True) and np.array_equal(env._backend().state.cpu().numpy(), o.state, equal_nan=True)
    # the reference reads the CLASS attributes in every step: a change reaches a running env with its next step
    cls.dist_rew_weight = 1.0; WalkerBaseBulletEnv.electricity_cost = -2.0
    o.cfg.flag_dist_rew_weight = 1.0; o.cfg.walker_electricity_cost = -2.0
    a = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
    ob, r, d, info = env.step(torch.from_numpy(a).cuda()); o.step(a)
    assert np.array_equal(r.cpu().numpy(), o.rew, equal_nan=True) and env.reward_weights['dist_rew_weight'] == 1.0 and env.reward_weights['path_rew_weight'] == 0.0
    # ... and an AntMazeBulletEnv of the same process runs with whatever the upstream class holds now (in the reference: 0 after any flagrun reset)
    WalkerBaseBulletEnv.electricity_cost = 0
    mz = H.AntMazeBulletEnv(num_envs=32, seed=2, inner_rew_weight=1.0)
    mo = orc.OracleEnv(orc.default_config(K.HRL_ANT_MAZE, num_envs=32, seed=2, auto_reset=1, inner_rew_weight=1.0, max_episode_steps=2000,
                                          walker_electricity_cost=0.0, walker_stall_torque_cost=0.0, walker_joints_at_limit_cost=0.0), np.float32)
    assert bytes(mo.cfg) == bytes(mz._cfg)
    mz.reset(); mo.reset()
    for t in range(6):
        if t == 3:
            WalkerBaseBulletEnv.electricity_cost, WalkerBaseBulletEnv.stall_torque_cost = -2.0, -0.1     # someone sets upstream's defaults again
            mo.cfg.walker_electricity_cost, mo.cfg.walker_stall_torque_cost = -2.0, -0.1
        a = rng.uniform(-1, 1, (32, 8)).astype(np.float32)
        _, r, _, _ = mz.step(torch.from_numpy(a).cuda()); mo.step(a)
        assert np.array_equal(r.cpu().numpy(), mo.rew), t
    mz.close(); env.close()


def test_flagrun_close_goal_class():
    """AntFlagrunBulletEnv(max_targets=0, max_target_dist=d): goals around the robot (ant_flagrun_env.py:80-89), never out of goals."""
    import hrl_pybullet_envs_amd as H
    env = H.AntFlagrunBulletEnv(max_targets=0, max_target_dist=3, timeout=10, num_envs=64, seed=2)
    ob = env.reset()
    assert ob.shape == (64, 28)
    goals = set()
    for t in range(60):
        ob, rew, done, info = env.step(torch.rand(64, 8, device='cuda') * 2 - 1)
        g = env._env.items[:, 0:2].cpu().numpy(); xy = env._env.state[:, 0:2].cpu().numpy()
        assert np.all(np.abs(g) < 5.0) and bool(torch.isfinite(ob).all())
        goals.add(tuple(np.round(g[0], 4)))
    assert len(goals) >= 4  # the 10-step timeout moved env 0's goal several times
    env.close()


def _bench_line(args, timeout=600):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py')] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_mixed_kind_one_gpu():
    """BASELINE.json configs[4] per GPU through bench.py: one JSON line, both sub-shards stepped on their own streams."""
    out = _bench_line(['--kind', 'mixed', '--envs', '4096', '--steps', '30', '--warmup', '5', '--no-cpu-baseline'])
    assert 'mixed batch' in out['metric'] and out['n_gpus'] == 1 and out['config']['envs_per_gpu'] == 4096
    assert out['value'] > 1e6 and out['roofline']['kernel'] == 'k_step<gather>'
    assert set(out['roofline']['streams_ms_per_step']) == {'gather', 'point'}


def test_bench_starts_two_ranks_from_a_bare_shell():
    """`python bench.py --gpus 2` with no torch.distributed.run around it: the parent spawns the ranks (gloo here: the
    box has one GPU, both ranks share it; on a node the same path runs RCCL), weak scaling doubles the env count, the
    side-stream all-gather of episode returns ran."""
    out = _bench_line(['--gpus', '2', '--backend', 'gloo', '--envs', '512', '--steps', '40', '--warmup', '5',
                       '--gather-every', '10', '--no-cpu-baseline'])
    assert out['n_gpus'] == 2 and out['config']['global_envs'] == 1024 and out['scaling'] == 'weak'
    assert out['config']['returns_gathered_ok'] is True and out['value'] > 0
    # the N > 1 line proves what ran: backend, world size, every rank reported in, the timed region's collectives
    r = out['rccl']
    assert r['backend'] == 'gloo' and r['world_size'] == 2 and r['ranks_seen'] == [0, 1] and len(r['devices']) == 2
    assert r['gather_count'] == 4 and r['gather_every'] == 10 and r['gather_us_max'] > 0 and r['gather_bytes_per_rank'] == 4 * 512
    one = _bench_line(['--envs', '512', '--steps', '30', '--warmup', '5', '--no-cpu-baseline'])
    assert 'rccl' not in one and one['config']['parallelism'] == 'one GPU, one process, no collective'   # nothing claimed that did not run


def test_bench_launched_as_the_driver_launches_n_2():
    """The driver's own command for N = 2, word for word -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1
    --master-port P bench.py --gpus 2 --steps 20 --warmup 5` -- with nothing changed but the backend (gloo: both ranks share this box's one GPU, which
    RCCL refuses): default env count, default settle, default gather interval.  One JSON line from rank 0, weak scaling, the timed region's gather ran."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', str(port),
                        os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '20', '--warmup', '5', '--backend', 'gloo'],
                       env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 20 and out['warmup'] == 5 and out['scaling'] == 'weak' and out['steady_state'] is True
    assert out['config']['global_envs'] == 2 * 4096 and out['config']['returns_gathered_ok'] is True
    assert out["rccl"]["world_size"] == 2 and out["rccl"]["ranks_seen"] == [0, 1] and out["rccl"]["gather_count"] == 1   # a gather every 20 steps, queued mid-interval: one falls into the timed 20
    assert 'cpu_baseline' not in out    # the CPU leg belongs to the N = 1 line
    assert 0 < out['ms_per_step_before_barrier'] <= out['ms_per_step']   # the closing barrier is inside the contract's bracket; the line also says what the steps took without it


def test_bench_four_ranks_of_the_config5_shape():
    """The shape of the driver's N = 8 run of BASELINE.json configs[4], as far as a one-GPU box may go (at most six processes on the card: four
    ranks + this one): `--gpus 4 --kind mixed` over gloo, every rank half AntGather half PointGather on two streams, the ReturnGatherer
    gathering both halves of all four shards.  The line proves what ran: world size, every rank seen, the collectives of the timed region."""
    out = _bench_line(['--gpus', '4', '--backend', 'gloo', '--kind', 'mixed', '--envs', '64', '--steps', '20', '--warmup', '5', '--settle', '20',
                       '--gather-every', '5', '--no-cpu-baseline'])
    assert out['n_gpus'] == 4 and out['config']['global_envs'] == 256 and 'mixed batch' in out['metric']
    r = out['rccl']
    assert r['world_size'] == 4 and r['ranks_seen'] == [0, 1, 2, 3] and len(r['devices']) == 4 and r['gather_count'] == 4 and r['gather_bytes_per_rank'] == 4 * 64
    assert out['config']['returns_gathered_ok'] is True and set(out['solver_rows_per_env_step']) == {'gather', 'point'}


def test_bench_default_line_is_a_steady_state_line():
    """Whatever the driver passes for --warmup, the timed window starts 300 untimed steps after the reset (--settle): `steady_state` is true, and
    the line carries the regime it was measured in -- the mean solver rows per env step (a standing ant: four feet on the ground and a few joints
    near their stops: ~50 rows over four substeps; just after a reset: fewer)."""
    out = _bench_line(['--steps', '20', '--warmup', '5', '--no-cpu-baseline'])   # the driver's flags
    assert out['steady_state'] is True and out['settle_steps'] == 300 and out['warmup'] == 5 and out['steps'] == 20
    rows = out['solver_rows_per_env_step']['gather']
    assert 30 < rows < 120, rows
    cold = _bench_line(['--steps', '20', '--warmup', '5', '--settle', '0', '--no-cpu-baseline'])
    assert cold['steady_state'] is False and cold['solver_rows_per_env_step']['gather'] < rows


def test_bench_watchdog_names_the_stage_a_rank_hangs_in():
    """A rank that stops making progress exits 3 with the stage it hung in (a test hook makes it hang); a long run that keeps going is left alone."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--envs', '512', '--steps', '50', '--warmup', '5', '--no-cpu-baseline',
                        '--watchdog', '6', '--stall-in', 'warmup'], capture_output=True, text=True, timeout=300)
    assert p.returncode == 3 and 'watchdog: rank 0 made no progress in stage "warmup" for 6 s' in p.stderr, (p.returncode, p.stderr[-500:])
    # a stall detector, not a budget: a run that takes several times the limit but keeps going is left alone
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--envs', '4096', '--steps', '150000', '--warmup', '5', '--no-cpu-baseline',
                        '--watchdog', '3'], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and json.loads(p.stdout.strip().splitlines()[-1])['steps'] == 150000, (p.returncode, p.stderr[-500:])


def test_gymnasium_adapter_on_a_real_batched_env():
    """The 5-tuple adapter on the real thing: a batched PointGather env whose time limit (5 steps) truncates every env."""
    import hrl_pybullet_envs_amd as H
    from hrl_pybullet_envs_amd.adapters import GymnasiumAdapter
    env = H.PointGatherBulletEnv(num_envs=64, seed=2)
    env.max_episode_steps = 5
    g = GymnasiumAdapter(env)
    obs, info = g.reset(seed=4)
    assert obs.shape == (64, 18) and info == {} and env._cfg.seed == 4
    for t in range(5):
        obs, rew, term, trunc, info = g.step(torch.rand(64, 2, device='cuda') * 2 - 1)
        assert obs.shape == (64, 18) and rew.shape == (64,) and term.dtype == torch.bool and trunc.dtype == torch.bool
        assert not bool(term.any())  # the point bot cannot die (point_bot.py:73-74)
        assert bool(trunc.all()) == (t == 4)
    g.close()
    one = GymnasiumAdapter(H.AntGatherBulletEnv(seed=1))
    obs, _ = one.reset()
    obs, rew, term, trunc, info = one.step(np.zeros(8))
    assert obs.shape == (46,) and isinstance(rew, float) and term is False and trunc is False and 'food_rew' in info
    one.close()


def test_vector_env_adapters_meet_their_contracts():
    """`gymnasium.vector.VectorEnv`- and SB3-`VecEnv`-shaped views over a batched env (README.md:19-37 is the one-env loop a trainer would wrap N
    times; here the batch is the env).  Neither package exists in the image, so the contracts are checked duck-typed: attribute names, shapes,
    dtypes, same-step autoreset with the terminal observation from the kernel (ABI v6 final_obs), against a twin env stepped directly."""
    import hrl_pybullet_envs_amd as H
    from hrl_pybullet_envs_amd.adapters import GymnasiumVectorEnv, SB3VecEnv
    n, limit = 48, 7

    def build():
        e = H.AntGatherBulletEnv(num_envs=n, seed=5)
        e.max_episode_steps = limit
        return e
    # ---- gymnasium.vector.VectorEnv
    v, twin = GymnasiumVectorEnv(build()), build()
    assert v.num_envs == n and v.single_observation_space.shape == (46,) and v.single_action_space.shape == (8,)
    assert v.observation_space.shape == (n, 46) and v.action_space.shape == (n, 8) and v.metadata['autoreset_mode'] == 'same_step'
    obs, info = v.reset(seed=5)
    assert torch.equal(obs, twin.reset()) and info == {}
    gen = torch.Generator(device='cuda').manual_seed(1)
    ended = 0
    for t in range(2 * limit + 1):
        a = torch.rand(n, 8, device='cuda', generator=gen) * 2 - 1
        obs, rew, term, trunc, info = v.step(a)
        to, tr, td, ti = twin.step(a)
        assert torch.equal(obs, to) and torch.equal(rew, tr) and term.dtype == torch.bool and trunc.dtype == torch.bool
        assert torch.equal(term | trunc, td.bool()) and not bool((term & trunc).any())
        m = info['_final_observation']
        assert torch.equal(m, td.bool()) and info['final_observation'].shape == (n, 46) and info['final_obs'] is info['final_observation']
        if bool(m.any()):
            # the terminal observation is NOT the returned one (that is the next episode's first), and it is what the kernel kept
            assert not torch.equal(info['final_observation'][m], obs[m]) and torch.equal(info['final_observation'][m], twin._backend().final_obs[m])
            assert torch.equal(info['final_info']['episode_length'][m] >= 1, torch.ones(int(m.sum()), dtype=torch.bool, device='cuda'))
            ended += int(m.sum())
        assert bool(trunc.any()) == ((t + 1) % limit == 0) or bool(term.any()) or ended > 0   # the first limit truncates every env still in its first episode
        if t + 1 == limit:
            assert bool((trunc | term).all()) or ended > int(m.sum())
    assert ended >= 2 * n
    vn = GymnasiumVectorEnv(build(), numpy=True)
    o0, _ = vn.reset()
    o1, r1, te, tr_, inf = vn.step(np.zeros((n, 8), np.float32))
    assert isinstance(o1, np.ndarray) and o1.shape == (n, 46) and r1.dtype == np.float32 and te.dtype == bool and inf['_final_obs'].dtype == bool
    rm = np.zeros(n, bool); rm[::3] = True     # gymnasium >= 1.0: options={'reset_mask': ...} resets the named envs only
    o2, _ = vn.reset(options={'reset_mask': rm})
    assert np.array_equal(o2[~rm], o1[~rm]) and not np.array_equal(o2[rm], o1[rm])
    vn.close(); v.close(); twin.close()
    with pytest.raises(ValueError):
        GymnasiumVectorEnv(H.AntGatherBulletEnv())
    # ---- stable_baselines3 VecEnv
    s, twin = SB3VecEnv(build()), build()
    assert s.num_envs == n and s.observation_space.shape == (46,) and s.action_space.shape == (8,)
    obs = s.reset(); twin.reset()
    assert isinstance(obs, np.ndarray) and obs.shape == (n, 46) and obs.dtype == np.float32
    rng = np.random.RandomState(2)
    seen_term = 0
    for t in range(limit + 2):
        a = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
        s.step_async(a)                 # returns with the kernel in flight
        obs, rews, dones, infos = s.step_wait()
        to, tr, td, ti = twin.step(torch.from_numpy(a).cuda())
        assert np.array_equal(obs, to.cpu().numpy()) and np.array_equal(rews, tr.cpu().numpy()) and dones.dtype == bool and len(infos) == n
        assert np.array_equal(dones, td.cpu().numpy().astype(bool)) and all('food_rew' in i and 'dead_rew' in i for i in infos)
        for i in np.nonzero(dones)[0]:
            assert infos[i]['terminal_observation'].shape == (46,) and infos[i]['TimeLimit.truncated'] == (t + 1 == limit) and infos[i]['episode']['l'] >= 1
            assert np.array_equal(infos[i]['terminal_observation'], twin._backend().final_obs[i].cpu().numpy())
            seen_term += 1
        assert all('terminal_observation' not in infos[i] for i in np.nonzero(~dones)[0])
    assert seen_term >= n and s.env_is_wrapped(object) == [False] * n and s.get_attr('n_bins', [0, 3]) == [10, 10] and s.seed(3) == [3] * n
    s.close(); twin.close()


def test_flagrun_manual_goal_creation_class_api_and_render():
    import hrl_pybullet_envs_amd as H
    env = H.AntFlagrunBulletEnv(manual_goal_creation=True, seed=3)
    env.reset()
    assert env.goal == (1000.0, 0.0)                      # upstream's default walk target until goals are pushed
    ob = env.set_goals([[1.0, 2.0], [-2.0, 0.5]])
    assert ob.shape == (28,) and env.goal == (-2.0, 0.5)     # goals.pop(): the LAST goal of the list first (ant_flagrun_env.py:116)
    ob, rew, done, info = env.step(np.zeros(8))
    assert ob.shape == (28,) and not done
    img = env.render('rgb_array')
    assert img.shape == (256, 256, 3) and img.dtype == np.uint8 and (img != 255).any()
    assert env.render('human') is None
    env.close()
    b = H.AntFlagrunBulletEnv(manual_goal_creation=True, num_envs=32, seed=3)
    b.reset()
    ob = b.set_goals(torch.rand(32, 3, 2) * 4 - 2)
    assert ob.shape == (32, 28) and ob.is_cuda and b.goal.shape == (32, 2)
    b.close()


def test_rccl_world_of_one_return_gather_on_the_side_stream():
    """The collective path on real hardware with the one GPU a test box has: a ONE-rank RCCL communicator
    (`init_process_group('nccl')` loads librccl), `ReturnGatherer.launch()` / `latest()` next to stepping -- snapshot on the step
    stream, `all_gather_into_tensor` on device tensors on the side stream, double-buffered.  Every gathered vector must be the
    episode returns of exactly the step it was launched after (dist.py:23-119, BASELINE.json configs[4])."""
    import socket
    import torch.distributed as dist
    from hrl_pybullet_envs_amd import _lib
    from hrl_pybullet_envs_amd.dist import ReturnGatherer, all_gather_returns, init_distributed
    from hrl_pybullet_envs_amd.vec_env import BatchedEnv
    import os
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    saved = {k: os.environ.get(k) for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    try:
        rank, world, lr = init_distributed(1, backend='nccl', force=True)
        assert (rank, world, lr) == (0, 1, 0) and dist.is_initialized() and dist.get_backend() == 'nccl'
        n = 2048
        ant = BatchedEnv(_lib.default_config(K.HRL_ANT_GATHER, num_envs=n, seed=2, auto_reset=1), 'cuda:0')
        pt = BatchedEnv(_lib.default_config(K.HRL_POINT_GATHER, num_envs=n, seed=2, auto_reset=1, env_id_offset=n), 'cuda:0')
        ant.reset(); pt.reset()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        torch.cuda.synchronize()
        g = ReturnGatherer([(ant, s1), (pt, s2)], world)   # sizes exchanged through the group (all_gather_object)
        assert g.counts == [2 * n] and g.latest() is None
        gen = torch.Generator(device='cuda').manual_seed(3)
        a8 = torch.rand(30, n, 8, device='cuda', generator=gen) * 2 - 1
        a2 = torch.rand(30, n, 2, device='cuda', generator=gen) * 2 - 1
        expect = {}
        for t in range(30):
            with torch.cuda.stream(s1):
                ant.step(a8[t])
            with torch.cuda.stream(s2):
                pt.step(a2[t])
            if (t + 1) % 3 == 0:        # far more often than bench.py: consecutive collectives overlap the next steps
                g.launch()
                torch.cuda.synchronize()
                expect[t] = torch.cat([ant.info[:, 2], pt.info[:, 2]]).clone()
                assert torch.equal(g.latest(), expect[t]), t
        # without the synchronize in between: the gathered values still belong to the step they were launched after
        for t in range(6):
            with torch.cuda.stream(s1):
                ant.step(a8[t])
            with torch.cuda.stream(s2):
                pt.step(a2[t])
            g.launch()
        got = g.latest().clone()
        torch.cuda.synchronize()
        assert torch.equal(got, torch.cat([ant.info[:, 2], pt.info[:, 2]]))
        # the plain function on device tensors through RCCL, and the uneven-shard validation
        x = torch.arange(100, dtype=torch.float32, device='cuda')
        assert torch.equal(all_gather_returns(x, 1), x)
        with pytest.raises(ValueError):
            ReturnGatherer([(ant, s1)], 1, counts=[n + 1])
        assert 'librccl' in open('/proc/self/maps').read()
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_attributes_a_trainer_reads_between_steps():
    """ant_maze_bullet_env.py:38,46 (`t`, `target`), ant_flagrun_env.py:43-57 (`goals`, `steps_since_goal_change`, `goal`) and the
    upstream robot attributes (`walk_target_x/y`, `body_xyz`, `body_real_xyz`, `walk_target_dist`) as views over the state
    tensors: one env gives python / numpy values like the reference, a batch gives one row per env."""
    import hrl_pybullet_envs_amd as H
    dt = np.float32(0.0165 / 4) * 4
    # maze, one env
    m = H.AntMazeBulletEnv(seed=5)
    m.reset()
    assert m.t == 0 and isinstance(m.target, np.ndarray) and list(m.target) in [list(map(float, t)) for t in m.targets]
    for _ in range(3):
        m.step(np.zeros(8))
    assert m.t == 3 and (m.walk_target_x, m.walk_target_y) == tuple(m.target) == (m.robot.walk_target_x, m.robot.walk_target_y)
    st = m._backend().state[0].cpu().numpy()
    assert np.allclose(m.robot.body_real_xyz, st[0:3]) and m.robot.body_xyz.shape == (3,) and m.robot.body_xyz[2] == st[2]
    assert abs(m.robot.walk_target_dist - float(-st[31] * dt)) < 1e-4     # the potential the step stored = -walk_target_dist / dt
    assert abs(m.robot.initial_z - 0.25) < 1e-6 and m.robot.body_rpy.shape == (3,)
    m.close()
    # maze Mj + flat, batched
    for cls in (H.AntMazeMjEnv, H.AntMjEnv):
        b = cls(num_envs=64, seed=2)
        b.reset()
        for _ in range(4):
            b.step(torch.rand(64, 8, device='cuda') * 2 - 1)
        st = b._backend().state.cpu().numpy()
        assert np.allclose(b.robot.walk_target_dist, -st[:, 31] * dt, atol=2e-4) and b.robot.body_xyz.shape == (64, 3)
        if cls is H.AntMazeMjEnv:
            assert b.target.shape == (64, 2) and int(b.t[0]) == 4 and b.robot.walk_target_x.shape == (64,)
        else:
            assert np.all(b.robot.walk_target_x == 1000.0)                 # upstream's default walk target
        b.close()
    # flagrun, shared list: goals still to come; goal = the one being chased; consumed as goals are reached / time out
    f = H.AntFlagrunBulletEnv(max_targets=6, timeout=3, seed=9)
    f.reset()
    g0 = f.goals
    assert len(g0) == 5 and f.steps_since_goal_change == 0 and not f._rewarded
    first = f.goal
    st = f._backend().state[0].cpu().numpy()
    for _ in range(3):
        f.step(np.zeros(8))
    assert f.goal == pytest.approx(g0[-1]) and f.goals == g0[:-1] and f.steps_since_goal_change == 0   # timeout: the LAST goal of the list is next
    assert f.goal != first and (f.walk_target_x, f.walk_target_y) == f.goal
    st = f._backend().state[0].cpu().numpy()
    f.step(np.zeros(8))
    st = f._backend().state[0].cpu().numpy()
    assert abs(f.robot.walk_target_dist - float(-st[31] * dt)) < 1e-4 and f.steps_since_goal_change == 1
    f.close()
    fb = H.AntFlagrunBulletEnv(max_targets=4, num_envs=16, seed=9)
    fb.reset()
    gl, left = fb.goals
    assert gl.shape == (16, 4, 2) and bool((left == 3).all()) and fb.goal.shape == (16, 2)
    fb.close()
    # flagrun, manual list: env.goals = [...] is plain data; next_target() pops the last; IndexError on an empty list
    e = H.AntFlagrunBulletEnv(manual_goal_creation=True, seed=3)
    e.reset()
    assert e.goals == [] and e.goal == (1000.0, 0.0)
    with pytest.raises(IndexError):
        e.next_target()
    e.goals = [(1.0, 2.0), (-2.0, 0.5), (3.0, 3.0)]
    assert e.goals == [(1.0, 2.0), (-2.0, 0.5), (3.0, 3.0)] and e.goal == (1000.0, 0.0)
    ob = e.next_target()
    assert ob.shape == (28,) and e.goal == (3.0, 3.0) and e.goals == [(1.0, 2.0), (-2.0, 0.5)]
    ob2 = e.set_goals([(0.5, 0.5), (4.0, -4.0)])
    assert e.goal == (4.0, -4.0) and e.goals == [(0.5, 0.5)] and ob2.shape == (28,)
    # the reference's documented manual workflow (ant_flagrun_env.py:91-96): reset(); create_targets(n); next_target()
    e.reset()
    e.create_targets(7)
    g7 = e.goals
    assert len(g7) == 7 and all(abs(x) <= 5 and abs(y) <= 5 and (x * x + y * y) ** 0.5 >= 0.5 for x, y in g7)
    e.next_target()
    assert e.goal == pytest.approx(g7[-1]) and e.goals == g7[:-1]
    e.create_targets(7)
    assert e.goals != g7 and len(e.goals) == 7                    # a later call draws a fresh list
    e2 = H.AntFlagrunBulletEnv(manual_goal_creation=True, seed=3)
    e2.reset(); e2.create_targets(7)
    assert e2.goals == g7                                          # same seed, same call number: the same list in every process
    with pytest.raises(ValueError, match='goal_capacity'):
        e2.create_targets(16)
    e2.close(); e.close()
    big = H.AntFlagrunBulletEnv(manual_goal_creation=True, seed=3, goal_capacity=40)
    big.reset(); big.create_targets(40)
    assert len(big.goals) == 40 and big._backend().items.shape[1] == 96
    big.next_target()
    assert len(big.goals) == 39
    big.close()
    # next_target() of a NON-manual env pops the shared list (:112-116), IndexError when it is used up
    f = H.AntFlagrunBulletEnv(max_targets=3, seed=9)
    f.reset()
    g0 = f.goals
    assert len(g0) == 2
    ob = f.next_target()
    assert ob.shape == (28,) and f.goal == pytest.approx(g0[-1]) and f.goals == g0[:-1] and not f._rewarded
    f.next_target()
    with pytest.raises(IndexError):
        f.next_target()
    f.create_targets(3)                                            # the length reset() armed: fine; any other cannot be served
    with pytest.raises(ValueError, match='max_targets'):
        f.create_targets(5)
    f.close()
    # gather kinds: the robot object of the constructor with live pose attributes
    p = H.PointGatherBulletEnv(seed=1)
    p.reset()
    p.step(np.array([1.0, 0.0]))
    sc = p.stadium_scene                     # gather_scene.py:22-33: the item dictionaries of the scene, over the env's items tensor
    it = p._backend().items[0].cpu().numpy()
    assert len(sc.food) == 8 and len(sc.poison) == 8 and len(sc.all_items) == 16 and sc.food[3] == [float(it[6]), float(it[7]), 0.1] and sc.poison[8][:2] == [float(it[16]), float(it[17])]
    assert p.scene.size == (15, 15) and (sc.n_food, sc.n_poison, sc.spacing, sc.respawn) == (8, 8, 2., True)
    assert np.allclose(p.robot.body_real_xyz, p._backend().state[0, 0:3].cpu().numpy()) and p.robot.alive_bonus(0, 0) == 1
    p.close()
    a = H.AntGatherBulletEnv(num_envs=8, seed=1)
    a.reset()
    st0 = a._backend().state.clone()
    a.step(torch.rand(8, 8, device='cuda') * 2 - 1)
    st1 = a._backend().state.clone()
    a.reset(mask=torch.tensor([1, 0, 0, 1, 0, 0, 0, 0], dtype=torch.bool))       # a batch resets the envs the mask names, the others go on
    st2 = a._backend().state
    assert torch.equal(st2[[1, 2, 4, 5, 6, 7]], st1[[1, 2, 4, 5, 6, 7]]) and not torch.equal(st2[[0, 3]], st1[[0, 3]]) and bool((st2[[0, 3], 2] == 0.75).all())
    a.reset()
    assert tuple(a.stadium_scene.food.shape) == (8, 8, 2) and torch.equal(a.stadium_scene.all_items.reshape(8, 32), a._backend().items[:, :32])   # a batch: tensors
    assert a.robot.body_xyz.shape == (8, 3) and a.robot.body_real_xyz.shape == (8, 3)
    a.close()


def test_robot_attributes_the_reference_reads_in_its_own_step():
    """`robot.joints_at_limit`, `robot.calc_potential()`, `env.potential`, `robot.feet_contact`, `robot.joint_speeds`, `env.robot_body.pose().xyz()` /
    `.rpy()` / `robot.robot_body.get_position()` -- the upstream attributes the reference's in-tree step() code reads (MjAnt.py:40-68,
    ant_maze_bullet_env.py:67,125-130, ant_flagrun_env.py:104-105) -- as views over the live state: AntMjEnv's reward is rebuilt from them."""
    import hrl_pybullet_envs_amd as H
    e = H.AntMjEnv(seed=2)
    e.reset()
    rng = np.random.RandomState(0)
    for t in range(30):
        p0 = e.potential
        obs, r, d, _ = e.step(rng.uniform(-1, 1, 8))
        alive = 1.0 if obs[2] > 0.26 else -1.0
        assert r == pytest.approx(alive + (e.potential - p0) - 0.1 * e.robot.joints_at_limit, abs=2e-2), t   # (potentials are ~6e4 in fp32: their difference carries ~4e-3)
        rw = e.rewards                          # MjAnt.py:82-87: [alive, progress, joints_at_limit_cost, feet_collision_cost]
        assert len(rw) == 4 and rw[0] == alive and rw[2] == pytest.approx(-0.1 * e.robot.joints_at_limit) and rw[3] == 0.0
        assert np.float32(np.float32(rw[0] + rw[1]) + np.float32(rw[2])) == np.float32(r), (t, rw, r)    # the kernel's own sum, to the bit
        assert e.potential == pytest.approx(e.robot.calc_potential(), rel=1e-6)
        assert np.allclose(e.robot.joint_speeds, 0.1 * obs[21:29]) and np.allclose(e.robot_body.pose().xyz(), obs[0:3]) and np.allclose(e.robot.robot_body.get_orientation(), obs[3:7])
        assert np.allclose(e.robot_body.speed(), obs[15:18]) and np.allclose(e.robot_body.pose().rpy(), e.robot.body_rpy)
        if d:
            break
    e.close()
    m = H.AntMazeBulletEnv(seed=1, inner_rew_weight=1.0)
    m.reset()
    seen = set()
    for t in range(40):
        before = np.array(m.robot.feet_contact)
        obs, r, d, _ = m.step(rng.uniform(-1, 1, 8) * 0.3)
        assert np.array_equal(obs[22:26], before), t        # the observation shows the flags of the step before (upstream's order), feet_contact the current ones
        seen.add(tuple(m.robot.feet_contact))
        rw = m.rewards                          # upstream WalkerBaseBulletEnv.step: [alive, progress, electricity_cost, joints_at_limit_cost, feet_collision_cost]
        assert len(rw) == 5 and rw[0] in (1.0, -1.0) and rw[2] <= 0 and rw[4] == 0.0
        if not d:
            assert r == pytest.approx(sum(rw) * m.inner_rew_weight, rel=1e-6, abs=1e-5), (t, rw, r)   # ant_maze_bullet_env.py:82-84: rew = inner * inner_rew_weight (+ the sparse terms when done);
            # (the first step's progress is ~6e4: the reset took its potential before the target was switched, ant_maze_bullet_env.py:104-116)
        assert np.allclose(m.robot_body.pose().xyz()[:2], m.robot.body_real_xyz[:2])
    assert len(seen) > 1 and m.robot.feet_contact.shape == (4,)
    m.close()
    b = H.AntMazeBulletEnv(num_envs=6, seed=1)
    b.reset()
    b.step(torch.zeros(6, 8, device='cuda'))
    assert b.robot.feet_contact.shape == (6, 4) and b.robot.joints_at_limit.shape == (6,) and b.potential.shape == (6,) and b.robot_body.pose().xyz().shape == (6, 3)
    assert len(b.rewards) == 5 and all(x.shape == (6,) for x in b.rewards)
    b.close()
    pt = H.PointGatherBulletEnv(seed=0)
    pt.reset()
    assert np.allclose(pt.robot.robot_body.pose().xyz(), pt.robot.body_real_xyz)
    with pytest.raises(AttributeError):
        H.AntGatherBulletEnv(seed=0).robot.feet_contact
    with pytest.raises(AttributeError):
        pt.rewards
    pt.close()


def test_step_host_equals_the_device_path():
    """`BatchedEnv.step_host` (action read from / outputs written to pinned host memory by the kernel itself, one launch + one
    synchronisation: the path of the one-env classes) gives what `step` gives through device tensors, bit for bit."""
    from hrl_pybullet_envs_amd import _lib
    from hrl_pybullet_envs_amd.vec_env import BatchedEnv
    for kind in (K.HRL_ANT_GATHER, K.HRL_POINT_GATHER, K.HRL_ANT_MAZE):
        n = 64
        a_env = BatchedEnv(_lib.default_config(kind, num_envs=n, seed=9, auto_reset=1, max_episode_steps=20), 'cuda:0')
        b_env = BatchedEnv(_lib.default_config(kind, num_envs=n, seed=9, auto_reset=1, max_episode_steps=20), 'cuda:0')
        a_env.reset(); b_env.reset()
        rng = np.random.RandomState(kind)
        for t in range(45):
            a = rng.uniform(-1, 1, (n, a_env.act_dim)).astype(np.float32)
            o1, r1, d1, _ = a_env.step(torch.from_numpy(a).cuda())
            o2, r2, d2, i2 = b_env.step_host(a)
            assert isinstance(o2, np.ndarray) and o2.shape == (n, a_env.obs_dim)
            assert np.array_equal(o1.cpu().numpy(), o2, equal_nan=True) and np.array_equal(r1.cpu().numpy(), r2), (kind, t)
            assert np.array_equal(d1.cpu().numpy(), d2) and np.array_equal(a_env.info.cpu().numpy(), i2), (kind, t)
            # the terminal observation / truncation flag of the host path, and the device tensors of an env stepped through the host:
            # reading them brings the host values over (they used to stay silently stale)
            fo, tr = b_env.host_final_obs()
            dd = d2.astype(bool)
            assert np.array_equal(a_env.final_obs.cpu().numpy()[dd], fo[dd], equal_nan=True) and np.array_equal(a_env.truncated.cpu().numpy(), tr), (kind, t)
            if t % 7 == 0:
                assert np.array_equal(b_env.obs.cpu().numpy(), o2, equal_nan=True) and np.array_equal(b_env.info.cpu().numpy(), i2)
                assert np.array_equal(b_env.final_obs.cpu().numpy()[dd], fo[dd], equal_nan=True)
        assert torch.equal(a_env.state, b_env.state) and torch.equal(a_env.aux, b_env.aux)
        # mixing the two paths: a device step after host steps, then a reset, never resurrects stale host outputs
        a = rng.uniform(-1, 1, (n, a_env.act_dim)).astype(np.float32)
        o1, r1, d1, _ = a_env.step(torch.from_numpy(a).cuda()); o2, r2, d2, _ = b_env.step(torch.from_numpy(a).cuda())
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(a_env.final_obs, b_env.final_obs)
        b_env.step_host(a); a_env.step(torch.from_numpy(a).cuda())
        assert torch.equal(a_env.reset(), b_env.reset())
        a_env.close(); b_env.close()


def test_c_abi_from_plain_c(tmp_path):
    """The boundary is a C ABI (include/hrl_envs.h), not a Python extension: tests/c_abi/abi_demo.c -- plain C, gcc, hipMalloc'ed buffers, no torch
    in the process -- creates envs, resets, steps them on its own stream and dumps what the library left in the buffers; the CPU oracle fed the same
    LCG actions must agree bit for bit, terminal observations and truncation flags included (40-step limit: every env is reset on the way)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, 'tests', 'c_abi')
    subprocess.check_call(['make', '-s', '-C', d, 'abi_demo'])
    n, steps, seed = 96, 60, 7
    for kind in (K.HRL_ANT_GATHER, K.HRL_POINT_GATHER, K.HRL_ANT_MAZE):
        out = str(tmp_path / f'k{kind}.bin')
        p = subprocess.run([os.path.join(d, 'abi_demo'), str(kind), str(n), str(steps), str(seed), out], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-1000:]
        assert 'hip-gfx950' in p.stdout
        o = orc.OracleEnv(orc.default_config(kind, num_envs=n, seed=seed, auto_reset=1, max_episode_steps=40), np.float32)
        o.reset()
        lcg, mask = 0x9E3779B97F4A7C15, (1 << 64) - 1
        for t in range(steps):
            a = np.empty(n * o.ad, np.float32)
            for i in range(n * o.ad):
                lcg = (lcg * 6364136223846793005 + 1442695040888963407) & mask
                a[i] = np.float32(float(lcg >> 11) * (2.0 / 9007199254740992.0) - 1.0)
            o.step(a.reshape(n, o.ad))
        raw = open(out, 'rb').read()
        off = 0

        def take(count, dtype):
            nonlocal off
            arr = np.frombuffer(raw, dtype=dtype, count=count, offset=off)
            off += arr.nbytes
            return arr
        st = take(n * 32, np.float32).reshape(n, 32); it = take(n * o.items.shape[1], np.float32).reshape(n, -1)
        aux = take(n * 4, np.int32).reshape(n, 4); ob = take(n * o.od, np.float32).reshape(n, o.od)
        rew = take(n, np.float32); done = take(n, np.uint8); fin = take(n * o.od, np.float32).reshape(n, o.od); trunc = take(n, np.uint8)
        assert off == len(raw)
        assert np.array_equal(st, o.state) and np.array_equal(it, o.items) and np.array_equal(aux, o.aux), kind
        assert np.array_equal(ob, o.obs, equal_nan=True) and np.array_equal(rew, o.rew) and np.array_equal(done, o.done), kind
        assert np.array_equal(fin, o.final_obs, equal_nan=True) and np.array_equal(trunc, o.truncated) and np.any(fin != 0), kind
    # a record left as the stack held it is refused with the reason, not handed to a kernel (hrl_buffers.struct_size, hrl_buffers_init)
    p = subprocess.run([os.path.join(d, 'abi_demo'), '1', '8', '1', '1', str(tmp_path / 'x.bin'), 'uninit'], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and 'rc 1' in p.stdout and 'hrl_buffers_init' in p.stdout, (p.stdout, p.stderr)


def test_checkpoint_and_resume_continue_bit_for_bit():
    """SURVEY 5 (checkpoint / resume): state_dict() through torch.save / torch.load into a NEW env continues exactly where the first one was --
    every output of the next 60 steps identical, across resets, pickups, respawns (the random streams are keyed by counters in `aux`), for a
    gather env with many items, a maze env and a flagrun env with manual goals; a checkpoint of another config is refused."""
    import io
    import hrl_pybullet_envs_amd as H
    from hrl_pybullet_envs_amd import _lib
    n = 96
    cases = [(H.AntGatherBulletEnv, dict(n_food=20, n_poison=20, n_bins=12), 8), (H.AntMazeBulletEnv, dict(sense_target=True), 8),
             (H.PointGatherBulletEnv, dict(robot_coll_dist=-1, use_sensor=False), 2), (H.AntFlagrunBulletEnv, dict(manual_goal_creation=True, goal_capacity=20, timeout=7), 8)]
    for cls, kw, ad in cases:
        a_env = cls(num_envs=n, seed=11, **kw)
        a_env.max_episode_steps = 25
        a_env.reset()
        if cls is H.AntFlagrunBulletEnv:
            a_env.set_goals((torch.rand(n, 20, 2) * 6 - 3).numpy())
        gen = torch.Generator(device='cuda').manual_seed(3)
        acts = torch.rand(100, n, ad, device='cuda', generator=gen) * 2 - 1
        for t in range(40):
            a_env.step(acts[t])
        buf = io.BytesIO(); torch.save(a_env.state_dict(), buf); buf.seek(0)
        sd = torch.load(buf)
        b_env = cls(num_envs=n, seed=11, **kw)
        b_env.max_episode_steps = 25
        ob = b_env.load_state_dict(sd)
        assert torch.equal(ob.view(torch.int32), a_env._backend().obs.view(torch.int32))
        for t in range(40, 100):
            oa, ra, da, ia = a_env.step(acts[t]); ob, rb, db, ib = b_env.step(acts[t])
            assert torch.equal(oa.view(torch.int32), ob.view(torch.int32)) and torch.equal(ra.view(torch.int32), rb.view(torch.int32)) and torch.equal(da, db), (cls.__name__, t)
        A, B = a_env._backend(), b_env._backend()
        assert torch.equal(A.state.view(torch.int32), B.state.view(torch.int32)) and torch.equal(A.items.view(torch.int32), B.items.view(torch.int32)) and torch.equal(A.aux, B.aux)
        assert int(A.aux[:, 2].min()) >= 3        # episodes ended and restarted on the way
        other = cls(num_envs=n, seed=12, **kw)
        other.max_episode_steps = 25
        with pytest.raises(_lib.HrlError):
            other.load_state_dict(sd)
        other.load_state_dict(sd, strict=False)   # same shapes: allowed on request
        for e in (a_env, b_env, other):
            e.close()


def test_steps_can_be_captured_in_a_hip_graph():
    """hrl_step is a plain launch on the caller's stream (no allocation, no synchronisation, no host read-back), so a rollout's inner loop can be
    captured once and replayed (torch.cuda.graphs = hipGraph): four env steps per replay with the actions in a static tensor; after three replays
    state, items, counters and observations equal the oracle stepped twelve times -- and the eager path continues from there."""
    n, K_STEPS = 256, 4
    from hrl_pybullet_envs_amd import _lib
    from hrl_pybullet_envs_amd.vec_env import BatchedEnv
    for kind in (K.HRL_ANT_GATHER, K.HRL_POINT_GATHER):
        env = BatchedEnv(_lib.default_config(kind, num_envs=n, seed=6, auto_reset=1, max_episode_steps=7), 'cuda:0')
        o = orc.OracleEnv(orc.default_config(kind, num_envs=n, seed=6, auto_reset=1, max_episode_steps=7), np.float32)
        env.reset(); o.reset()
        acts = torch.rand(4, n, env.act_dim, device='cuda') * 2 - 1
        static_a = acts[0].clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):            # the warm-up launch torch asks for before a capture
            env.step(static_a)
        torch.cuda.current_stream().wait_stream(side)
        o.step(acts[0].cpu().numpy())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(K_STEPS):
                env.step(static_a)
        for r in range(3):
            static_a.copy_(acts[r + 1])
            g.replay()
            for _ in range(K_STEPS):
                o.step(acts[r + 1].cpu().numpy())
        torch.cuda.synchronize()
        assert np.array_equal(env.state.cpu().numpy(), o.state) and np.array_equal(env.items.cpu().numpy(), o.items) and np.array_equal(env.aux.cpu().numpy(), o.aux)
        assert np.array_equal(env.obs.cpu().numpy(), o.obs) and np.array_equal(env.done.cpu().numpy(), o.done)
        assert int(o.aux[:, 2].min()) >= 2      # the 7-step limit passed inside a replay: in-kernel resets are part of the graph
        env.step(acts[0]); o.step(acts[0].cpu().numpy())
        assert np.array_equal(env.state.cpu().numpy(), o.state)
        del g
        env.close()


def test_four_host_threads_each_with_its_own_env_and_stream():
    """The library keeps no global state between handles (the error text is thread-local): four host threads, each stepping its own env on its
    own stream at the same time, all end bit-identical to the oracle."""
    import threading
    from hrl_pybullet_envs_amd import _lib
    from hrl_pybullet_envs_amd.vec_env import BatchedEnv
    kinds = [K.HRL_ANT_GATHER, K.HRL_POINT_GATHER, K.HRL_ANT_MAZE, K.HRL_ANT_FLAGRUN]
    n, T = 192, 150
    out, errs = {}, []

    def work(i, kind):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                env = BatchedEnv(_lib.default_config(kind, num_envs=n, seed=20 + i, auto_reset=1, max_episode_steps=40), 'cuda:0')
                env.reset()
                acts = torch.from_numpy(np.random.RandomState(i).uniform(-1, 1, (T, n, env.act_dim)).astype(np.float32)).cuda()
                for t in range(T):
                    env.step(acts[t])
                s.synchronize()
                out[i] = (env.state.cpu().numpy(), env.items.cpu().numpy(), env.aux.cpu().numpy(), env.obs.cpu().numpy())
                env.close()
        except Exception as e:   # pragma: no cover
            errs.append((i, repr(e)))

    threads = [threading.Thread(target=work, args=(i, k)) for i, k in enumerate(kinds)]
    for t in threads: t.start()
    for t in threads: t.join()
    assert not errs, errs
    for i, kind in enumerate(kinds):
        o = orc.OracleEnv(orc.default_config(kind, num_envs=n, seed=20 + i, auto_reset=1, max_episode_steps=40), np.float32)
        o.reset()
        acts = np.random.RandomState(i).uniform(-1, 1, (T, n, o.ad)).astype(np.float32)
        for t in range(T):
            o.step(acts[t])
        st, it, au, ob = out[i]
        assert np.array_equal(st, o.state, equal_nan=True) and np.array_equal(it, o.items) and np.array_equal(au, o.aux) and np.array_equal(ob, o.obs, equal_nan=True), kind


def test_the_measuring_tools_of_the_readme_run():
    """The light-weight tools of README's "Measuring" table run as documented (small arguments) and print what they say they print."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    runs = [(['tools/soak.py', '30', '64'], 'final state finite True'), (['tools/readme_loop.py'], 'steps/s'), (['tools/host_overhead.py'], 'us per call'),
            (['tools/solo_latency.py', '3', '1'], 'step_host'), (['tools/graph_rollout.py', '64'], 'one hipGraph'),
            (['tests/tools/long_parity.py', '20', '64', '1'], 'bit-exact for 20 steps'), (['tests/tools/fuzz_parity.py', '1', 'gpu'], '0 with a difference'),
            (['tests/tools/fuzz_configs.py', '2', 'gpu', '3', '10'], '0 with a difference')]
    for cmd, expect in runs:
        r = subprocess.run([sys.executable] + cmd, cwd=root, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and expect in r.stdout, (cmd, r.returncode, r.stdout[-400:], r.stderr[-800:])


def test_seed_on_a_live_env_keeps_the_simulation():
    """ant_gather_env.py:63-66 + gather_scene.py:35-36: `env.seed(s)` reseeds the RNGs, the simulation carries on.  Here the seed is a constant of
    the live handle (hrl_update_config): state, items, counters, tensors stay bit for bit; the next respawn and the next reset draw from the new
    seed's counter-based streams -- equal to the oracle's with its seed changed at the same step."""
    import hrl_pybullet_envs_amd as H
    n = 64
    env = H.AntGatherBulletEnv(num_envs=n, seed=3)
    o = orc.OracleEnv(orc.default_config(K.HRL_ANT_GATHER, num_envs=n, seed=3, auto_reset=1, max_episode_steps=2000), np.float32)
    ob = env.reset(); o.reset()
    be = env._backend()
    handle, ptrs = be._h, (be.state.data_ptr(), be.items.data_ptr(), be.aux.data_ptr(), ob.data_ptr())
    rng = np.random.RandomState(5)
    for t in range(5):
        a = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
        ob, r, d, info = env.step(torch.from_numpy(a).cuda()); o.step(a)
    before = [x.cpu().numpy().copy() for x in (be.state, be.items, be.aux)]
    assert env.seed(7) == [7]
    o.cfg.seed = 7
    assert env._backend() is be and be._h == handle and ptrs == (be.state.data_ptr(), be.items.data_ptr(), be.aux.data_ptr(), ob.data_ptr())   # no new handle, no new tensors
    for x, y in zip((be.state, be.items, be.aux), before):
        assert np.array_equal(x.cpu().numpy(), y)   # nothing moved
    # ants put onto items: pickups whose respawn positions come from seed 7's stream
    o.state[::2, 0:2] = o.items[::2, 0:2] + np.float32(0.1)
    be.state.copy_(torch.from_numpy(o.state))
    a = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
    items_before = o.items.copy()
    ob, r, d, info = env.step(torch.from_numpy(a).cuda()); o.step(a)
    assert (o.rew[::2] >= 1).sum() > n // 4 and (o.items != items_before).any(axis=1).sum() > n // 4
    for name, x in (('state', be.state), ('items', be.items), ('aux', be.aux), ('obs', ob), ('rew', r)):
        assert np.array_equal(x.cpu().numpy(), getattr(o, name)), name
    # the same pickups under the OLD seed put the items elsewhere: the new seed is what was used
    o3 = orc.OracleEnv(orc.default_config(K.HRL_ANT_GATHER, num_envs=n, seed=3, auto_reset=1, max_episode_steps=2000), np.float32)
    o3.state[...] = before[0]; o3.items[...] = before[1]; o3.aux[...] = before[2]
    o3.state[::2, 0:2] = o3.items[::2, 0:2] + np.float32(0.1)
    o3.step(a)
    assert not np.array_equal(o3.items, o.items)
    mask = torch.zeros(n, dtype=torch.uint8, device='cuda'); mask[1::4] = 1
    ob = env.reset(mask); o.reset(mask.cpu().numpy())   # a reset after the reseed: the new seed's reset streams
    assert np.array_equal(be.state.cpu().numpy(), o.state) and np.array_equal(be.items.cpu().numpy(), o.items) and np.array_equal(ob.cpu().numpy(), o.obs)
    env.close()
    # AntMazeBulletEnv keeps its own `rs` for the target draw (ant_maze_bullet_env.py:48,99-102,110): same constant here
    mz = H.AntMazeBulletEnv(num_envs=32, seed=1)
    om = orc.OracleEnv(orc.default_config(K.HRL_ANT_MAZE, num_envs=32, seed=1, auto_reset=1, max_episode_steps=2000), np.float32)
    mz.reset(); om.reset()
    for t in range(3):
        a = rng.uniform(-1, 1, (32, 8)).astype(np.float32)
        mz.step(torch.from_numpy(a).cuda()); om.step(a)
    mz.seed(11); om.cfg.seed = 11
    assert np.array_equal(mz._backend().state.cpu().numpy(), om.state)
    ob = mz.reset(); om.reset()
    assert np.array_equal(mz._backend().aux.cpu().numpy(), om.aux) and np.array_equal(ob.cpu().numpy(), om.obs)
    # the one-env object of the README loop: seed() between steps keeps the pinned host buffers too
    e1 = H.AntGatherBulletEnv(seed=2)
    e1.reset()
    e1.step(np.zeros(8)); hb = e1._backend()._host
    e1.seed(5)
    ob1, _, _, _ = e1.step(np.zeros(8))
    assert e1._backend()._host is hb and e1._cfg.seed == 5 and np.all(np.isfinite(ob1))
    e1.close()


def test_flagrun_path_reward_switched_on_mid_episode_is_finite_and_matches_the_oracle():
    """`env.set_reward_weights(path_rew_weight=...)` / `AntFlagrunBulletEnv.path_rew_weight = ...` on a running env built without a path
    reward paid +-inf until the next goal (x / 0: the record of set_target()'s bookkeeping was only kept when the weight was on).  Every flagrun env
    keeps it now (ant_flagrun_env.py:98-103 does, whatever the weights are)."""
    import hrl_pybullet_envs_amd as H
    n = 96
    env = H.AntFlagrunBulletEnv(timeout=30, num_envs=n, seed=8)
    o = orc.OracleEnv(orc.default_config(K.HRL_ANT_FLAGRUN, num_envs=n, seed=8, auto_reset=1, flag_timeout=30, max_episode_steps=2000), np.float32)
    env.reset(); o.reset()
    be = env._backend()
    assert np.array_equal(be.items.cpu().numpy(), o.items) and np.all(o.items[:, K.HRL_FLAG_SQDIST_OFF] > 0)
    rng = np.random.RandomState(2)
    try:
        for t in range(60):
            if t == 17:
                env.set_reward_weights(path_rew_weight=0.5); o.cfg.flag_path_rew_weight = 0.5
            if t == 40:   # ... and through the class attribute, as the reference's users do
                H.AntFlagrunBulletEnv.path_rew_weight = 1.5; o.cfg.flag_path_rew_weight = 1.5
            a = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
            ob, r, d, info = env.step(torch.from_numpy(a).cuda()); o.step(a)
            rr = r.cpu().numpy()
            assert np.all(np.isfinite(rr)), (t, rr)
            assert np.array_equal(rr, o.rew) and np.array_equal(be.items.cpu().numpy(), o.items) and np.array_equal(be.state.cpu().numpy(), o.state), t
    finally:
        H.AntFlagrunBulletEnv.path_rew_weight = 0
    assert np.abs(o.info[:, 2]).max() < 1e6   # episode returns were not poisoned
    env.close()


def test_update_config_refusals_leave_host_and_device_configs_equal():
    """hrl_update_config: a NULL cfg and changes the records' meaning depends on are HRL_ERR_BAD_ARG with nothing changed; the Python classes
    change a COPY of their config and commit it only when the library took it."""
    import ctypes as C
    import hrl_pybullet_envs_amd as H
    from hrl_pybullet_envs_amd import _lib
    env = H.AntGatherBulletEnv(num_envs=16, seed=1)
    env.reset()
    be = env._backend()
    L = _lib.lib()
    assert L.hrl_update_config(be._h, None, None) == K.HRL_ERR_BAD_ARG and b'null config' in L.hrl_last_error()
    before = bytes(env._cfg)
    with pytest.raises(_lib.HrlError, match='n_food'):
        env._change_config(n_food=7, n_poison=9)   # same stride, other slots
    with pytest.raises(_lib.HrlError):
        env._change_config(n_bins=11)              # another observation width
    with pytest.raises(_lib.HrlError):
        env._change_config(sensor_range=-1.0)      # an invalid config
    assert bytes(env._cfg) == before and bytes(be.cfg) == before
    o = orc.OracleEnv(orc.default_config(K.HRL_ANT_GATHER, num_envs=16, seed=1, auto_reset=1, max_episode_steps=2000), np.float32)
    o.reset()
    a = np.zeros((16, 8), np.float32)
    ob, r, d, info = env.step(torch.from_numpy(a).cuda()); o.step(a)
    assert np.array_equal(ob.cpu().numpy(), o.obs)   # still the config it was built with
    env.close()
    fl = H.AntFlagrunBulletEnv(num_envs=8, seed=1)
    fl.reset()
    with pytest.raises(_lib.HrlError, match='goal mode'):
        fl._change_config(flag_max_targets=0, flag_max_target_dist=3.0)   # the shared list -> goals near the robot: the same record read differently
    fl.close()
