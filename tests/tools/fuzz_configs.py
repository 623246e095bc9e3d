#!/usr/bin/env python3
"""Parity over RANDOM LEGAL CONFIGS: every constructor argument of the reference classes and every engine parameter of `hrl_model` drawn at random
(biased towards the edges of the capacity ranges: 0 / 1 / 16 / 17 / 48 / 49 / 64 items, 1 / 16 / 17 / 64 bins, observations 64 / 65 / 128 / 129 wide,
1 / 8 / 9 / 64 targets, env counts that leave parked waves in the last group), free-running with random and saturated actions, the wave phases (HIP
kernels on a GPU box, `gpu`; the lock-step host executor of tests/emu, `emu`) against the fp32 oracle, every output compared bit for bit (NaNs as
equal) after every step.  Between steps, at random: robots teleported next to items / walls / targets with random headings and velocities (the
same state pushed into both sides), NaN / inf / 1e20 / denormals written into states and items, masked resets, and for AntFlagrun with manual
goals `set_goals` / `next_target` on random masks.
The hand-picked CONFIG_MATRIX of the parity tests covers the branches; this covers their combinations.

    python tests/tools/fuzz_configs.py [runs] [gpu|emu] [first_seed] [steps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import orc  # noqa: E402
from hrl_pybullet_envs_amd import _capi as K  # noqa: E402

KINDS = [K.HRL_ANT_FLAT, K.HRL_ANT_GATHER, K.HRL_ANT_MAZE, K.HRL_POINT_GATHER, K.HRL_ANT_MAZE_MJ, K.HRL_ANT_FLAGRUN]
PI, TWO_PI = 3.14159265358979323846, 6.28318530717958647692
EDGE_ITEMS = [0, 1, 2, 7, 8, 15, 16, 17, 24, 31, 32, 33, 47, 48, 49, 56, 63, 64]
EDGE_BINS = [1, 2, 3, 5, 8, 10, 15, 16, 17, 19, 20, 31, 32, 33, 51, 63, 64]
ABSURD = [np.nan, np.inf, -np.inf, 1e20, -1e20, 3e38, -3e38, 1e-40, -1e-40, -0.0, 0.0, 1e-20, 1e10, 5.0, -5.0, 100.0]


def pick(rng, xs):
    return xs[rng.randint(len(xs))]


def f32(x):
    return float(np.float32(x))


def draw_model(rng, kw, kind):
    """engine parameters: the defaults most of the time, each one moved now and then"""
    if rng.rand() < 0.5:
        return
    if rng.rand() < 0.4: kw['model_solver_iters'] = int(pick(rng, [1, 2, 3, 5, 8, 13]))
    if rng.rand() < 0.4: kw['model_frame_skip'] = int(pick(rng, [1, 2, 3, 4, 5, 7]))
    if rng.rand() < 0.3: kw['model_timestep'] = f32(rng.uniform(0.001, 0.008))
    if rng.rand() < 0.3: kw['model_gravity'] = f32(pick(rng, [0.0, 1.6, 9.8, 9.81, 20.0]))
    if rng.rand() < 0.3: kw['model_contact_erp'] = f32(rng.uniform(0.0, 1.0))
    if rng.rand() < 0.3: kw['model_limit_erp'] = f32(rng.uniform(0.0, 1.0))
    if rng.rand() < 0.3: kw['model_friction_ground'] = f32(pick(rng, [0.0, 0.3, 0.8, 1.0, 3.0]))
    if rng.rand() < 0.3: kw['model_friction_robot'] = f32(pick(rng, [0.0, 0.1, 1.0, 1.5, 4.0]))
    if rng.rand() < 0.3: kw['model_contact_dist'] = f32(pick(rng, [0.0, 0.005, 0.02, 0.08]))
    if rng.rand() < 0.3: kw['model_limit_margin'] = f32(pick(rng, [0.0, 0.05, 0.25, 1.0]))
    if rng.rand() < 0.3: kw['model_max_joint_vel'] = f32(pick(rng, [5.0, 30.0, 100.0, 1000.0]))
    if rng.rand() < 0.3: kw['model_limit_max_impulse'] = f32(pick(rng, [0.5, 10.0, 100.0, 1e6]))
    if rng.rand() < 0.3: kw['model_torque_scale'] = f32(pick(rng, [0.0, 50.0, 250.0, 900.0]))
    if rng.rand() < 0.3: kw['model_density'] = f32(pick(rng, [5.0, 200.0, 1000.0, 3000.0]))
    if rng.rand() < 0.3: kw['model_point_force'] = f32(pick(rng, [0.0, 100.0, 500.0, 2500.0]))
    if rng.rand() < 0.3: kw['model_ground_z'] = f32(pick(rng, [0.0, 0.005, 0.05]))
    if rng.rand() < 0.2: kw['model_linear_damping'] = f32(pick(rng, [0.0, 0.04, 1.0, 300.0]))
    if rng.rand() < 0.2: kw['model_angular_damping'] = f32(pick(rng, [0.0, 0.04, 2.0, 300.0]))
    if rng.rand() < 0.2: kw['model_restitution'] = f32(pick(rng, [0.0, 0.25, 0.5, 1.0])); kw['model_restitution_threshold'] = f32(pick(rng, [0.0, 0.2, 1.0]))
    if rng.rand() < 0.2: kw['model_max_contacts'] = int(pick(rng, [1, 2, 4, 7, 11, 12]))
    if rng.rand() < 0.2: kw['model_joint_damping'] = f32(pick(rng, [0.0, 1.0, 30.0])); kw['model_joint_armature'] = f32(pick(rng, [0.0, 1.0, 0.01]))
    if kind != K.HRL_POINT_GATHER and rng.rand() < 0.3: kw['model_self_collision'] = int(rng.randint(2))
    if rng.rand() < 0.15: kw['model_step_group'] = 1


def draw_sensor(rng, kw, need_two_bins):
    kw['sensor_range'] = f32(pick(rng, [0.5, 2.0, 5.0, 9.0, 20.0, 60.0]))
    kw['sensor_span'] = pick(rng, [PI, TWO_PI, f32(rng.uniform(0.2, 7.0)), f32(PI), f32(TWO_PI)])
    nb = int(pick(rng, EDGE_BINS))
    if need_two_bins and nb < 2 and kw['sensor_span'] != TWO_PI:
        nb = 2
    kw['n_bins'] = nb


def draw_config(rng, kind):
    kw = dict(num_envs=int(pick(rng, [1, 2, 3, 4, 5, 13, 31, 48, 61])), seed=int(rng.randint(0, 2**31)) * int(rng.randint(1, 2**31)),
              auto_reset=int(rng.rand() < 0.7), max_episode_steps=int(pick(rng, [0, 1, 2, 5, 11, 40, 2000])),
              env_id_offset=int(pick(rng, [0, 1, 4093, 2**31 - 7, 2**32 + 5, 2**40 + 123])))
    gather = kind in (K.HRL_ANT_GATHER, K.HRL_POINT_GATHER)
    if gather:
        total = int(pick(rng, EDGE_ITEMS))
        nf = int(rng.randint(0, total + 1)) if rng.rand() < 0.8 else int(pick(rng, [0, total]))
        kw.update(n_food=nf, n_poison=total - nf)
        draw_sensor(rng, kw, False)
        kw['use_sensor'] = int(rng.rand() < 0.7)
        kw['respawn'] = int(rng.rand() < 0.7)
        wx, wy = f32(rng.uniform(3.0, 30.0)), f32(rng.uniform(3.0, 30.0))
        if rng.rand() < 0.5: wx, wy = 15.0, 15.0
        kw['world_size'] = (wx, wy)
        kw['robot_object_spacing'] = f32(rng.uniform(0.1, min(wx, wy) / 3.2))   # the respawn rejection loop must be able to end (gather_scene.py:42-50)
        kw['robot_coll_dist'] = f32(pick(rng, [1.0, 1.0, 0.3, 2.5, 6.0, 0.0, -1.0]))
        kw['model_item_collision'] = 1 if kw['robot_coll_dist'] <= 0 else int(rng.rand() < 0.7)
        kw['dying_cost'] = f32(pick(rng, [-10.0, 0.0, -1.5, 3.0]))
        if rng.rand() < 0.3: kw['centroid_n_static'] = int(rng.randint(0, 5)); kw['centroid_static_sum'] = (f32(rng.uniform(-9, 9)), f32(rng.uniform(-9, 9)))
    if kind in (K.HRL_ANT_MAZE, K.HRL_ANT_MAZE_MJ):
        mj = kind == K.HRL_ANT_MAZE_MJ
        walls = mj or rng.rand() < 0.7
        draw_sensor(rng, kw, walls)
        if not mj:
            kw['sense_walls'] = int(walls)
            kw['sense_target'] = int(rng.rand() < 0.5)
            kw['target_encoding'] = int(rng.rand() < 0.4)
        else:
            kw['inner_rew_weight'] = f32(pick(rng, [0.0, 0.0, 0.5, 1.0, -2.0]))
        nt = int(pick(rng, [1, 2, 4, 5, 8, 9, 16, 33, 63, 64]))
        kw['targets'] = [(f32(rng.uniform(-4.5, 4.5)), f32(rng.uniform(-8.5, 8.5))) for _ in range(nt)]
        kw['done_at_target'] = int(rng.rand() < 0.6)
        kw['max_steps'] = int(pick(rng, [-1, -1, 0, 1, 2, 4, 25]))
        kw['targ_dist_rew'] = int(rng.rand() < 0.4)
        if rng.rand() < 0.3:   # upstream WalkerBaseBulletEnv's class-level cost weights (0 after any flagrun reset in the reference's process)
            kw['walker_electricity_cost'] = f32(pick(rng, [0.0, -2.0, -0.5])); kw['walker_stall_torque_cost'] = f32(pick(rng, [0.0, -0.1]))
            kw['walker_joints_at_limit_cost'] = f32(pick(rng, [0.0, -0.1, -1.0]))
        kw['tol'] = f32(pick(rng, [1.5, 0.2, 0.8, 3.0, 12.0]))
        if rng.rand() < 0.4: kw['start_pos'] = (f32(rng.uniform(-4, 4)), f32(rng.uniform(-8, -3)), f32(pick(rng, [0.25, 0.4, 0.75])))
        if rng.rand() < 0.3: kw['centroid_n_static'] = int(rng.randint(0, 5)); kw['centroid_static_sum'] = (f32(rng.uniform(-9, 9)), f32(rng.uniform(-9, 9)))
    if kind == K.HRL_ANT_FLAGRUN:
        kw['use_sensor'] = int(rng.rand() < 0.5)
        draw_sensor(rng, kw, bool(kw['use_sensor']))
        size = f32(pick(rng, [10.0, 3.0, 1.5, 6.0, 25.0]))
        kw['flag_size'] = size
        kw['world_size'] = (size + 2.0, size + 2.0)                        # ant_flagrun_env.py:59-61
        kw['centroid_static_sum'] = (-(size + 2.0) / 2.0, 0.0)
        kw['tol'] = f32(pick(rng, [0.5, 0.2, 1.0, 2.0]))
        if rng.rand() < 0.6:   # a goal list
            kw['flag_max_targets'] = int(pick(rng, [1, 2, 3, 7, 100, 65535])); kw['flag_max_target_dist'] = 0.0
        else:                  # goals near the robot
            kw['flag_max_targets'] = int(pick(rng, [0, -1])); kw['flag_max_target_dist'] = f32(2.0 * kw['tol'] + rng.uniform(0.1, 6.0))
        kw['flag_timeout'] = int(pick(rng, [200, 1, 2, 5, 17, 32767, 0, -3]))
        kw['flag_switch_on_collision'] = int(rng.rand() < 0.7)
        kw['flag_enclosed'] = int(rng.rand() < 0.7)
        if not kw['flag_enclosed'] and not kw['use_sensor']:
            kw['centroid_n_static'] = 1; kw['centroid_static_sum'] = (0.0, 0.0)
        if rng.rand() < 0.3:   # upstream WalkerBaseBulletEnv's cost weights set back after reset() zeroed them
            kw['walker_electricity_cost'] = f32(pick(rng, [0.0, -2.0, -0.5])); kw['walker_stall_torque_cost'] = f32(pick(rng, [0.0, -0.1]))
            kw['walker_joints_at_limit_cost'] = f32(pick(rng, [0.0, -0.1, -1.0]))
        if rng.rand() < 0.4:   # the class-level reward weights (ant_flagrun_env.py:157-160)
            kw['flag_ant_env_rew_weight'] = f32(pick(rng, [1.0, 0.0, 0.5, -2.0]))
            kw['flag_path_rew_weight'] = f32(pick(rng, [0.0, 0.0, 1.0, 0.3, -4.0]))
            kw['flag_dist_rew_weight'] = f32(pick(rng, [0.0, 1.0, 0.05]))
            kw['flag_goal_reach_rew'] = f32(pick(rng, [5000.0, 0.0, 10.0, -1.0]))
        if rng.rand() < 0.35:
            kw['flag_manual_goals'] = 1; kw['flag_goal_capacity'] = int(pick(rng, [1, 2, 13, 14, 15, 16, 17, 29, 30, 40, 61]))
    if kind == K.HRL_ANT_FLAT and rng.rand() < 0.5:
        kw['walk_target'] = (f32(rng.uniform(-50, 1000)), f32(rng.uniform(-50, 50)))
    draw_model(rng, kw, kind)
    return kw


def make_cfg(kind, kw):
    cfg = orc.default_config(kind)
    for k, v in kw.items():
        if k.startswith('model_'): setattr(cfg.model, k[6:], v)
        elif k in ('world_size', 'start_pos', 'walk_target', 'centroid_static_sum'):
            arr = getattr(cfg, k)
            for i, x in enumerate(v): arr[i] = x
        elif k == 'targets':
            cfg.n_targets = len(v)
            for i, t in enumerate(v): cfg.targets[i][0], cfg.targets[i][1] = t
        else: setattr(cfg, k, v)
    return cfg


def clone(cfg):
    return K.hrl_config.from_buffer_copy(bytes(cfg))


class GpuSide:
    def __init__(self, cfg):
        import torch
        from hrl_pybullet_envs_amd.vec_env import BatchedEnv
        self.t, self.g = torch, BatchedEnv(cfg, 'cuda:0')
        self.g.count_solver_rows()

    def reset(self, mask=None): self.g.reset(None if mask is None else self.t.from_numpy(mask).cuda())
    def step(self, a): self.g.step(self.t.from_numpy(a).cuda())

    def push(self, o):
        t, g = self.t, self.g
        g.state.copy_(t.from_numpy(o.state)); g.items.copy_(t.from_numpy(o.items)); g.aux.copy_(t.from_numpy(o.aux))

    def observe(self, mask): self.g.observe(None if mask is None else self.t.from_numpy(mask).cuda())
    def set_goals(self, goals, mask): self.g.set_goals(self.t.from_numpy(goals).cuda(), None if mask is None else self.t.from_numpy(mask).cuda())
    def next_target(self, mask): return self.g.next_target(None if mask is None else self.t.from_numpy(mask).cuda())[1].cpu().numpy()

    def outputs(self):
        g = self.g
        d = dict(state=g.state, items=g.items, aux=g.aux, obs=g.obs, rew=g.reward, done=g.done, info=g.info, final_obs=g.final_obs, truncated=g.truncated)
        if g.cfg.env_kind == K.HRL_ANT_FLAGRUN: d['goal'] = g.goal
        d['solver_rows'] = g.solver_rows
        return {k: v.cpu().numpy() for k, v in d.items()}

    def close(self): self.g.close()


class EmuSide:
    ASAN = bool(os.environ.get('HRL_EMU_ASAN'))   # the AddressSanitizer + UBSan build of the executor (tests/test_emu_asan.py preloads libasan)

    def __init__(self, cfg):
        import emu_env
        self.e = emu_env.EmuEnv(cfg, asan=self.ASAN)
        msg = emu_env.lib(self.ASAN).emu_validate(orc.C.byref(cfg))
        if msg: raise ValueError(msg.decode())

    def reset(self, mask=None): self.e.reset(mask)
    def step(self, a): self.e.step(a)

    def push(self, o):
        e = self.e
        e.state[...] = o.state; e.items[...] = o.items; e.aux[...] = o.aux

    def observe(self, mask): self.e.observe(mask)

    def set_goals(self, goals, mask):
        import emu_env
        assert emu_env.lib(self.ASAN).emu_set_goals(orc.C.byref(self.e.cfg), orc.C.byref(self.e._bufs()), orc.ptr(goals), goals.shape[1], orc.ptr(mask), 0) == 0

    def next_target(self, mask):
        import emu_env
        ok = np.ones(self.e.N, np.uint8)
        assert emu_env.lib(self.ASAN).emu_next_target(orc.C.byref(self.e.cfg), orc.C.byref(self.e._bufs()), orc.ptr(mask), orc.ptr(ok), 0) == 0
        return ok

    def outputs(self):
        e = self.e
        return dict(state=e.state, items=e.items, aux=e.aux, obs=e.obs, rew=e.rew, done=e.done, info=e.info, final_obs=e.final_obs, truncated=e.truncated, goal=e.goal, solver_rows=e.solver_rows)

    def close(self): pass


def run(Side, kind, seed, T):
    rng = np.random.RandomState(seed)
    kw = draw_config(rng, kind)
    cfg = make_cfg(kind, kw)
    n = cfg.num_envs
    o = orc.OracleEnv(clone(cfg), np.float32)
    t0 = time.time()
    try:
        s = Side(clone(cfg))
    except Exception as e:  # a config the draw meant to be legal and the library refuses is a finding too
        return f'kind {kind} seed {seed} {kw}: refused: {e}', 0
    o.reset(); s.reset()
    CLOCK['create'] += time.time() - t0
    ended = 0
    names = ('state', 'items', 'aux', 'obs', 'rew', 'done', 'info', 'final_obs', 'truncated') + (('goal',) if kind == K.HRL_ANT_FLAGRUN else ()) + ('solver_rows',)
    manual = kind == K.HRL_ANT_FLAGRUN and cfg.flag_manual_goals and cfg.flag_max_targets >= 1   # goals near the robot ignore the list: hrl_set_goals refuses
    L = orc.lib()

    def some(p=0.5):
        return None if rng.rand() < 0.3 else np.ascontiguousarray(rng.rand(n) < p, np.uint8)

    def give_goals(mask):
        G = int(rng.randint(1, cfg.flag_goal_capacity + 1))
        goals = rng.uniform(-cfg.flag_size / 2, cfg.flag_size / 2, (n, G, 2)).astype(np.float32)
        L.orc_set_goals_batch_f32(orc.C.byref(o.cfg), orc.ptr(o.state), orc.ptr(o.items), orc.ptr(o.aux), orc.ptr(goals), G, orc.ptr(mask), orc.ptr(o.obs))
        s.set_goals(goals, mask)

    if manual: give_goals(None)
    for t in range(-1, T):
        what = rng.randint(12) if t >= 0 else -1
        if what == 0:      # robots teleported: next to an item, a wall, a target, or anywhere; any heading; moving
            rows = np.nonzero(rng.rand(n) < 0.5)[0]
            hx, hy = (5.0, 9.0) if kind in (K.HRL_ANT_MAZE, K.HRL_ANT_MAZE_MJ) else (cfg.world_size[0] / 2, cfg.world_size[1] / 2)
            for r in rows:
                where = rng.randint(4)
                xy = rng.uniform(-1, 1, 2) * (hx, hy)
                if where == 0 and kind in (K.HRL_ANT_GATHER, K.HRL_POINT_GATHER) and cfg.n_food + cfg.n_poison > 0:
                    k = rng.randint(cfg.n_food + cfg.n_poison); xy = o.items[r, 2 * k:2 * k + 2] + rng.uniform(-1.2, 1.2, 2)
                if where == 1: ax = rng.randint(2); sg = rng.choice([-1, 1]); xy[ax] = sg * ((hx, hy)[ax] - rng.uniform(-0.2, 0.9))
                if where == 2 and kind in (K.HRL_ANT_MAZE, K.HRL_ANT_MAZE_MJ): k = rng.randint(cfg.n_targets); xy = np.array(cfg.targets[k][:]) + rng.uniform(-1, 1, 2) * cfg.tol
                if where == 2 and kind == K.HRL_ANT_FLAGRUN: xy = o.items[r, 0:2] + rng.uniform(-1, 1, 2) * 1.5 * cfg.tol
                yaw = rng.uniform(-np.pi, np.pi)
                o.state[r, 0:2] = xy; o.state[r, 2] = pick(rng, [0.35, 0.5, 0.75, 1.2]); o.state[r, 3:7] = [0, 0, np.sin(yaw / 2), np.cos(yaw / 2)]
                o.state[r, 15:21] = rng.uniform(-2, 2, 6)
            s.push(o)
        if what == 4:      # absurd values in running envs (what a simulation that blows up can reach; tests/tools/fuzz_parity.py does this on the default configs)
            nq, nv = (7, 6) if kind == K.HRL_POINT_GATHER else (15, 14)
            vmax = float(cfg.model.max_joint_vel)
            for r in rng.permutation(n)[:max(1, n // 3)]:
                field, v = rng.randint(0, 3), np.float32(ABSURD[rng.randint(len(ABSURD))])
                if field == 0:    # a position, a quaternion component (unit or NaN), a joint angle (moderate or NaN)
                    idx = rng.randint(0, nq)
                    if 3 <= idx < 7: v = np.float32(np.nan)
                    if idx >= 7: v = np.float32(rng.choice([np.nan, 2000.0, -1500.0, 3.0, -0.0]))
                    o.state[r, idx] = v
                elif field == 1:  # a velocity (joint rates leave a step clamped to the model's bound)
                    idx = 15 + rng.randint(0, nv)
                    if idx >= 21 and np.isfinite(v) and abs(v) > vmax: v = np.float32(vmax * np.sign(v))
                    o.state[r, idx] = v
                elif kind in (K.HRL_ANT_GATHER, K.HRL_POINT_GATHER) and cfg.n_food + cfg.n_poison > 0:
                    o.items[r, rng.randint(0, 2 * (cfg.n_food + cfg.n_poison))] = v
            s.push(o)
        if what == 5:      # the observation recomputed from the records as they stand (hrl_observe)
            m = some(0.6)
            o.observe(m); s.observe(m)
        if what == 1:      # masked reset from outside
            m = some(0.3)
            o.reset(m); s.reset(m)
        if what == 2 and manual: give_goals(some())
        if what == 3 and kind == K.HRL_ANT_FLAGRUN:
            m, ok = some(), np.ones(n, np.uint8)
            L.orc_next_target_batch_f32(orc.C.byref(o.cfg), orc.ptr(o.state), orc.ptr(o.items), orc.ptr(o.aux), orc.ptr(m), orc.ptr(o.obs), orc.ptr(ok))
            if not np.array_equal(ok, s.next_target(m)):
                s.close()
                return f'kind {kind} seed {seed} step {t}: next_target ok flags differ; config {kw}', ended
        if t >= 0:
            mode = rng.randint(4)
            a = rng.uniform(-1, 1, (n, o.ad)).astype(np.float32)
            if mode == 1: a = np.sign(a).astype(np.float32)            # saturated
            if mode == 2: a *= np.float32(3.0)                         # beyond the clip
            if mode == 3 and rng.rand() < 0.3: a[rng.randint(n), rng.randint(o.ad)] = np.float32(ABSURD[rng.randint(len(ABSURD))])   # NaN / inf / 1e20 torques
            t0 = time.time(); s.step(a); t1 = time.time(); o.step(a); t2 = time.time()
            CLOCK['side steps'] += t1 - t0; CLOCK['oracle steps'] += t2 - t1
            ended += int(o.done.sum())
        t0 = time.time()
        out = s.outputs()
        CLOCK['read back'] += time.time() - t0
        for name in names if t >= 0 else ('state', 'items', 'aux', 'obs'):
            A, B = getattr(o, name).reshape(n, -1), out[name].reshape(n, -1)
            ok = (A == B) | ((A != A) & (B != B))
            if not ok.all():
                e = int(np.where(~ok.all(1))[0][0])
                s.close()
                return (f'kind {kind} seed {seed} step {t}: {name} differs for env {e} at {np.where(~ok[e])[0][:8]}; oracle {A[e][~ok[e]][:6]} '
                        f'other {B[e][~ok[e]][:6]}; config {kw}'), ended
    s.close()
    return None, ended


CLOCK = {'create': 0.0, 'side steps': 0.0, 'oracle steps': 0.0, 'read back': 0.0}


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    Side = GpuSide if (len(sys.argv) > 2 and sys.argv[2] == 'gpu') else EmuSide
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    T = int(sys.argv[4]) if len(sys.argv) > 4 else 40
    fails, ended, count = 0, 0, 0
    for seed in range(first, first + runs):
        for kind in KINDS:
            r, e = run(Side, kind, seed * 16 + kind, T)
            ended += e; count += 1
            if r:
                fails += 1
                print('FAIL', r, flush=True)
        if (seed - first) % 10 == 9:
            print(f'  ... {count} configs, {fails} with a difference', flush=True)
    print('seconds: ' + ', '.join(f'{k} {v:.1f}' for k, v in CLOCK.items()))
    print(f'{Side.__name__}: {count} random configs (seeds {first}..{first + runs - 1} x 6 kinds) x {T} steps, {ended} episode ends, {fails} with a difference')
    return 1 if fails else 0


if __name__ == '__main__':
    sys.exit(main())
