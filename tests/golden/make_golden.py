#!/usr/bin/env python3
"""Generate golden vectors from the reference's own in-tree Python task logic.

Runs ONLY in the build container (needs /root/reference). Writes small JSON fixtures
(inputs + expected outputs, float64) next to this file; the fixtures travel to the GPU
box, the reference does not.

The reference's physics lives in third-party packages that are absent here (pybullet,
pybullet_envs, pybulletgym, gym -> ModuleNotFoundError).  Those modules are replaced in
sys.modules by EMPTY stand-ins (base classes with no behaviour) so that the in-tree
modules import; every function called below is in-tree reference code, called on
duck-typed `self` objects that carry only the attributes the function reads.  Nothing
from the stand-ins contributes arithmetic to a fixture (the one exception is
pybullet.getQuaternionFromEuler, used by the reference only as a class constant that the
fixtures never read).

Usage:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden.py
"""
import json
import math
import os
import sys
import types
import warnings

import numpy as np

OUT_DIR = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


# ----------------------------------------------------------------------------------------------
# stand-ins for the absent third-party packages
# ----------------------------------------------------------------------------------------------
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Empty:
    def __init__(self, *a, **k):
        pass


REGISTERED = []   # what hrl_pybullet_envs/__init__.py hands to gym.envs.register


def install_stubs():
    class Box(_Empty):
        def __init__(self, low=None, high=None, shape=None, **k):
            self.low, self.high, self.shape = low, high, shape

    class Scene(_Empty):
        def episode_restart(self, bullet_client):
            pass

    class AntBulletEnv(_Empty):
        """Stand-in base: step() returns whatever the test parked on the instance.

        With `_bookkeeping` set on the instance (the reset-sequence fixtures only) reset()/step() restate, from memory of
        upstream WalkerBaseBulletEnv (unverified), the ORDER in which it touches `potential`:
        reset: robot back to its home pose -> calc_state -> potential = calc_potential();
        step: calc_state -> potential_old = potential; potential = calc_potential(); reward = alive(+1) + progress.
        The fixtures generated this way pin the order of operations of the IN-TREE reset()/next_target()/step() around
        those calls (which target a potential belongs to), not upstream arithmetic."""

        def reset(self):
            self.robot._pos = list(self.robot._home)
            s = self.robot.calc_state()
            self.potential = self.robot.calc_potential()
            return s

        def step(self, a):
            if not getattr(self, '_bookkeeping', False):
                return self._super_step_result
            s = self.robot.calc_state()
            potential_old = self.potential
            self.potential = self.robot.calc_potential()
            return s, 1.0 + float(self.potential - potential_old), False, {}

    class WalkerBaseMuJoCoEnv(_Empty):
        def HUD(self, *a):
            pass

    class WalkerBase(_Empty):
        def calc_state(self):
            return None

    _mod('pybullet', getQuaternionFromEuler=lambda e: (0.0, 0.0, 0.0, 1.0),
         COV_ENABLE_PLANAR_REFLECTION=0, GUI=1)
    _mod('pybullet_data', getDataPath=lambda: '/nonexistent')
    _mod('pybullet_envs')
    _mod('pybullet_envs.robot_bases', BodyPart=_Empty, MJCFBasedRobot=_Empty, Pose_Helper=_Empty)
    _mod('pybullet_envs.env_bases', MJCFBaseBulletEnv=_Empty)
    _mod('pybullet_envs.scene_abstract', Scene=Scene)
    _mod('pybullet_envs.gym_locomotion_envs', AntBulletEnv=AntBulletEnv, WalkerBaseBulletEnv=_Empty)
    for n in ('pybulletgym', 'pybulletgym.envs', 'pybulletgym.envs.mujoco', 'pybulletgym.envs.mujoco.envs',
              'pybulletgym.envs.mujoco.envs.locomotion', 'pybulletgym.envs.mujoco.robots',
              'pybulletgym.envs.mujoco.robots.locomotors'):
        _mod(n)
    _mod('pybulletgym.envs.mujoco.envs.locomotion.walker_base_env', WalkerBaseMuJoCoEnv=WalkerBaseMuJoCoEnv)
    _mod('pybulletgym.envs.mujoco.robots.locomotors.walker_base', WalkerBase=WalkerBase)
    _mod('pybulletgym.envs.mujoco.robots.robot_bases', MJCFBasedRobot=_Empty)
    _mod('pybulletgym.envs.mujoco.robots.locomotors.ant', Ant=_Empty)
    seeding = _mod('gym.utils.seeding', np_random=lambda s=None: (np.random.RandomState(s), s))
    utils = _mod('gym.utils', seeding=seeding)
    spaces = _mod('gym.spaces', Box=Box)
    envs = _mod('gym.envs', register=lambda **k: REGISTERED.append(k))
    _mod('gym', utils=utils, spaces=spaces, envs=envs)


def NS(**k):
    return types.SimpleNamespace(**k)


class Pose:
    def __init__(self, xyz, rpy):
        self._xyz, self._rpy = list(xyz), list(rpy)

    def xyz(self):
        return self._xyz

    def rpy(self):
        return self._rpy


class Body:
    """Duck-typed BodyPart: only what the in-tree code reads."""

    def __init__(self, xyz, rpy, speed=(0, 0, 0)):
        self._pose = Pose(xyz, rpy)
        self._speed = np.array(speed, dtype=float)

    def pose(self):
        return self._pose

    def get_pose(self):
        return list(self._pose.xyz()) + [0, 0, 0, 1]

    def get_position(self):
        return list(self._pose.xyz())

    def speed(self):
        return self._speed


class LoggingRS:
    """RandomState wrapper that records every uniform pair handed out by .rand(2)."""

    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)
        self.log = []

    def rand(self, n):
        v = self.rs.rand(n)
        self.log.append(v.tolist())
        return v

    def randint(self, lo, hi):
        return self.rs.randint(lo, hi)


class FakeClient:
    def __init__(self):
        self.next_id = 11

    def loadURDF(self, path, basePosition=None, *a, **k):
        i = self.next_id
        self.next_id += 1
        return i

    def configureDebugVisualizer(self, *a, **k):
        pass

    def resetBasePositionAndOrientation(self, *a, **k):
        pass

    def changeDynamics(self, *a, **k):
        pass


def tolist(x):
    if isinstance(x, np.ndarray):
        return x.tolist()
    if isinstance(x, (list, tuple)):
        return [tolist(v) for v in x]
    if isinstance(x, (np.floating, np.integer)):
        return x.item()
    return x


def main():
    warnings.simplefilter('ignore')
    install_stubs()
    sys.path.insert(0, REF)
    import hrl_pybullet_envs.envs.intersection_utils as iu
    from hrl_pybullet_envs.envs.sizeable_enclosed_scene import SizeableEnclosedScene
    from hrl_pybullet_envs.envs.ant_maze.maze_scene import MazeScene
    from hrl_pybullet_envs.envs.gather.gather_scene import GatherScene
    from hrl_pybullet_envs.envs.gather.ant_gather_env import AntGatherBulletEnv
    from hrl_pybullet_envs.envs.gather.gather_base import GatherBulletEnv
    from hrl_pybullet_envs.envs.gather.point_bot import PointBot
    from hrl_pybullet_envs.envs.ant_maze.ant_maze_bullet_env import AntMazeBulletEnv
    from hrl_pybullet_envs.envs.MjAnt import AntMjEnv, MjAnt
    from hrl_pybullet_envs.utils import PositionEncoding

    rng = np.random.RandomState(20261003)
    G = {}

    # ---------------------------------------------------------------- intersection_utils
    P = iu.Point
    cases = []
    fixed = [((0, 0), (1, 1), (0, 2), (2, 0)), ((0, 0), (1, 0), (0, 1), (1, 1)), ((0, 0), (0, 1), (3, -2), (3, 5)),
             ((1, 1), (2, 2), (3, 3), (4, 4)), ((-2, -5), (-2, 0), (5, 9), (-5, 9))]
    for c in fixed:
        cases.append([list(map(float, p)) for p in c])
    for _ in range(60):
        cases.append(rng.uniform(-10, 10, size=(4, 2)).round(3).tolist())
    inter = []
    for p1, p2, p3, p4 in cases:
        r = iu.inf_intersection(P(*p1), P(*p2), P(*p3), P(*p4))
        s = iu.segment_intersection(P(*p1), P(*p2), P(*p3), P(*p4))
        inter.append({'p': [p1, p2, p3, p4], 'inf': None if r is None else [r.x, r.y], 'seg': bool(s)})
    # collinear / touching segment special cases (reference's geeksforgeeks branches)
    seg_special = [((0, 0), (2, 0), (1, 0), (3, 0)), ((0, 0), (1, 0), (2, 0), (3, 0)), ((0, 0), (2, 2), (2, 2), (3, 0)),
                   ((0, 0), (2, 2), (1, 1), (1, 5)), ((0, 0), (0, 2), (0, 3), (0, 4)), ((0, 0), (4, 0), (2, 0), (2, 0))]
    for p1, p2, p3, p4 in seg_special:
        r = iu.inf_intersection(P(*p1), P(*p2), P(*p3), P(*p4))
        s = iu.segment_intersection(P(*p1), P(*p2), P(*p3), P(*p4))
        inter.append({'p': [list(map(float, p)) for p in (p1, p2, p3, p4)],
                      'inf': None if r is None else [r.x, r.y], 'seg': bool(s)})
    quad_pts = [(1, 1), (1, -1), (-1, 1), (-1, -1), (0, 0), (0, 1), (0, -1), (1, 0), (-1, 0)] + \
        rng.uniform(-3, 3, size=(20, 2)).tolist()
    quads = [{'p': [float(x), float(y)], 'q': iu.quadrant(P(x, y))} for x, y in quad_pts]
    pol = [{'rho': float(r), 'phi': float(ph), 'xy': list(map(float, iu.pol2cart(r, ph)))}
           for r, ph in rng.uniform(-7, 7, size=(10, 2))]
    G['intersection'] = {'lines': inter, 'quadrant': quads, 'pol2cart': pol}

    # ---------------------------------------------------------------- sense_walls
    maze = MazeScene(None, 9.8, 0.0165 / 4, 4)
    arena = SizeableEnclosedScene(None, 9.8, 0.0165 / 4, 4, (15, 15))

    def bounds_list(scene):
        return [[[a.x, a.y], [b.x, b.y]] for a, b in scene.bounds]

    sw = {'maze_bounds': bounds_list(maze), 'arena_bounds': bounds_list(arena),
          'maze_box_pos': list(maze.box_pos), 'cases': []}

    def sw_case(scene_name, scene, bins, span, rng_, pos, yaw):
        out = scene.sense_walls(bins, span, rng_, np.array(pos, dtype=float), yaw)
        sw['cases'].append({'scene': scene_name, 'bins': bins, 'span': span, 'range': rng_,
                            'pos': list(map(float, pos)), 'yaw': float(yaw), 'out': tolist(out)})

    sw_case('maze', maze, 10, 2 * np.pi, 5.0, (-2, -5), 0.0)
    sw_case('maze', maze, 10, 2 * np.pi, 5.0, (2, 0), 0.7)
    for _ in range(120):
        pos = (rng.uniform(-4.9, 4.9), rng.uniform(-8.9, 8.9))
        sw_case('maze', maze, 10, 2 * np.pi, 5.0, pos, rng.uniform(-np.pi, np.pi))
    for _ in range(30):
        pos = (rng.uniform(-4.9, 4.9), rng.uniform(-8.9, 8.9))
        sw_case('maze', maze, 8, np.pi, 4.0, pos, rng.uniform(-np.pi, np.pi))
    for _ in range(30):
        pos = (rng.uniform(-7.4, 7.4), rng.uniform(-7.4, 7.4))
        sw_case('arena', arena, 10, 2 * np.pi, 5.0, pos, rng.uniform(-np.pi, np.pi))
    G['sense_walls'] = sw

    # ---------------------------------------------------------------- food sensor / abs pos / sq dist
    def sensor_case(cls, n_bins, span, srange, n_food, n_poison, robot_xy, yaw, layout_rs):
        food = {11 + i: (layout_rs.rand(2) * 14 - 7).tolist() + [0.1] for i in range(n_food)}
        poison = {11 + n_food + i: (layout_rs.rand(2) * 14 - 7).tolist() + [0.1] for i in range(n_poison)}
        body = Body([robot_xy[0], robot_xy[1], 0.5], [0.01, -0.02, yaw])
        scene = NS(food=food, poison=poison, all_items={**food, **poison})
        if cls is AntGatherBulletEnv:
            self = NS(n_bins=n_bins, sensor_span=span, sensor_range=srange, robot_body=body, stadium_scene=scene,
                      FOOD='food', POISON='poison', debug=False, parts={'torso': body})
        else:
            self = NS(n_bins=n_bins, sensor_span=span, sensor_range=srange, robot=NS(robot_body=body),
                      stadium_scene=scene, FOOD='food', POISON='poison', debug=False)
        dists = {i: cls.sq_dist_robot(self, p) for i, p in scene.all_items.items()}
        fr, pr = cls.get_sensor_readings(self, dists)
        af, ap = cls.get_abs_pos(self, dists)
        return {'cls': cls.__name__, 'n_bins': n_bins, 'span': span, 'range': srange,
                'robot_xy': list(map(float, robot_xy)), 'yaw': float(yaw),
                'food': [food[k][:2] for k in food], 'poison': [poison[k][:2] for k in poison],
                'sq_dists': [dists[k] for k in scene.all_items], 'food_readings': tolist(fr),
                'poison_readings': tolist(pr), 'abs_food': tolist(af), 'abs_poison': tolist(ap)}

    sens = [sensor_case(AntGatherBulletEnv, 10, np.pi, 20., 8, 8, (0.3, -0.2), 0.4, np.random.RandomState(0))]
    for k in range(80):
        sens.append(sensor_case(AntGatherBulletEnv, 10, np.pi, 20., 8, 8, rng.uniform(-7, 7, 2),
                                rng.uniform(-np.pi, np.pi), np.random.RandomState(1000 + k)))
    for k in range(40):
        sens.append(sensor_case(GatherBulletEnv, 5, np.pi, 20., 8, 8, rng.uniform(-7, 7, 2),
                                rng.uniform(-np.pi, np.pi), np.random.RandomState(2000 + k)))
    for k in range(10):
        sens.append(sensor_case(AntGatherBulletEnv, 7, 2.0, 9.0, 5, 3, rng.uniform(-7, 7, 2),
                                rng.uniform(-np.pi, np.pi), np.random.RandomState(3000 + k)))
    G['food_sensor'] = sens

    # ---------------------------------------------------------------- GatherScene spawn / respawn
    def scene_run(seed, world, n_food, n_poison, spacing, respawn, hits):
        sc = GatherScene(None, 9.8, 0.0165 / 4, 4, world, n_food, n_poison, spacing, respawn)
        sc.rs = LoggingRS(seed)
        sc.loaded = True  # skip plane/wall loading (pybullet calls only)
        cl = FakeClient()
        sc._p = cl
        sc.episode_restart(cl)
        after_restart = {'food': [sc.food[k] for k in sc.food], 'poison': [sc.poison[k] for k in sc.poison]}
        n_draws_restart = len(sc.rs.log)
        events = []
        for obj_idx, agent in hits:
            ids = list(sc.food.keys()) + list(sc.poison.keys()) + [999]
            oid = ids[obj_idx]
            n0 = len(sc.rs.log)
            rew = sc.reward_collision(oid, list(agent))
            pos = sc.all_items.get(oid)
            events.append({'obj_index': obj_idx, 'agent_xyz': list(map(float, agent)), 'rew': rew,
                           'new_pos': None if pos is None else list(map(float, pos)),
                           'draws': sc.rs.log[n0:]})
        return {'seed': seed, 'world': list(world), 'n_food': n_food, 'n_poison': n_poison, 'spacing': spacing,
                'respawn': respawn, 'restart_draws': sc.rs.log[:n_draws_restart], 'after_restart': after_restart,
                'events': events}

    hits = [(0, (1, 1, .5)), (8, (1, 1, .5)), (16, (0, 0, .5)), (3, (6.5, 6.5, .5)), (12, (-6.9, 6.9, .5)),
            (0, (0.2, -0.1, .5)), (5, (-3, 2, .5))]
    G['gather_scene'] = [scene_run(123, (15, 15), 8, 8, 2.0, True, hits),
                         scene_run(7, (15, 15), 8, 8, 2.0, True, hits),
                         scene_run(99, (15, 15), 8, 8, 2.0, False, hits),
                         scene_run(5, (9, 11), 4, 6, 3.0, True, [(0, (0, 0, .5)), (4, (2, 2, .5)), (9, (-3, 1, .5))])]

    # ---------------------------------------------------------------- AntGather / Gather(Point) full step (task half)
    def gather_step_case(cls, k, force=None, n_food=8, n_poison=8, n_bins=None):
        lrs = np.random.RandomState(5000 + k)
        if n_bins is None:
            n_bins = 10 if cls is AntGatherBulletEnv else 5
        sc = GatherScene(None, 9.8, 0.0165 / 4, 4, (15, 15), n_food, n_poison, 2.0, True)
        sc.rs = LoggingRS(6000 + k)
        sc.loaded = True
        cl = FakeClient()
        sc._p = cl
        sc.episode_restart(cl)
        n_restart = len(sc.rs.log)
        items0 = [list(sc.all_items[i]) for i in sc.all_items]
        # robot placed near a random item so pickups happen often
        ids = list(sc.all_items.keys())
        if force == 'far':
            xy = np.array([0.0, 0.0])
        else:
            tgt = sc.all_items[ids[lrs.randint(0, n_food + n_poison)]]
            xy = np.array(tgt[:2]) + lrs.uniform(-0.9, 0.9, 2)
        z = 0.2 if force == 'dead' else lrs.uniform(0.3, 0.9)
        rpy = [lrs.uniform(-.3, .3), lrs.uniform(-.3, .3), lrs.uniform(-np.pi, np.pi)]
        body = Body([xy[0], xy[1], z], rpy)
        nstate = 28 if cls is AntGatherBulletEnv else 8
        st = lrs.uniform(-1, 1, nstate)
        initial_z = 0.75 if cls is AntGatherBulletEnv else 1.0
        st[0] = z - initial_z
        if force == 'nan':
            st[5] = np.nan
        st = st.astype(np.float32)
        if cls is AntGatherBulletEnv:
            robot = NS(apply_action=lambda a: None, calc_state=lambda: st.copy(), initial_z=initial_z,
                       body_rpy=rpy, alive_bonus=lambda zz, p: +1 if zz > 0.26 else -1)
            self = cls.__new__(cls)
            self.__dict__.update(dict(robot=robot, scene=NS(global_step=lambda: None), stadium_scene=sc,
                                      parts={'torso': body}, robot_body=body, robot_coll_dist=1, use_sensor=True,
                                      n_bins=n_bins, sensor_span=np.pi, sensor_range=20., dying_cost=-10,
                                      debug=False, _p=cl))
        else:
            robot = NS(apply_action=lambda a: None, calc_state=lambda: st.copy(), initial_z=initial_z,
                       robot_body=body, alive_bonus=lambda zz, p: 1)
            self = cls.__new__(cls)
            self.__dict__.update(dict(robot=robot, scene=NS(global_step=lambda: None), stadium_scene=sc,
                                      robot_coll_dist=1, use_sensor=True, n_bins=n_bins, sensor_span=np.pi,
                                      sensor_range=20., dying_cost=-10, debug=False, _p=cl))
        obs, rew, done, info = cls.step(self, np.zeros(8))
        return {'cls': cls.__name__, 'n_bins': n_bins, 'state_in': st.astype(float).tolist(), 'initial_z': initial_z,
                'torso_xyz': [float(xy[0]), float(xy[1]), float(z)], 'rpy': list(map(float, rpy)),
                'items_before': [p[:2] for p in items0], 'respawn_draws': sc.rs.log[n_restart:],
                'items_after': [list(sc.all_items[i])[:2] for i in sc.all_items],
                'obs': tolist(obs), 'rew': float(rew), 'done': bool(done), 'food_rew': float(info['food_rew']),
                'dead_rew': float(info['dead_rew'])}

    gs = []
    for k in range(60):
        gs.append(gather_step_case(AntGatherBulletEnv, k))
    gs.append(gather_step_case(AntGatherBulletEnv, 100, 'dead'))
    gs.append(gather_step_case(AntGatherBulletEnv, 101, 'nan'))
    gs.append(gather_step_case(AntGatherBulletEnv, 102, 'far'))
    for k in range(30):
        gs.append(gather_step_case(GatherBulletEnv, 200 + k))
    gs.append(gather_step_case(GatherBulletEnv, 300, 'nan'))
    G['gather_step'] = gs

    # ---------------------------------------------------------------- maze: target obs + full step (task half)
    def maze_case(k, encoding, sense_target=False, force_target=None, near=False, n_bins=10, targets=None):
        lrs = np.random.RandomState(7000 + k)
        xy = np.array([lrs.uniform(-4.5, 4.5), lrs.uniform(-8.5, 8.5)])
        rpy = [lrs.uniform(-.2, .2), lrs.uniform(-.2, .2), lrs.uniform(-np.pi, np.pi)]
        body = Body([xy[0], xy[1], 0.45], rpy)
        if targets is None:
            targets = ([2, -3], [2, 0], [2, 3], [-2, 4])
        target = np.array(targets[lrs.randint(0, len(targets))] if force_target is None else force_target)
        if near:  # put the robot next to its target -> sparse-reward / done branch
            xy = target + lrs.uniform(-1.0, 1.0, 2)
            body = Body([xy[0], xy[1], 0.45], rpy)
        ant_obs = lrs.uniform(-1, 1, 28).astype(np.float32)
        inner_rew = float(lrs.uniform(-1, 1))
        centroid = xy + lrs.uniform(-0.3, 0.3, 2)
        wtd = float(np.linalg.norm(target - centroid))
        self = AntMazeBulletEnv.__new__(AntMazeBulletEnv)
        self.__dict__.update(dict(
            n_bins=n_bins, sensor_range=5.0, sensor_span=2 * np.pi, targets=targets, sense_walls=True,
            sense_target=sense_target, done_at_target=True, max_steps=-1, t=int(lrs.randint(0, 50)), tol=1.5,
            inner_rew_weight=0, targ_dist_rew=False, target_encoding=PositionEncoding(encoding), target=target,
            debug=0, scene=maze, robot=NS(body_real_xyz=[xy[0], xy[1], 0.45], walk_target_dist=wtd),
            robot_body=body, _super_step_result=(ant_obs, inner_rew, False, {})))
        obs, rew, d, _ = AntMazeBulletEnv.step(self, np.zeros(8))
        tv = AntMazeBulletEnv.get_target_vec_obs(self)
        ts = AntMazeBulletEnv.get_target_sensor_obs(self)
        return {'encoding': encoding, 'sense_target': sense_target, 'torso_xy': xy.tolist(), 'rpy': list(map(float, rpy)),
                'target': target.tolist(), 'ant_obs': ant_obs.astype(float).tolist(), 'inner_rew': inner_rew,
                'walk_target_dist': wtd, 'obs': tolist(obs), 'rew': float(rew), 'done': bool(d),
                'target_vec_obs': tolist(tv), 'target_sensor_obs': tolist(ts)}

    mz = [maze_case(k, 0) for k in range(40)] + [maze_case(100 + k, 1) for k in range(15)] + \
        [maze_case(200 + k, 0, sense_target=True) for k in range(25)]
    # robot close to the target -> sparse reward branch
    for k in range(12):
        c = maze_case(300 + k, 0, near=True)
        mz.append(c)
    G['maze_step'] = mz
    # SURVEY Appendix D spot values
    body = Body([0.3, -0.2, 0.5], [0, 0, 0.4])
    s0 = NS(target=np.array([-2, 4]), robot_body=body, target_encoding=PositionEncoding.normed_vec)
    s1 = NS(target=np.array([-2, 4]), robot_body=body, target_encoding=PositionEncoding.angle)
    G['target_vec_spot'] = {'normed': tolist(AntMazeBulletEnv.get_target_vec_obs(s0)),
                            'angle': tolist(AntMazeBulletEnv.get_target_vec_obs(s1))}

    # ---------------------------------------------------------------- PointBot.calc_state
    pb = []
    spot = NS(robot_body=Body([0.3, -0.2, 0.6], [0.05, -0.1, 0.4], [1, 2, -0.5]), walk_target_x=0, walk_target_y=0,
              initial_z=1)
    pb.append({'xyz': [0.3, -0.2, 0.6], 'rpy': [0.05, -0.1, 0.4], 'speed': [1, 2, -0.5], 'target': [0, 0],
               'out': PointBot.calc_state(spot).astype(float).tolist()})
    for _ in range(30):
        xyz = rng.uniform(-7, 7, 3).tolist()
        rpy = rng.uniform(-np.pi, np.pi, 3).tolist()
        sp = rng.uniform(-5, 5, 3).tolist()
        s = NS(robot_body=Body(xyz, rpy, sp), walk_target_x=0, walk_target_y=0, initial_z=1)
        pb.append({'xyz': xyz, 'rpy': rpy, 'speed': sp, 'target': [0, 0],
                   'out': PointBot.calc_state(s).astype(float).tolist()})
    G['pointbot_state'] = pb

    # ---------------------------------------------------------------- AntMjEnv.step reward assembly (MjAnt.py:36-97)
    mj = []
    for k in range(30):
        lrs = np.random.RandomState(9000 + k)
        state = lrs.uniform(-1, 1, 29)
        state[2] = lrs.uniform(0.2, 0.8)
        if k == 29:
            state[10] = np.inf
        initial_z = 0.75
        pot_old = float(lrs.uniform(-70000, -50000))
        pot_new = pot_old + float(lrs.uniform(-30, 30))
        n_lim = int(lrs.randint(0, 5))
        robot = NS(apply_action=lambda a: None, calc_state=lambda: state.copy(), initial_z=initial_z,
                   body_rpy=[0, 0.1, 0], calc_potential=lambda: pot_new, feet=[], feet_contact=np.zeros(4),
                   joints_at_limit=n_lim)
        robot.alive_bonus = lambda z, pitch: MjAnt.alive_bonus(robot, z, pitch)
        self = AntMjEnv.__new__(AntMjEnv)
        self.__dict__.update(dict(robot=robot, scene=NS(global_step=lambda: None), potential=pot_old,
                                  joints_at_limit_cost=-0.1, ground_ids=set(), reward=0.0))
        obs, rew, done, _ = AntMjEnv.step(self, np.zeros(8))
        mj.append({'state': state.tolist(), 'initial_z': initial_z, 'potential_old': pot_old, 'potential_new': pot_new,
                   'joints_at_limit': n_lim, 'joints_at_limit_cost': -0.1, 'rew': float(rew), 'done': bool(done)})
    G['antmj_step'] = mj

    # ---------------------------------------------------------------- AntMazeMjEnv.step / _get_obs (ant_maze_mj_env.py:57-78)
    from hrl_pybullet_envs.envs.ant_maze.ant_maze_mj_env import AntMazeMjEnv
    def maze_mj_case(k, n_bins=10):
        lrs = np.random.RandomState(11000 + k)
        targets = ([2, -4], [2, 0], [2, 4], [0, 4], [-2, 4])
        target = np.array(targets[lrs.randint(0, 5)])
        xy = np.array([lrs.uniform(-4.5, 4.5), lrs.uniform(-8.5, 8.5)])
        if k >= 38:
            xy = target + lrs.uniform(-1.0, 1.0, 2)  # next to the target -> sparse reward / done
        state = lrs.uniform(-1, 1, 29)
        state[0], state[1], state[2] = xy[0], xy[1], lrs.uniform(0.2, 0.8)
        rpy = [lrs.uniform(-.2, .2), lrs.uniform(-.2, .2), lrs.uniform(-np.pi, np.pi)]
        pot_old = float(lrs.uniform(-700, -100)); pot_new = pot_old + float(lrs.uniform(-3, 3))
        n_lim = int(lrs.randint(0, 4)); t0 = int(lrs.randint(0, 500))
        centroid = xy + lrs.uniform(-0.3, 0.3, 2)
        wtd = float(np.linalg.norm(target - centroid))
        robot = NS(apply_action=lambda a: None, calc_state=lambda: state.copy(), initial_z=0.25, body_rpy=rpy,
                   calc_potential=lambda: pot_new, feet=[], feet_contact=np.zeros(4), joints_at_limit=n_lim,
                   walk_target_dist=wtd)
        robot.alive_bonus = lambda z, pitch: MjAnt.alive_bonus(robot, z, pitch)
        self = AntMazeMjEnv.__new__(AntMazeMjEnv)
        self.__dict__.update(dict(robot=robot, scene=maze, potential=pot_old, joints_at_limit_cost=-0.1, ground_ids=set(),
                                  reward=0.0, n_bins=n_bins, sensor_span=2 * np.pi, sensor_range=5.0, targets=targets,
                                  tol=1.5, inner_rew_weight=float(k % 3 == 0) * 0.5, t=t0, target=target, debug=0,
                                  robot_body=Body([xy[0], xy[1], state[2]], rpy)))
        maze.global_step = lambda: None
        obs, rew, d, _ = AntMazeMjEnv.step(self, np.zeros(8))
        return {'state': state.tolist(), 'rpy': list(map(float, rpy)), 'target': target.tolist(), 't_before': t0,
                'potential_old': pot_old, 'potential_new': pot_new, 'joints_at_limit': n_lim,
                'walk_target_dist': wtd, 'inner_rew_weight': float(self.inner_rew_weight), 'obs': tolist(obs),
                'rew': float(rew), 'done': bool(d), 't_after': int(self.t)}

    G['maze_mj_step'] = [maze_mj_case(k) for k in range(50)]

    # ---------------------------------------------------------------- AntFlagrunBulletEnv.step bookkeeping (ant_flagrun_env.py:162-204)
    from hrl_pybullet_envs.envs.ant_flagrun.ant_flagrun_env import AntFlagrunBulletEnv
    fr = []
    for k in range(60):
        lrs = np.random.RandomState(13000 + k)
        n_goals = int(lrs.randint(0, 4))                      # goals left in the list (0 -> IndexError -> done)
        goals = [tuple(lrs.uniform(-5, 5, 2)) for _ in range(n_goals)]
        tol, timeout = 0.5, int(lrs.choice([5, 200]))
        switch = bool(lrs.randint(0, 2)) if k % 4 == 0 else True
        steps0 = int(lrs.randint(0, 7)); rewarded0 = bool(lrs.randint(0, 2)) if not switch else False
        wtd = float(lrs.uniform(0.0, 1.0)) if k % 2 == 0 else float(lrs.uniform(0.5, 6.0))
        inner_r = float(lrs.uniform(-2, 2)); inner_d = bool(lrs.randint(0, 8) == 0)
        s_old = lrs.uniform(-1, 1, 28).astype(np.float32); s_new = lrs.uniform(-1, 1, 28).astype(np.float32)
        pos = lrs.uniform(-4, 4, 3)
        robot = NS(walk_target_dist=wtd, walk_target_x=0.0, walk_target_y=0.0, body_real_xyz=pos,
                   robot_body=NS(get_position=lambda: pos), calc_potential=lambda: -wtd / 0.0165,
                   calc_state=lambda: s_new.copy())
        self = AntFlagrunBulletEnv.__new__(AntFlagrunBulletEnv)
        self.__dict__.update(dict(robot=robot, tol=tol, timeout=timeout, switch_flag_on_collision=switch, max_targets=100,
                                  goals=list(goals), steps_since_goal_change=steps0, _rewarded=rewarded0, debug=False,
                                  use_sensor=False, isRender=False, flag=None, walk_target_x=1.0, walk_target_y=2.0,
                                  _sq_dist_goal=3.0, _goal_start_pos=np.array([0.5, 0.5]), potential=-77.0,
                                  _super_step_result=(s_old.copy(), inner_r, inner_d, {})))
        obs, r, d, info = AntFlagrunBulletEnv.step(self, np.zeros(8))
        fr.append({'n_goals': n_goals, 'last_goal': list(map(float, goals[-1])) if goals else None, 'tol': tol, 'timeout': timeout,
                   'switch': switch, 'steps_before': steps0, 'rewarded_before': rewarded0, 'walk_target_dist': wtd,
                   'inner_rew': inner_r, 'inner_done': inner_d, 'rew': float(r), 'done': bool(d),
                   'steps_after': int(self.steps_since_goal_change), 'rewarded_after': bool(self._rewarded),
                   'goals_left': len(self.goals), 'retargeted': bool('target' in info),
                   'obs_is_new_state': bool(np.array_equal(obs, s_new)),
                   'target_after': [float(self.walk_target_x), float(self.walk_target_y)]})
    G['flagrun_step'] = fr

    # the same step() with the CLASS-LEVEL reward weights moved off their defaults (ant_flagrun_env.py:157-160, read off the class in :169-186), the
    # path reward's state (`_goal_start_pos`, `_sq_dist_goal`, :100-103) drawn at random -- including the constructor's 0 / (0, 0) of an env that
    # never got a goal (:47-49: 0 / 0 -> NaN) -- and what set_target() leaves behind when the step switches goals; info['target'] (:191,199)
    fw = []
    defaults = {k: getattr(AntFlagrunBulletEnv, k) for k in ('ant_env_rew_weight', 'path_rew_weight', 'dist_rew_weight', 'goal_reach_rew')}
    for k in range(90):
        lrs = np.random.RandomState(13500 + k)
        w = {'ant_env_rew_weight': float(lrs.choice([1, 0, 0.5, -2])), 'path_rew_weight': float(lrs.choice([0, 1, 0.3, -4])),
             'dist_rew_weight': float(lrs.choice([0, 1, 0.05])), 'goal_reach_rew': float(lrs.choice([5000, 0, 10, -1]))}
        if k % 9 == 0: w = dict(defaults)
        n_goals = int(lrs.randint(0, 4))
        goals = [tuple(lrs.uniform(-5, 5, 2)) for _ in range(n_goals)]
        tol, timeout = float(lrs.choice([0.5, 1.0])), int(lrs.choice([5, 200]))
        switch = bool(lrs.randint(0, 2)) if k % 4 == 0 else True
        steps0 = int(lrs.randint(0, 7)); rewarded0 = bool(lrs.randint(0, 2)) if not switch else False
        pos = lrs.uniform(-4, 4, 3)
        goal0 = lrs.uniform(-5, 5, 2)
        if k % 2 == 0: goal0 = pos[:2] + lrs.uniform(-1, 1, 2) * 0.6 * tol     # near the goal: rewards and switches
        wtd = float(np.linalg.norm(goal0 - pos[:2]))
        start = lrs.uniform(-4, 4, 2); sq = float(np.linalg.norm(goal0 - start) ** 2)
        if k % 10 == 3: start, sq = np.array([0, 0]), 0                          # as constructed: no goal received yet
        inner_r = float(lrs.uniform(-2, 2)); inner_d = bool(lrs.randint(0, 8) == 0)
        s_old = lrs.uniform(-1, 1, 28).astype(np.float32); s_new = lrs.uniform(-1, 1, 28).astype(np.float32)
        robot = NS(walk_target_dist=wtd, walk_target_x=float(goal0[0]), walk_target_y=float(goal0[1]), body_real_xyz=pos,
                   robot_body=NS(get_position=lambda: pos), calc_potential=lambda: -wtd / 0.0165, calc_state=lambda: s_new.copy())
        self = AntFlagrunBulletEnv.__new__(AntFlagrunBulletEnv)
        self.__dict__.update(dict(robot=robot, tol=tol, timeout=timeout, switch_flag_on_collision=switch, max_targets=100,
                                  goals=list(goals), steps_since_goal_change=steps0, _rewarded=rewarded0, debug=False,
                                  use_sensor=False, isRender=False, flag=None, walk_target_x=float(goal0[0]), walk_target_y=float(goal0[1]),
                                  _sq_dist_goal=sq, _goal_start_pos=np.array(start), potential=-77.0,
                                  _super_step_result=(s_old.copy(), inner_r, inner_d, {})))
        for name, v in w.items(): setattr(AntFlagrunBulletEnv, name, v)
        try:
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                with np.errstate(all='ignore'):
                    obs, r, d, info = AntFlagrunBulletEnv.step(self, np.zeros(8))
        finally:
            for name, v in defaults.items(): setattr(AntFlagrunBulletEnv, name, v)
        fw.append({'weights': w, 'n_goals': n_goals, 'goals': [list(map(float, g)) for g in goals], 'tol': tol, 'timeout': timeout, 'switch': switch,
                   'steps_before': steps0, 'rewarded_before': rewarded0, 'robot_xy': [float(pos[0]), float(pos[1])], 'goal': [float(goal0[0]), float(goal0[1])],
                   'goal_start_pos': [float(start[0]), float(start[1])], 'sq_dist_goal': float(sq), 'walk_target_dist': wtd,
                   'inner_rew': inner_r, 'inner_done': inner_d, 'rew': float(r), 'done': bool(d),
                   'steps_after': int(self.steps_since_goal_change), 'rewarded_after': bool(self._rewarded), 'goals_left': len(self.goals),
                   'info_target': [float(info['target'][0]), float(info['target'][1])] if 'target' in info else None,
                   'goal_start_pos_after': [float(self._goal_start_pos[0]), float(self._goal_start_pos[1])], 'sq_dist_goal_after': float(self._sq_dist_goal)})
    G['flagrun_weights_step'] = fw
    # create_target rejection logic (ant_flagrun_env.py:71-78) with the uniforms it consumed
    class LogU:
        def __init__(self, seed): self.rs = np.random.RandomState(seed); self.log = []
        def uniform(self, lo, hi):
            u = self.rs.uniform(lo, hi); self.log.append(float(u)); return u
    ct = []
    for k in range(40):
        self = NS(size=10 if k % 2 == 0 else 1.2, mpi_common_rand=LogU(14000 + k))
        g = AntFlagrunBulletEnv.create_target(self)
        ct.append({'size': self.size, 'draws': self.mpi_common_rand.log, 'goal': [float(g[0]), float(g[1])]})
    G['flagrun_create_target'] = ct

    # create_close_target (ant_flagrun_env.py:80-89) with the [0, 1) uniforms and the randint results it consumed, and the
    # step bookkeeping in max_target_dist mode (next_target -> create_close_target, :111-112: goals never run out)
    class LogR:
        def __init__(self, seed): self.rs = np.random.RandomState(seed); self.u = []; self.b = []
        def uniform(self, lo, hi):  # numpy: lo + (hi - lo) * random_sample()
            u01 = float(self.rs.random_sample()); self.u.append(u01); return lo + (hi - lo) * u01
        def randint(self, lo, hi):
            v = int(self.rs.randint(lo, hi)); self.b.append(v); return v
    cc = []
    for k in range(40):
        lrs = np.random.RandomState(15000 + k)
        size = 10.0 if k % 3 else 3.0
        tol, mtd = 0.5, float(lrs.uniform(1.5, 6.0))
        pos = lrs.uniform(-size / 2, size / 2, 3) * (0.98 if k % 2 else 0.5)   # odd cases hug the arena edge -> rejections
        self = NS(size=size, tol=tol, max_target_dist=mtd, mpi_common_rand=LogR(15500 + k), robot=NS(body_real_xyz=pos))
        g = AntFlagrunBulletEnv.create_close_target(self)
        cc.append({'size': size, 'tol': tol, 'max_target_dist': mtd, 'robot_xy': [float(pos[0]), float(pos[1])],
                   'u': self.mpi_common_rand.u, 'b': self.mpi_common_rand.b, 'goal': [float(g[0]), float(g[1])]})
    G['flagrun_create_close_target'] = cc
    fc = []
    for k in range(30):
        lrs = np.random.RandomState(16000 + k)
        tol, timeout = 0.5, int(lrs.choice([5, 200]))
        switch = bool(lrs.randint(0, 2)) if k % 4 == 0 else True
        steps0 = int(lrs.randint(0, 7)); rewarded0 = bool(lrs.randint(0, 2)) if not switch else False
        wtd = float(lrs.uniform(0.0, 1.0)) if k % 2 == 0 else float(lrs.uniform(0.5, 6.0))
        inner_r = float(lrs.uniform(-2, 2)); inner_d = bool(lrs.randint(0, 8) == 0)
        s_old = lrs.uniform(-1, 1, 28).astype(np.float32); s_new = lrs.uniform(-1, 1, 28).astype(np.float32)
        pos = lrs.uniform(-4, 4, 3)
        robot = NS(walk_target_dist=wtd, walk_target_x=0.0, walk_target_y=0.0, body_real_xyz=pos,
                   robot_body=NS(get_position=lambda: pos), calc_potential=lambda: -wtd / 0.0165,
                   calc_state=lambda: s_new.copy())
        self = AntFlagrunBulletEnv.__new__(AntFlagrunBulletEnv)
        rnd = LogR(16500 + k)
        self.__dict__.update(dict(robot=robot, tol=tol, timeout=timeout, switch_flag_on_collision=switch, max_targets=0,
                                  max_target_dist=3.0, size=10, mpi_common_rand=rnd,
                                  goals=[], steps_since_goal_change=steps0, _rewarded=rewarded0, debug=False,
                                  use_sensor=False, isRender=False, flag=None, walk_target_x=1.0, walk_target_y=2.0,
                                  _sq_dist_goal=3.0, _goal_start_pos=np.array([0.5, 0.5]), potential=-77.0,
                                  _super_step_result=(s_old.copy(), inner_r, inner_d, {})))
        obs, r, d, info = AntFlagrunBulletEnv.step(self, np.zeros(8))
        fc.append({'tol': tol, 'timeout': timeout, 'switch': switch, 'steps_before': steps0, 'rewarded_before': rewarded0,
                   'walk_target_dist': wtd, 'inner_rew': inner_r, 'inner_done': inner_d, 'rew': float(r), 'done': bool(d),
                   'steps_after': int(self.steps_since_goal_change), 'rewarded_after': bool(self._rewarded),
                   'retargeted': bool('target' in info), 'robot_xy': [float(pos[0]), float(pos[1])], 'u': rnd.u, 'b': rnd.b,
                   'target_after': [float(self.walk_target_x), float(self.walk_target_y)]})
    G['flagrun_close_step'] = fc

    # ---------------------------------------------------------------- contact-based pickup (robot_coll_dist <= 0):
    # ant_gather_env.py:113-116 / gather_base.py:103-106 with scripted getContactPoints() results
    def contact_step_case(cls, k):
        lrs = np.random.RandomState(17000 + k)
        n_bins = 10 if cls is AntGatherBulletEnv else 5
        respawn = k % 5 != 0
        sc = GatherScene(None, 9.8, 0.0165 / 4, 4, (15, 15), 8, 8, 2.0, respawn)
        sc.rs = LoggingRS(17500 + k)
        sc.loaded = True
        cl = FakeClient()
        sc._p = cl
        sc.episode_restart(cl)
        n_restart = len(sc.rs.log)
        items0 = [list(sc.all_items[i]) for i in sc.all_items]
        ids = list(sc.all_items.keys())
        xy = lrs.uniform(-6, 6, 2)
        z = lrs.uniform(0.3, 0.9)
        rpy = [lrs.uniform(-.3, .3), lrs.uniform(-.3, .3), lrs.uniform(-np.pi, np.pi)]
        body = Body([xy[0], xy[1], z], rpy)
        nstate = 28 if cls is AntGatherBulletEnv else 8
        st = lrs.uniform(-1, 1, nstate)
        initial_z = 0.75 if cls is AntGatherBulletEnv else 1.0
        st[0] = z - initial_z
        st = st.astype(np.float32)
        # contact points of the robot: ground (id 3), a wall (id 4) and items, some of them several times
        n_cp = int(lrs.randint(0, 9))
        touched = [int(lrs.choice([3, 4] + ids[:6] + ids[8:12])) for _ in range(n_cp)]
        cl.getContactPoints = lambda body_id: [(0, body_id, oid, -1, -1) for oid in touched]
        if cls is AntGatherBulletEnv:
            robot = NS(apply_action=lambda a: None, calc_state=lambda: st.copy(), initial_z=initial_z, objects=[1],
                       body_rpy=rpy, alive_bonus=lambda zz, p: +1 if zz > 0.26 else -1)
            self = cls.__new__(cls)
            self.__dict__.update(dict(robot=robot, scene=NS(global_step=lambda: None), stadium_scene=sc,
                                      parts={'torso': body}, robot_body=body, robot_coll_dist=0, use_sensor=True,
                                      n_bins=n_bins, sensor_span=np.pi, sensor_range=20., dying_cost=-10,
                                      debug=False, _p=cl))
        else:
            robot = NS(apply_action=lambda a: None, calc_state=lambda: st.copy(), initial_z=initial_z, objects=[1],
                       robot_body=body, alive_bonus=lambda zz, p: 1)
            self = cls.__new__(cls)
            self.__dict__.update(dict(robot=robot, scene=NS(global_step=lambda: None), stadium_scene=sc,
                                      robot_coll_dist=-1, use_sensor=True, n_bins=n_bins, sensor_span=np.pi,
                                      sensor_range=20., dying_cost=-10, debug=False, _p=cl))
        obs, rew, done, info = cls.step(self, np.zeros(8))
        return {'cls': cls.__name__, 'n_bins': n_bins, 'respawn': respawn, 'state_in': st.astype(float).tolist(), 'initial_z': initial_z,
                'torso_xyz': [float(xy[0]), float(xy[1]), float(z)], 'rpy': list(map(float, rpy)),
                'items_before': [p[:2] for p in items0], 'respawn_draws': sc.rs.log[n_restart:],
                'contact_items': [ids.index(o) if o in ids else -1 for o in touched],
                'items_after': [list(sc.all_items[i])[:2] for i in sc.all_items],
                'obs': tolist(obs), 'rew': float(rew), 'done': bool(done), 'food_rew': float(info['food_rew']),
                'dead_rew': float(info['dead_rew'])}

    G['gather_contact_step'] = [contact_step_case(AntGatherBulletEnv, k) for k in range(40)] + \
        [contact_step_case(GatherBulletEnv, 100 + k) for k in range(20)]

    # ---------------------------------------------------------------- which target the potential of a reset belongs to
    # AntFlagrunBulletEnv.reset / next_target (ant_flagrun_env.py:110-155) and AntMazeBulletEnv.reset
    # (ant_maze_bullet_env.py:104-121) run IN-TREE around the stand-in bookkeeping of AntBulletEnv above.
    DT = 0.0165

    class SeqRobot:
        """Duck-typed walker: walk_target_dist is measured from the body position (stand-in for upstream's parts centroid)."""

        def __init__(self, home):
            self._home, self._pos = list(home), list(home)
            self.walk_target_x, self.walk_target_y = 1e3, 0.0   # upstream WalkerBase default
            self.walk_target_dist = 0.0
            self.objects = [1]
            self.robot_body = NS(get_position=lambda: np.array(self._pos, dtype=float))
            self.start_pos_x, self.start_pos_y, self.start_pos_z = -2.0, -5.0, 0.25

        body_real_xyz = property(lambda self: np.array(self._pos, dtype=float))

        def calc_state(self):
            self.walk_target_dist = float(np.linalg.norm([self.walk_target_y - self._pos[1], self.walk_target_x - self._pos[0]]))
            return np.array(self._pos, dtype=np.float32)

        def calc_potential(self):
            return -self.walk_target_dist / DT

        def robot_specific_reset(self, p):
            pass

    class SeqClient(FakeClient):
        def __init__(self, robot):
            super().__init__()
            self.robot = robot

        def resetBasePositionAndOrientation(self, obj, pos, orn):
            self.robot._pos = list(pos)

    seq = {'dt': DT, 'flagrun': [], 'maze': []}
    for k in range(6):
        lrs = np.random.RandomState(18000 + k)
        robot = SeqRobot([0.0, 0.0, 0.75])
        env = AntFlagrunBulletEnv.__new__(AntFlagrunBulletEnv)
        env.__dict__.update(dict(robot=robot, _p=SeqClient(robot), scene=NS(_p=None), _bookkeeping=True, tol=0.5, timeout=200,
                                 switch_flag_on_collision=True, max_targets=3, max_target_dist=0, manual_goal_creation=False,
                                 size=10, mpi_common_rand=np.random.RandomState(18100 + k), goals=[], steps_since_goal_change=0,
                                 _rewarded=False, debug=False, use_sensor=False, isRender=False, flag=None,
                                 walk_target_x=1e3, walk_target_y=0.0, _sq_dist_goal=0, _goal_start_pos=np.array([0, 0])))
        events = []
        for ep in range(3):
            AntFlagrunBulletEnv.reset(env)
            events.append({'op': 'reset', 'pos': list(map(float, robot._pos)), 'target': [float(env.walk_target_x), float(env.walk_target_y)],
                           'potential': float(env.potential)})
            for t in range(3):
                robot._pos = [float(v) for v in (np.array(robot._pos) + np.r_[lrs.uniform(-0.3, 0.3, 2), 0.0])]
                if t == 1 and ep == 1:  # jump onto the goal: +5000, retarget
                    robot._pos = [float(env.walk_target_x) + 0.1, float(env.walk_target_y), 0.5]
                _, r, d, info = AntFlagrunBulletEnv.step(env, np.zeros(8))
                events.append({'op': 'step', 'pos': list(map(float, robot._pos)), 'target': [float(env.walk_target_x), float(env.walk_target_y)],
                               'potential': float(env.potential), 'rew': float(r), 'done': bool(d), 'retargeted': 'target' in info})
        seq['flagrun'].append(events)
    for k in range(6):
        lrs = np.random.RandomState(18500 + k)
        robot = SeqRobot([0.0, 0.0, 0.75])
        env = AntMazeBulletEnv.__new__(AntMazeBulletEnv)
        targets = ([2, -3], [2, 0], [2, 3], [-2, 4])
        maze._p = None  # reset() hands scene._p to robot_specific_reset (a no-op here)
        env.__dict__.update(dict(robot=robot, _p=SeqClient(robot), scene=maze, _bookkeeping=True, n_bins=10, sensor_range=5.0,
                                 sensor_span=2 * np.pi, targets=targets, sense_walls=True, sense_target=False, done_at_target=True,
                                 max_steps=-1, t=0, tol=1.5, inner_rew_weight=1.0, targ_dist_rew=False,
                                 target_encoding=PositionEncoding(0), debug=0, rs=np.random.RandomState(18600 + k),
                                 robot_body=NS(pose=lambda: NS(xyz=lambda: np.array(robot._pos, dtype=float), rpy=lambda: [0, 0, 0.3]))))
        events = []
        for ep in range(3):
            AntMazeBulletEnv.reset(env)
            events.append({'op': 'reset', 'pos': list(map(float, robot._pos)), 'target': list(map(float, env.target)),
                           'potential': float(env.potential)})
            for t in range(2):
                robot._pos = [float(v) for v in (np.array(robot._pos) + np.r_[lrs.uniform(-0.3, 0.3, 2), 0.0])]
                _, r, d, _ = AntMazeBulletEnv.step(env, np.zeros(8))
                events.append({'op': 'step', 'pos': list(map(float, robot._pos)), 'target': list(map(float, env.target)),
                               'potential': float(env.potential), 'rew': float(r), 'done': bool(d)})
        seq['maze'].append(events)
    G['reset_potential_seq'] = seq

    # ---------------------------------------------------------------- manual_goal_creation (ant_flagrun_env.py:27,45,112-120,150-153)
    # reset() / `env.goals = [...]` / next_target() / step() run IN-TREE on the sequence robot above: which goal of the list
    # next_target() takes (goals.pop(): the LAST), goals running out (IndexError -> done), the timeout, what a reset keeps
    # (the walk target) and drops (the list); with max_targets < 1 next_target() ignores the list and calls
    # create_close_target (the uniforms / randint results it consumed are logged).
    man = {'dt': DT, 'list': [], 'close': []}
    for k in range(8):
        lrs = np.random.RandomState(19000 + k)
        robot = SeqRobot([0.0, 0.0, 0.75])
        timeout = 200 if k % 2 == 0 else 2
        env = AntFlagrunBulletEnv.__new__(AntFlagrunBulletEnv)
        env.__dict__.update(dict(robot=robot, _p=SeqClient(robot), scene=NS(_p=None), _bookkeeping=True, tol=0.5, timeout=timeout,
                                 switch_flag_on_collision=(k != 5), max_targets=100, max_target_dist=0, manual_goal_creation=True,
                                 size=10, mpi_common_rand=np.random.RandomState(19100 + k), goals=[], steps_since_goal_change=0,
                                 _rewarded=False, debug=False, use_sensor=False, isRender=False, flag=None,
                                 walk_target_x=1e3, walk_target_y=0.0, _sq_dist_goal=0, _goal_start_pos=np.array([0, 0])))
        events = []

        def snap(op, **kw):
            e = {'op': op, 'pos': list(map(float, robot._pos)), 'target': [float(env.walk_target_x), float(env.walk_target_y)],
                 'potential': float(env.potential), 'goals_left': [list(map(float, g)) for g in env.goals],
                 'steps': int(env.steps_since_goal_change), 'rewarded': bool(env._rewarded)}
            e.update(kw)
            events.append(e)

        for ep in range(2):
            AntFlagrunBulletEnv.reset(env)
            snap('reset')
            n_goals = int(lrs.randint(1, 5))
            goals = [tuple(map(float, lrs.uniform(-4, 4, 2))) for _ in range(n_goals)]
            env.goals = list(goals)
            AntFlagrunBulletEnv.next_target(env)
            snap('set_goals', goals=[list(g) for g in goals])
            for t in range(3 * n_goals + 3):
                robot._pos = [float(v) for v in (np.array(robot._pos) + np.r_[lrs.uniform(-0.3, 0.3, 2), 0.0])]
                if t % 3 == 1:  # jump onto the goal: +5000, retarget (or IndexError -> done when the list is empty)
                    robot._pos = [float(env.walk_target_x) + 0.1, float(env.walk_target_y) - 0.2, 0.5]
                _, r, d, info = AntFlagrunBulletEnv.step(env, np.zeros(8))
                snap('step', rew=float(r), done=bool(d), retargeted='target' in info)
                if d:
                    break
            if ep == 0:  # next_target() from outside on an empty / a one-goal list
                env.goals = []
                try:
                    AntFlagrunBulletEnv.next_target(env)
                    raised = False
                except IndexError:
                    raised = True
                snap('next_target', raised=raised, goals=[])
                g1 = [tuple(map(float, lrs.uniform(-4, 4, 2)))]
                env.goals = list(g1)
                AntFlagrunBulletEnv.next_target(env)
                snap('next_target', raised=False, goals=[list(g1[0])])
        man['list'].append({'timeout': timeout, 'switch': bool(env.switch_flag_on_collision), 'events': events})
    for k in range(6):
        lrs = np.random.RandomState(19500 + k)
        robot = SeqRobot([0.0, 0.0, 0.75])
        rnd = LogR(19600 + k)
        env = AntFlagrunBulletEnv.__new__(AntFlagrunBulletEnv)
        env.__dict__.update(dict(robot=robot, _p=SeqClient(robot), scene=NS(_p=None), _bookkeeping=True, tol=0.5, timeout=3,
                                 switch_flag_on_collision=True, max_targets=0, max_target_dist=3.0, manual_goal_creation=True,
                                 size=10, mpi_common_rand=rnd, goals=[], steps_since_goal_change=0,
                                 _rewarded=False, debug=False, use_sensor=False, isRender=False, flag=None,
                                 walk_target_x=1e3, walk_target_y=0.0, _sq_dist_goal=0, _goal_start_pos=np.array([0, 0])))
        events = []

        def snapc(op, nu, nb, **kw):
            e = {'op': op, 'pos': list(map(float, robot._pos)), 'target': [float(env.walk_target_x), float(env.walk_target_y)],
                 'potential': float(env.potential), 'steps': int(env.steps_since_goal_change), 'rewarded': bool(env._rewarded),
                 'u': rnd.u[nu:], 'b': rnd.b[nb:]}
            e.update(kw)
            events.append(e)

        for ep in range(2):
            nu, nb = len(rnd.u), len(rnd.b)
            AntFlagrunBulletEnv.reset(env)
            snapc('reset', nu, nb)
            nu, nb = len(rnd.u), len(rnd.b)
            env.goals = [(9.0, 9.0)]          # ignored: max_targets < 1
            AntFlagrunBulletEnv.next_target(env)
            snapc('next_target', nu, nb, list_len_after=len(env.goals))
            for t in range(7):
                robot._pos = [float(v) for v in (np.array(robot._pos) + np.r_[lrs.uniform(-0.3, 0.3, 2), 0.0])]
                if t == 4:
                    robot._pos = [float(env.walk_target_x) - 0.1, float(env.walk_target_y) + 0.1, 0.5]
                nu, nb = len(rnd.u), len(rnd.b)
                _, r, d, info = AntFlagrunBulletEnv.step(env, np.zeros(8))
                snapc('step', nu, nb, rew=float(r), done=bool(d), retargeted='target' in info)
        man['close'].append({'timeout': 3, 'max_target_dist': 3.0, 'size': 10, 'tol': 0.5, 'events': events})
    G['flagrun_manual_seq'] = man

    # ---------------------------------------------------------------- constructor arguments beyond the defaults' sizes
    # ant_gather_env.py:16-29 / point_gather_env.py:8-21 take any n_food, n_poison, n_bins; ant_maze_bullet_env.py:23-25 any `targets`
    # and n_bins; ant_maze_mj_env.py:50 any n_bins.  The same reference functions as above, on 20 food + 12 poison items with 24 bins,
    # 12 maze targets with 33 bins, AntMazeMj with 16 and 64 bins (its own RandomStates: the fixtures above stay byte-identical).
    big = {'food_sensor': [], 'gather_step': [], 'maze_step': [], 'maze_mj_step': []}
    brs = np.random.RandomState(424242)
    for k in range(30):
        big['food_sensor'].append(sensor_case(AntGatherBulletEnv if k % 3 else GatherBulletEnv, 24, np.pi, 20., 20, 12, brs.uniform(-7, 7, 2),
                                              brs.uniform(-np.pi, np.pi), np.random.RandomState(31000 + k)))
    for k in range(6):
        big['food_sensor'].append(sensor_case(AntGatherBulletEnv, 64, 2.5, 30., 40, 24, brs.uniform(-7, 7, 2),
                                              brs.uniform(-np.pi, np.pi), np.random.RandomState(32000 + k)))
    for k in range(30):
        big['gather_step'].append(gather_step_case(AntGatherBulletEnv if k % 3 else GatherBulletEnv, 400 + k, None, 20, 12, 24))
    big['gather_scene'] = [scene_run(31, (15, 15), 20, 12, 2.0, True, [(0, (1, 1, .5)), (19, (1, 1, .5)), (20, (0, 0, .5)), (31, (6.5, 6.5, .5)), (32, (0, 0, .5))])]
    many_targets = tuple([float(-2.0 + 0.5 * i), float(-4.0 + 0.7 * i)] for i in range(12))
    for k in range(20):
        big['maze_step'].append(maze_case(500 + k, k % 2, sense_target=k % 4 < 2, near=k >= 14, n_bins=33, targets=many_targets))
    big['maze_targets'] = [list(t) for t in many_targets]
    for k in range(10):
        big['maze_mj_step'].append(dict(maze_mj_case(600 + k, n_bins=16), n_bins=16))
    for k in range(4):
        big['maze_mj_step'].append(dict(maze_mj_case(700 + k, n_bins=64), n_bins=64))
    G['big_config'] = big

    # ---------------------------------------------------------------- random constructor arguments, whole task half of step()
    # ant_gather_env.py:76-119 / gather_base.py:74-109 with EVERY constructor argument drawn at random (n_food, n_poison, n_bins, use_sensor,
    # robot_coll_dist of either sign, respawn, world size, spacing, sensor span / range, dying cost), the robot next to an item, for the contact
    # mode a random list of touched bodies: the combinations of branches the fixtures above take one at a time (its own RandomStates again).
    def f32(x):
        return float(np.float32(x))

    def random_gather_case(k):
        lrs = np.random.RandomState(23000 + k)
        cls = AntGatherBulletEnv if lrs.rand() < 0.6 else GatherBulletEnv
        ant = cls is AntGatherBulletEnv
        total = int(lrs.choice([1, 2, 7, 8, 15, 16, 17, 24, 33, 48, 49, 64]))
        n_food = int(lrs.randint(0, total + 1)) if lrs.rand() < 0.8 else int(lrs.choice([0, total]))
        n_poison = total - n_food
        n_bins = int(lrs.choice([1, 2, 3, 5, 10, 16, 17, 24, 33, 64]))
        use_sensor = bool(lrs.rand() < 0.5)
        coll = float(lrs.choice([1.0, 1.0, 0.3, 4.0, 0.0, -1.0]))
        respawn = bool(lrs.rand() < 0.7)
        # the C-ABI carries these as fp32: draw values fp32 holds exactly (pi and 2 pi are recognised by the config, as in the other fixtures)
        world = (f32(lrs.uniform(6, 24)), f32(lrs.uniform(6, 24)))
        spacing = f32(lrs.uniform(0.3, min(world) / 3.2))
        span = float(lrs.choice([np.pi, 2 * np.pi, f32(lrs.uniform(0.3, 6.5))]))
        srange = float(lrs.choice([2.0, 9.0, 20.0, 60.0]))
        dying_cost = float(lrs.choice([-10.0, 0.0, -1.5, 3.0]))
        sc = GatherScene(None, 9.8, 0.0165 / 4, 4, world, n_food, n_poison, spacing, respawn)
        sc.rs = LoggingRS(23500 + k)
        sc.loaded = True
        cl = FakeClient()
        sc._p = cl
        sc.episode_restart(cl)
        n_restart = len(sc.rs.log)
        items0 = [list(sc.all_items[i]) for i in sc.all_items]
        ids = list(sc.all_items.keys())
        tgt = sc.all_items[ids[lrs.randint(0, total)]]
        xy = np.array(tgt[:2]) + lrs.uniform(-1.2, 1.2, 2)
        z = 0.2 if k % 17 == 0 else lrs.uniform(0.3, 0.9)
        rpy = [lrs.uniform(-.3, .3), lrs.uniform(-.3, .3), lrs.uniform(-np.pi, np.pi)]
        body = Body([xy[0], xy[1], z], rpy)
        st = lrs.uniform(-1, 1, 28 if ant else 8)
        initial_z = 0.75 if ant else 1.0
        st[0] = z - initial_z
        st = st.astype(np.float32)
        touched = []
        if coll <= 0:
            n_cp = int(lrs.randint(0, 9))
            touched = [int(lrs.choice([3, 4] + ids[:min(total, 6)] + ids[-min(total, 4):])) for _ in range(n_cp)]
        cl.getContactPoints = lambda body_id: [(0, body_id, oid, -1, -1) for oid in touched]
        robot = NS(apply_action=lambda a: None, calc_state=lambda: st.copy(), initial_z=initial_z, objects=[1], body_rpy=rpy, robot_body=body,
                   alive_bonus=(lambda zz, p: +1 if zz > 0.26 else -1) if ant else (lambda zz, p: 1))
        self = cls.__new__(cls)
        self.__dict__.update(dict(robot=robot, scene=NS(global_step=lambda: None), stadium_scene=sc, parts={'torso': body}, robot_body=body,
                                  robot_coll_dist=coll, use_sensor=use_sensor, n_bins=n_bins, sensor_span=span, sensor_range=srange,
                                  dying_cost=dying_cost, debug=False, _p=cl))
        obs, rew, done, info = cls.step(self, np.zeros(8))
        return {'cls': cls.__name__, 'n_food': n_food, 'n_poison': n_poison, 'n_bins': n_bins, 'use_sensor': use_sensor, 'coll_dist': coll,
                'respawn': respawn, 'world': list(world), 'spacing': spacing, 'span': span, 'range': srange, 'dying_cost': dying_cost,
                'state_in': st.astype(float).tolist(), 'initial_z': initial_z, 'torso_xyz': [float(xy[0]), float(xy[1]), float(z)],
                'rpy': list(map(float, rpy)), 'items_before': [q[:2] for q in items0], 'respawn_draws': sc.rs.log[n_restart:],
                'contact_items': [ids.index(o) if o in ids else -1 for o in touched],
                'items_after': [list(sc.all_items[i])[:2] for i in sc.all_items],
                'obs': tolist(obs), 'rew': float(rew), 'done': bool(done), 'food_rew': float(info['food_rew']), 'dead_rew': float(info['dead_rew'])}

    def random_maze_case(k):
        """ant_maze_bullet_env.py:63-97,123-178 with every constructor argument drawn at random"""
        lrs = np.random.RandomState(24000 + k)
        n_bins = int(lrs.choice([2, 3, 5, 10, 16, 17, 33, 64]))
        span = float(lrs.choice([2 * np.pi, np.pi, f32(lrs.uniform(0.3, 6.5))]))
        srange = float(lrs.choice([2.0, 5.0, 9.0, 30.0]))
        nt = int(lrs.choice([1, 2, 4, 9, 33, 64]))
        targets = [[f32(lrs.uniform(-4.5, 4.5)), f32(lrs.uniform(-8.5, 8.5))] for _ in range(nt)]
        target = np.array(targets[lrs.randint(0, nt)])
        tol = float(lrs.choice([1.5, 0.2, 0.8, 3.0, 12.0]))
        xy = np.array([lrs.uniform(-4.5, 4.5), lrs.uniform(-8.5, 8.5)])
        if lrs.rand() < 0.5:
            xy = target + lrs.uniform(-1.0, 1.0, 2) * min(tol, 3.0)
        rpy = [lrs.uniform(-.2, .2), lrs.uniform(-.2, .2), lrs.uniform(-np.pi, np.pi)]
        body = Body([xy[0], xy[1], 0.45], rpy)
        ant_obs = lrs.uniform(-1, 1, 28).astype(np.float32)
        inner_rew, inner_done = float(lrs.uniform(-1, 1)), bool(lrs.rand() < 0.2)
        wtd = float(np.linalg.norm(target - (xy + lrs.uniform(-0.3, 0.3, 2))))
        args = dict(n_bins=n_bins, sensor_range=srange, sensor_span=span, sense_walls=bool(lrs.rand() < 0.7), sense_target=bool(lrs.rand() < 0.5),
                    done_at_target=bool(lrs.rand() < 0.5), max_steps=int(lrs.choice([-1, 1, 2, 5, 26])), tol=tol,
                    inner_rew_weight=float(lrs.choice([0.0, 0.5, 1.0, -2.0])), targ_dist_rew=bool(lrs.rand() < 0.5))
        encoding, t0 = int(lrs.rand() < 0.5), int(lrs.choice([0, 0, 1, 3, 4, 24, 25]))
        self = AntMazeBulletEnv.__new__(AntMazeBulletEnv)
        self.__dict__.update(dict(args, targets=targets, t=t0, target_encoding=PositionEncoding(encoding), target=target, debug=0, scene=maze,
                                  robot=NS(body_real_xyz=[xy[0], xy[1], 0.45], walk_target_dist=wtd), robot_body=body,
                                  _super_step_result=(ant_obs, inner_rew, inner_done, {})))
        obs, rew, d, _ = AntMazeBulletEnv.step(self, np.zeros(8))
        return dict(args, encoding=encoding, t_before=t0, targets=targets, torso_xy=xy.tolist(), rpy=list(map(float, rpy)), target=target.tolist(),
                    ant_obs=ant_obs.astype(float).tolist(), inner_rew=inner_rew, inner_done=inner_done, walk_target_dist=wtd,
                    obs=tolist(obs), rew=float(rew), done=bool(d))

    def random_flagrun_case(k):
        """ant_flagrun_env.py:122-130,162-204 with tolerance / timeout (0 = off) / switch_flag_on_collision / the wall sensor over an arena of
        any size drawn at random (list mode: goals are popped from the back of the list)"""
        lrs = np.random.RandomState(25000 + k)
        size = f32(lrs.choice([10.0, 3.0, 6.0, 25.0]))
        n_goals = int(lrs.randint(0, 4))
        goals = [tuple(lrs.uniform(-size / 2, size / 2, 2)) for _ in range(n_goals)]
        tol, timeout = f32(lrs.choice([0.5, 0.2, 1.0, 2.0])), int(lrs.choice([0, 1, 3, 5, 200]))
        switch = bool(lrs.rand() < 0.6)
        steps0, rewarded0 = int(lrs.randint(0, 7)), bool(lrs.rand() < 0.3)
        wtd = float(lrs.uniform(0.0, 2.5))
        inner_r, inner_d = float(lrs.uniform(-2, 2)), bool(lrs.rand() < 0.15)
        s_old = lrs.uniform(-1, 1, 28).astype(np.float32); s_new = lrs.uniform(-1, 1, 28).astype(np.float32)
        use_sensor = bool(lrs.rand() < 0.6)
        n_bins = int(lrs.choice([2, 5, 8, 17, 40])); span = float(lrs.choice([np.pi, 2 * np.pi, f32(lrs.uniform(0.4, 6.0))])); srange = f32(lrs.choice([1.0, 4.0, 9.0]))
        pos = np.r_[lrs.uniform(-size / 2, size / 2, 2), 0.5]
        yaw = float(lrs.uniform(-np.pi, np.pi))
        scene = SizeableEnclosedScene(None, 9.8, 0.0165 / 4, 4, (size + 2, size + 2))
        robot = NS(walk_target_dist=wtd, walk_target_x=0.0, walk_target_y=0.0, body_real_xyz=pos, robot_body=NS(get_position=lambda: pos),
                   calc_potential=lambda: -wtd / 0.0165, calc_state=lambda: s_new.copy())
        self = AntFlagrunBulletEnv.__new__(AntFlagrunBulletEnv)
        self.__dict__.update(dict(robot=robot, tol=tol, timeout=timeout, switch_flag_on_collision=switch, max_targets=100, goals=list(goals),
                                  steps_since_goal_change=steps0, _rewarded=rewarded0, debug=False, use_sensor=use_sensor, n_bins=n_bins,
                                  sensor_span=span, sensor_range=srange, scene=scene, robot_body=Body(list(pos), [0.01, -0.02, yaw]),
                                  isRender=False, flag=None, walk_target_x=1.0, walk_target_y=2.0, _sq_dist_goal=3.0,
                                  _goal_start_pos=np.array([0.5, 0.5]), potential=-77.0, _super_step_result=(s_old.copy(), inner_r, inner_d, {})))
        obs, r, d, info = AntFlagrunBulletEnv.step(self, np.zeros(8))
        return {'size': size, 'n_goals': n_goals, 'last_goal': list(map(float, goals[-1])) if goals else None, 'tol': tol, 'timeout': timeout, 'switch': switch,
                'steps_before': steps0, 'rewarded_before': rewarded0, 'walk_target_dist': wtd, 'inner_rew': inner_r, 'inner_done': inner_d,
                'use_sensor': use_sensor, 'n_bins': n_bins, 'span': span, 'range': srange, 'pos': [float(pos[0]), float(pos[1])], 'yaw': yaw,
                'arena_bounds': bounds_list(scene), 'rew': float(r), 'done': bool(d), 'steps_after': int(self.steps_since_goal_change),
                'rewarded_after': bool(self._rewarded), 'goals_left': len(self.goals), 'retargeted': bool('target' in info),
                'state_is_new': bool(np.array_equal(np.asarray(obs)[:28], s_new)), 'sensor': tolist(np.asarray(obs)[28:]),
                'target_after': [float(self.walk_target_x), float(self.walk_target_y)]}

    def random_maze_mj_case(k):
        """ant_maze_mj_env.py:57-78 with n_bins / sensor span and range / tol / inner_rew_weight drawn at random"""
        lrs = np.random.RandomState(26000 + k)
        n_bins = int(lrs.choice([2, 3, 10, 16, 17, 33, 64]))
        span = float(lrs.choice([2 * np.pi, np.pi, f32(lrs.uniform(0.3, 6.5))])); srange = f32(lrs.choice([2.0, 5.0, 9.0, 30.0]))
        tol, w = f32(lrs.choice([1.5, 0.2, 0.8, 3.0, 12.0])), f32(lrs.choice([0.0, 0.5, 1.0, -2.0]))
        target = np.array([f32(lrs.uniform(-4.5, 4.5)), f32(lrs.uniform(-8.5, 8.5))])
        xy = np.array([lrs.uniform(-4.5, 4.5), lrs.uniform(-8.5, 8.5)])
        if lrs.rand() < 0.5:
            xy = target + lrs.uniform(-1.0, 1.0, 2) * min(tol, 3.0)
        state = lrs.uniform(-1, 1, 29)
        state[0], state[1], state[2] = xy[0], xy[1], lrs.uniform(0.2, 0.8)
        rpy = [lrs.uniform(-.2, .2), lrs.uniform(-.2, .2), lrs.uniform(-np.pi, np.pi)]
        pot_old = float(lrs.uniform(-700, -100)); pot_new = pot_old + float(lrs.uniform(-3, 3))
        n_lim, t0 = int(lrs.randint(0, 4)), int(lrs.randint(0, 500))
        wtd = float(np.linalg.norm(target - (xy + lrs.uniform(-0.3, 0.3, 2))))
        robot = NS(apply_action=lambda a: None, calc_state=lambda: state.copy(), initial_z=0.25, body_rpy=rpy, calc_potential=lambda: pot_new, feet=[],
                   feet_contact=np.zeros(4), joints_at_limit=n_lim, walk_target_dist=wtd)
        robot.alive_bonus = lambda z, pitch: MjAnt.alive_bonus(robot, z, pitch)
        self = _AntMazeMjEnv.__new__(_AntMazeMjEnv)
        self.__dict__.update(dict(robot=robot, scene=maze, potential=pot_old, joints_at_limit_cost=-0.1, ground_ids=set(), reward=0.0, n_bins=n_bins,
                                  sensor_span=span, sensor_range=srange, targets=[list(target)], tol=tol, inner_rew_weight=w, t=t0, target=target, debug=0,
                                  robot_body=Body([xy[0], xy[1], state[2]], rpy)))
        maze.global_step = lambda: None
        obs, rew, d, _ = _AntMazeMjEnv.step(self, np.zeros(8))
        return {'n_bins': n_bins, 'span': span, 'range': srange, 'tol': tol, 'state': state.tolist(), 'rpy': list(map(float, rpy)), 'target': target.tolist(),
                't_before': t0, 'potential_old': pot_old, 'potential_new': pot_new, 'joints_at_limit': n_lim, 'walk_target_dist': wtd,
                'inner_rew_weight': w, 'obs': tolist(obs), 'rew': float(rew), 'done': bool(d), 't_after': int(self.t)}

    from hrl_pybullet_envs.envs.ant_maze.ant_maze_mj_env import AntMazeMjEnv as _AntMazeMjEnv
    G['random_config'] = {'gather_step': [random_gather_case(k) for k in range(70)], 'maze_step': [random_maze_case(k) for k in range(70)],
                          'flagrun_step': [random_flagrun_case(k) for k in range(80)], 'maze_mj_step': [random_maze_mj_case(k) for k in range(50)]}

    def random_scene(k):
        """gather_scene.py:38-62,95-114: spawn / restart / reward_collision with world size, item counts, spacing and respawn drawn at random"""
        lrs = np.random.RandomState(27000 + k)
        world = (f32(lrs.uniform(5, 26)), f32(lrs.uniform(5, 26)))
        nf, npo = int(lrs.choice([0, 1, 8, 17, 30])), int(lrs.choice([0, 1, 8, 17, 30]))
        if nf + npo == 0: nf = 1
        spacing = f32(lrs.uniform(0.3, min(world) / 3.2))
        hits = [(int(lrs.randint(0, nf + npo + 1)), (float(lrs.uniform(-world[0] / 2, world[0] / 2)), float(lrs.uniform(-world[1] / 2, world[1] / 2)), 0.5)) for _ in range(6)]
        return scene_run(27500 + k, world, nf, npo, spacing, bool(lrs.rand() < 0.7), hits)

    G['random_config']['gather_scene'] = [random_scene(k) for k in range(14)]

    # ---------------------------------------------------------------- the constructor surfaces and the registration (the boundary)
    # inspect.signature of every env class a user constructs, and the keyword arguments hrl_pybullet_envs/__init__.py:11-16 registered with gym
    import enum
    import inspect
    from hrl_pybullet_envs.envs.gather.point_gather_env import PointGatherBulletEnv
    from hrl_pybullet_envs.envs.ant_flagrun.ant_flagrun_env import AntFlagrunBulletEnv
    from hrl_pybullet_envs.envs.ant_maze.ant_maze_mj_env import AntMazeMjEnv as _AntMazeMjEnv

    def plain(v):
        if isinstance(v, enum.Enum): return {'enum': type(v).__name__, 'name': v.name}
        if isinstance(v, (tuple, list)): return [plain(x) for x in v]
        if isinstance(v, (bool, int, str)) or v is None: return v
        if isinstance(v, (float, np.floating, np.integer)): return float(v)
        return {'repr': repr(v)}

    def signature(cls):
        return [{'name': n, 'kind': q.kind.name, 'default': '<required>' if q.default is inspect.Parameter.empty else plain(q.default)}
                for n, q in inspect.signature(cls.__init__).parameters.items() if n != 'self']

    G['constructor_signatures'] = {
        'classes': {f'{c.__module__}:{c.__name__}': signature(c) for c in (AntGatherBulletEnv, PointGatherBulletEnv, GatherBulletEnv, AntMazeBulletEnv,
                                                                         _AntMazeMjEnv, AntFlagrunBulletEnv, AntMjEnv, PointBot, MjAnt)},
        'registered': [{k: plain(v) for k, v in r.items()} for r in REGISTERED]}

    # ------------------------------------------------------------------------------------------ assets (data files, parsed; no Python imported)
    # The numbers of hrl_pybullet_envs/assets/*.xml that decide the rigid-body MODEL (geometry, joint frames and ranges, obstacle sizes): the one
    # part of the physics the reference tree itself holds.  tests/test_assets.py builds the ant from them with a generic MJCF forward
    # kinematics and compares it with the oracle's / the textbook reference's hard-wired ant.
    import xml.etree.ElementTree as ET
    adir = os.path.join(REF, 'hrl_pybullet_envs', 'assets')

    def nums(t):
        return [float(x) for x in t.split()]

    def mj_body(b):
        return {'name': b.get('name'), 'pos': nums(b.get('pos', '0 0 0')),
                'joints': [{'name': j.get('name'), 'type': j.get('type'), 'axis': nums(j.get('axis', '0 0 1')), 'pos': nums(j.get('pos', '0 0 0')),
                            'range': nums(j.get('range')) if j.get('range') else None} for j in b.findall('joint')],
                'geoms': [{'name': g.get('name'), 'type': g.get('type'), 'size': nums(g.get('size')), 'pos': nums(g.get('pos', '0 0 0')),
                           'fromto': nums(g.get('fromto')) if g.get('fromto') else None,
                           **{k: (nums(g.get(k)) if k == 'friction' else float(g.get(k))) for k in ('mass', 'friction', 'condim') if g.get(k)}}
                          for g in b.findall('geom')],
                'bodies': [mj_body(c) for c in b.findall('body')]}

    ant = ET.parse(os.path.join(adir, 'ant.xml')).getroot()
    comp, dflt = ant.find('compiler'), ant.find('default')
    cube = ET.parse(os.path.join(adir, 'player_cube.xml')).getroot()

    def urdf_box(fn):
        link = ET.parse(os.path.join(adir, fn)).getroot().find('link')
        return {'collision_box_size': nums(link.find('collision/geometry/box').get('size')), 'mass': float(link.find('inertial/mass').get('value'))}

    G['assets'] = {
        'ant': {'compiler': dict(comp.attrib), 'default_joint': dict(dflt.find('joint').attrib), 'default_geom': dict(dflt.find('geom').attrib),
                'init_qpos': nums(ant.find('custom/numeric').get('data')),
                'torso': mj_body(ant.find('worldbody/body')),
                'actuators': [{'joint': m.get('joint'), 'gear': float(m.get('gear')), 'ctrlrange': nums(m.get('ctrlrange'))} for m in ant.findall('actuator/motor')]},
        'player_cube': mj_body(cube.find('worldbody/body')),
        'box': urdf_box('box.xml'), 'food': urdf_box('food.xml'), 'poison': urdf_box('poison.xml'), 'wall': urdf_box('wall.xml'), 'plane': urdf_box('plane.xml')}

    only = set(sys.argv[1:])  # optional: names of the fixtures to (re)write; default all
    for name, val in G.items():
        if only and name not in only:
            continue
        with open(os.path.join(OUT_DIR, name + '.json'), 'w') as f:
            json.dump(val, f, allow_nan=True)
        print(name, os.path.getsize(os.path.join(OUT_DIR, name + '.json')), 'bytes')


if __name__ == '__main__':
    main()
