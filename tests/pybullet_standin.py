"""Oracle-backed stand-ins for `gym`, `pybullet`, `pybullet_envs` and `hrl_pybullet_envs` -- for ONE purpose: a DRY RUN of the road that pins the
rigid-body half of the oracle (tools/make_pybullet_golden.py -> tests/test_pybullet_golden.py -> tests/tools/fit_model.py), so that whoever runs
it where the real packages exist does not execute ~300 lines for the first time.  Test infrastructure only.

What is stood in for: the pybullet GETTERS the generator calls (getBasePositionAndOrientation, getBaseVelocity, getJointState, getJointInfo,
getDynamicsInfo, getBodyInfo, getCollisionShapeData, getContactPoints, getNumBodies, getNumJoints, getPhysicsEngineParameters -- tuple layouts as
pybullet's quickstart guide documents them, restated from memory: SURVEY Appendix A, unverified), `gym.make` with its TimeLimit wrapper, and env
objects that carry the attributes of the reference's classes the generator reads (`unwrapped`, `_p`, `robot.robot_body.bodies / bodyIndex`,
`stadium_scene.food / poison` -- gather_scene.py:23-33 --, `target` -- ant_maze_bullet_env.py:46,110 --, `walk_target_x / y`, `potential`,
`steps_since_goal_change`, `_rewarded`, `goals`, `_sq_dist_goal`, `_goal_start_pos` -- ant_flagrun_env.py:41-52 --, `robot.initial_z`,
`robot.feet_contact`, `robot.walk_target_dist` ...).  Everything they answer comes from the fp64 CPU oracle (tests/orc.py) stepping ONE env under
a PERTURBED `hrl_model` (HRL_STANDIN_MODEL: JSON of model fields): the "engine" the dry run has to find again.

What a green dry run says: generator, replay and fit agree with one another on every convention the records carry (state order, quaternion xyzw,
joint order, world-frame velocities, item order, the task state a step reads besides (qpos, qvel)) and the fit recovers a model from the generator's
own JSON.  What it does NOT say: anything about pybullet.  Records made here carry `versions.standin` and are refused as fixtures
(tests/test_pybullet_golden.py); the dry run writes to tmp dirs only."""
import json
import math
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import orc  # noqa: E402
from hrl_pybullet_envs_amd import _capi as K  # noqa: E402

JOINTS = ['hip_1', 'ankle_1', 'hip_2', 'ankle_2', 'hip_3', 'ankle_3', 'hip_4', 'ankle_4']        # assets/ant.xml:18-54
LEGS = ['front_left_leg', 'front_right_leg', 'left_back_leg', 'right_back_leg']                 # assets/ant.xml:15,26,37,48: jointless bodies -> fixed links
AUX = ['aux_1', 'aux_2', 'aux_3', 'aux_4']
FEET = ['front_left_foot', 'front_right_foot', 'left_back_foot', 'right_back_foot']
LO = np.radians([-40, 30, -40, -100, -40, -100, -40, 30])
HI = np.radians([40, 100, 40, -30, 40, -30, 40, 100])
KINDS = {'AntGatherBulletEnv': K.HRL_ANT_GATHER, 'AntMazeBulletEnv': K.HRL_ANT_MAZE, 'PointGatherBulletEnv': K.HRL_POINT_GATHER,
         'AntFlagrunBulletEnv': K.HRL_ANT_FLAGRUN, 'AntMazeMjEnv': K.HRL_ANT_MAZE_MJ, 'AntMjEnv': K.HRL_ANT_FLAT}
REGISTERED = ['AntGatherBulletEnv', 'AntMazeBulletEnv', 'PointGatherBulletEnv', 'AntFlagrunBulletEnv', 'AntMazeMjEnv']   # hrl_pybullet_envs/__init__.py:9 (AntMjEnv is not)
MAZE_KINDS = (K.HRL_ANT_MAZE, K.HRL_ANT_MAZE_MJ)
JOINT_REVOLUTE, JOINT_FIXED = 0, 4   # pybullet.JOINT_REVOLUTE / JOINT_FIXED
GEOM_SPHERE, GEOM_BOX, GEOM_CAPSULE = 2, 3, 7


def standin_model():
    """the perturbed engine: hrl_model fields from HRL_STANDIN_MODEL (JSON), {} = the default specification"""
    return json.loads(os.environ.get('HRL_STANDIN_MODEL', '{}'))


def quat_to_rpy(q):
    x, y, z, w = q
    return [math.atan2(2 * (w * x + y * z), 1 - 2 * (x * x + y * y)), math.asin(max(-1.0, min(1.0, 2 * (w * y - z * x)))),
            math.atan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z))]


class Client:
    """the BulletClient of one env (`env._p`): getters over the oracle's records"""

    def __init__(self, env):
        self.e = env
        c = env.o.cfg
        st = []   # static bodies in load order: floor, walls, maze box, item cubes (sizeable_enclosed_scene.py:39-61, maze_scene.py:33-38, gather_scene.py:41-44)
        st.append(('floor', [0, 0, 0], [25, 25, 0.005]))
        hx = hy = 0.0
        if env.kind in (K.HRL_ANT_GATHER, K.HRL_POINT_GATHER) or (env.kind == K.HRL_ANT_FLAGRUN and (c.flag_enclosed or c.use_sensor)):
            hx, hy = c.world_size[0] / 2, c.world_size[1] / 2
        if env.kind in MAZE_KINDS:
            hx, hy = 5.0, 9.0
        if hx:
            st += [('wall', [0, hy, 2.5], [25, 0.05, 2.5]), ('wall', [0, -hy, 2.5], [25, 0.05, 2.5]), ('wall', [hx, 0, 2.5], [0.05, 25, 2.5]), ('wall', [-hx, 0, 2.5], [0.05, 25, 2.5])]
        if env.kind in MAZE_KINDS:
            st.append(('obstacle', [-2, 0, 1], [3, 2, 1]))
        self.statics = st
        self.n_items = c.n_food + c.n_poison if env.kind in (K.HRL_ANT_GATHER, K.HRL_POINT_GATHER) else 0
        self.item0 = len(st)
        self.robot = self.item0 + self.n_items

    # ---- bodies
    def getNumBodies(self):
        return self.robot + 1

    def getBodyInfo(self, b):
        if b == self.robot:
            return (b'torso' if self.e.ant else b'base', b'ant' if self.e.ant else b'player_cube')
        if b >= self.item0:
            return (b'food' if b - self.item0 < self.e.o.cfg.n_food else b'poison', b'cube')
        return (self.statics[b][0].encode(), self.statics[b][0].encode())

    def getBasePositionAndOrientation(self, b):
        if b == self.robot:
            s = self.e.o.state[0]
            return tuple(float(v) for v in s[0:3]), tuple(float(v) for v in s[3:7])   # quaternion x, y, z, w
        if b >= self.item0:
            xy = self.e.o.items[0, 2 * (b - self.item0):2 * (b - self.item0) + 2]
            return (float(xy[0]), float(xy[1]), 0.1), (0.0, 0.0, 0.0, 1.0)
        return tuple(float(v) for v in self.statics[b][1]), (0.0, 0.0, 0.0, 1.0)

    def getBaseVelocity(self, b):
        assert b == self.robot
        s = self.e.o.state[0]
        return tuple(float(v) for v in s[15:18]), tuple(float(v) for v in s[18:21])   # linear, angular: world frame

    # ---- the ant's links: per leg a jointless body on a fixed joint, the aux body on `hip`, the foot on `ankle` (link-index order)
    def getNumJoints(self, b):
        return 12 if (b == self.robot and self.e.ant) else 0

    def _joint(self, j):
        leg, part = divmod(j, 3)
        if part == 0:
            return 'jointfix_%d_%d' % (leg, j), JOINT_FIXED, LEGS[leg], None
        k = 2 * leg + (part - 1)
        return JOINTS[k], JOINT_REVOLUTE, (AUX if part == 1 else FEET)[leg], k

    def getJointInfo(self, b, j):
        name, typ, link, k = self._joint(j)
        m = self.e.o.cfg.model
        sx, sy = ((1, 1), (-1, 1), (-1, -1), (1, -1))[j // 3]
        axis = (0.0, 0.0, 0.0) if k is None else ((0.0, 0.0, 1.0) if k % 2 == 0 else tuple(v / math.sqrt(2) for v in ((-1, 1, 0), (1, 1, 0), (-1, 1, 0), (1, 1, 0))[j // 3]))
        lo, hi = (0.0, -1.0) if k is None else (float(LO[k]), float(HI[k]))
        return (j, name.encode(), typ, -1 if k is None else 7 + k, -1 if k is None else 6 + k, 1, float(m.joint_damping) if k is not None else 0.0, 0.0, lo, hi,
                0.0, float(m.max_joint_vel), link.encode(), axis, (0.2 * sx, 0.2 * sy, 0.0) if j % 3 else (0.0, 0.0, 0.0), (0.0, 0.0, 0.0, 1.0), j - 1 if j % 3 else -1)

    def getJointState(self, b, j):
        k = self._joint(j)[3]
        s = self.e.o.state[0]
        if k is None:
            return (0.0, 0.0, (0.0,) * 6, 0.0)
        return (float(s[7 + k]), float(s[21 + k]), (0.0,) * 6, float(self.e.last_tau[k]))

    # ---- model
    def _link_mass(self, link):
        rho, rc = float(self.e.o.cfg.model.density), 0.08
        cap = lambda L: rho * (math.pi * rc * rc * L + 4.0 / 3.0 * math.pi * rc ** 3)
        if not self.e.ant:
            return 10.0   # player_cube.xml:8
        if link < 0:
            return rho * 4.0 / 3.0 * math.pi * 0.25 ** 3
        return cap(0.4 * math.sqrt(2)) if link % 3 == 2 else cap(0.2 * math.sqrt(2))

    def getDynamicsInfo(self, b, link):
        m = self.e.o.cfg.model
        if b != self.robot:
            return (0.0, float(m.friction_ground), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0, 1.0), 0.5, 0.0, 0.0, -1.0, -1.0, 1, 0.0)
        mass = self._link_mass(link)
        r = 0.25 if link < 0 else 0.08
        i = 0.4 * mass * r * r   # (a sphere's: the stand-in reports a diagonal, the dry run reads only the masses)
        return (mass, float(m.friction_robot), (i, i, i), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0, 1.0), 0.0, 0.0, 0.0, -1.0, -1.0, 1, 0.0)

    def getCollisionShapeData(self, b, link):
        if b != self.robot:
            if b >= self.item0:
                return [(b, -1, GEOM_BOX, (0.25, 0.25, 0.25), b'', (0.0, 0.0, 0.0), (0.0, 0.0, 0.0, 1.0))]
            return [(b, -1, GEOM_BOX, tuple(2 * v for v in self.statics[b][2]), b'', (0.0, 0.0, 0.0), (0.0, 0.0, 0.0, 1.0))]
        if not self.e.ant:
            return [(b, -1, GEOM_BOX, (0.7, 0.7, 0.7), b'', (0.0, 0.0, 0.0), (0.0, 0.0, 0.0, 1.0))]
        if link < 0:
            return [(b, -1, GEOM_SPHERE, (0.25, 0.25, 0.25), b'', (0.0, 0.0, 0.0), (0.0, 0.0, 0.0, 1.0))]
        L = 0.4 * math.sqrt(2) if link % 3 == 2 else 0.2 * math.sqrt(2)
        return [(b, link, GEOM_CAPSULE, (L, 0.08, 0.0), b'', (0.0, 0.0, 0.0), (0.0, 0.0, 0.0, 1.0))]

    def getPhysicsEngineParameters(self):
        m = self.e.o.cfg.model
        return {'fixedTimeStep': float(m.timestep) * m.frame_skip, 'numSubSteps': int(m.frame_skip), 'numSolverIterations': int(m.solver_iters),
                'useRealTimeSimulation': 0, 'gravityAccelerationX': 0.0, 'gravityAccelerationY': 0.0, 'gravityAccelerationZ': -float(m.gravity),
                'numNonContactInnerIterations': 1, 'contactERP': float(m.contact_erp), 'erp': float(m.limit_erp), 'frictionERP': 0.2}

    def getContactPoints(self, bodyA=-1, **kw):
        """one record per foot whose flag the last step set (the oracle's batch interface hands out the flags, not the manifold): enough to run the
        generator's tuple unpacking; nothing downstream compares contacts"""
        out = []
        for l, f in enumerate(self.e.feet_now):
            if f:
                out.append((0, bodyA, 0, 3 * l + 2, -1, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (0.0, 0.0, 1.0), 0.0, 1.0, 0.0, (1.0, 0.0, 0.0), 0.0, (0.0, 1.0, 0.0)))
        return out


class BodyPart:
    def __init__(self, body):
        self.bodies, self.bodyIndex = [body], 0


class Robot:
    """`env.robot`: what the generator (and the reference's step code) reads off upstream's WalkerBase / the PointBot"""

    def __init__(self, env):
        self.e = env
        self.robot_body = BodyPart(env._p.robot)

    initial_z = property(lambda self: float(self.e.o.state[0, K.HRL_INITZ_OFF]))
    body_rpy = property(lambda self: quat_to_rpy(self.e.o.state[0, 3:7]))
    body_real_xyz = property(lambda self: [float(v) for v in self.e.o.state[0, 0:3]])
    body_xyz = property(lambda self: [float(v) for v in self.e.o.state[0, 0:3]])   # (upstream: parts centroid x, y; recorded, never replayed)
    walk_target_x = property(lambda self: float(self.e.goal_xy()[0]))
    walk_target_y = property(lambda self: float(self.e.goal_xy()[1]))

    @property
    def feet_contact(self):   # what the NEXT step's calc_state() will read: the flags the last step left (bits 28..31 of aux[1]; zeros after a reset)
        return np.array(self.e.feet_now, np.float32)

    @property
    def walk_target_dist(self):   # upstream: from the parts centroid; the oracle keeps it as potential = -dist / dt
        m = self.e.o.cfg.model
        return float(-self.e.o.state[0, K.HRL_POTENTIAL_OFF] * m.timestep * m.frame_skip)

    @property
    def joints_at_limit(self):
        q = self.e.o.state[0, 7:15]
        return int((np.abs(2 * (q - 0.5 * (LO + HI)) / (HI - LO)) > 0.99).sum()) if self.e.ant else 0


class Scene:
    """GatherScene's `food` / `poison`: {pybullet body id: [x, y, z]} in spawn order (gather_scene.py:23-24,66-75)"""

    def __init__(self, env):
        self.e = env

    def _dict(self, lo, n):
        it, b0 = self.e.o.items[0], self.e._p.item0
        return {b0 + i: [float(it[2 * i]), float(it[2 * i + 1]), 0.1] for i in range(lo, lo + n)}

    food = property(lambda self: self._dict(0, self.e.o.cfg.n_food))
    poison = property(lambda self: self._dict(self.e.o.cfg.n_food, self.e.o.cfg.n_poison))
    all_items = property(lambda self: {**self.food, **self.poison})


class Space:
    def __init__(self, shape):
        self.shape = tuple(shape)


class StandinEnv:
    """One env of the reference's class `name`, stepped by the fp64 oracle under the stand-in's model."""

    def __init__(self, name):
        self.name, self.kind = name, KINDS[name]
        self.ant = self.kind != K.HRL_POINT_GATHER
        self._seed = 0
        self._make()
        self.action_space = Space((self.o.ad,))
        self.observation_space = Space((self.o.od,))

    def _make(self):
        kw = {'model_' + k: v for k, v in standin_model().items()}
        cfg = orc.default_config(self.kind, num_envs=1, seed=self._seed, auto_reset=0, max_episode_steps=0, **kw)
        old = getattr(self, 'o', None)
        self.o = orc.OracleEnv(cfg, np.float64)
        if old is not None:
            self.o.aux[...] = old.aux   # the episode counters carry on (the streams are keyed by them)
        self._p = Client(self)
        self.robot = Robot(self)
        self.last_tau = np.zeros(8)
        if self.kind in (K.HRL_ANT_GATHER, K.HRL_POINT_GATHER):
            self.stadium_scene = self.scene = Scene(self)

    unwrapped = property(lambda self: self)

    def seed(self, s=None):
        self._seed = 0 if s is None else int(s)
        self.o.cfg.seed = self._seed
        return [s]

    def reset(self):
        return self.o.reset()[0].copy()

    def step(self, a):
        a = np.asarray(a, np.float64).reshape(1, -1)
        if self.ant:
            self.last_tau = float(self.o.cfg.model.torque_scale) * np.clip(a[0], -1, 1)
        obs, rew, done, info = self.o.step(a)
        out = {}
        if self.kind in (K.HRL_ANT_GATHER, K.HRL_POINT_GATHER):
            out = {'food_rew': float(info[0, 0]), 'dead_rew': float(info[0, 1])}   # ant_gather_env.py:119
        if self.kind == K.HRL_ANT_FLAGRUN and self.o.goal[0, 2] != 0:
            out['target'] = (float(self.o.goal[0, 0]), float(self.o.goal[0, 1]))   # ant_flagrun_env.py:191,199
        return obs[0].copy(), float(rew[0]), bool(done[0]), out

    def close(self):
        pass

    # ---- what the reference's classes carry between steps
    @property
    def feet_now(self):
        bits = (int(self.o.aux[0, 1]) >> 28) & 0xf if self.kind in (K.HRL_ANT_MAZE, K.HRL_ANT_FLAGRUN) else 0
        return [(bits >> l) & 1 for l in range(4)]

    def goal_xy(self):
        c = self.o.cfg
        if self.kind in MAZE_KINDS:
            return np.array(c.targets[int(self.o.aux[0, 3])][:], np.float64)
        if self.kind == K.HRL_ANT_FLAGRUN:
            g = np.zeros(2)
            orc.lib().orc_flag_goal_f64(orc.C.byref(c), int(self.o.aux[0, 2]), int(self.o.aux[0, 3]) & 0xffff, orc.ptr(g))
            return g
        return np.array([c.walk_target[0], c.walk_target[1]], np.float64)

    potential = property(lambda self: float(self.o.state[0, K.HRL_POTENTIAL_OFF]))
    walk_target_x = property(lambda self: float(self.goal_xy()[0]))
    walk_target_y = property(lambda self: float(self.goal_xy()[1]))

    def __getattr__(self, k):   # attributes only some of the reference's classes have (hasattr() in the generator must say so)
        kind = self.__dict__.get('kind')
        if 'o' not in self.__dict__:
            raise AttributeError(k)
        if kind in MAZE_KINDS:
            if k == 'target':
                return self.goal_xy()
            if k == 't':
                return int(self.o.aux[0, 0])
        if kind == K.HRL_ANT_FLAGRUN:
            a3 = int(self.o.aux[0, 3]) & 0xffffffff
            if k == 'steps_since_goal_change':
                return (a3 >> 16) & 0x7fff
            if k == '_rewarded':
                return bool(a3 >> 31)
            if k == 'goal':
                return tuple(float(v) for v in self.goal_xy())
            if k == 'goals':   # the pending list, popped from its end (ant_flagrun_env.py:91-96,116): the shared list's goals this episode has not used yet
                c, used = self.o.cfg, a3 & 0xffff
                out = []
                for j in range(c.flag_max_targets, used, -1):
                    g = np.zeros(2)
                    orc.lib().orc_flag_goal_f64(orc.C.byref(c), int(self.o.aux[0, 2]), j, orc.ptr(g))
                    out.append((float(g[0]), float(g[1])))
                return out
            if k == '_sq_dist_goal':
                return float(self.o.items[0, K.HRL_FLAG_SQDIST_OFF])
            if k == '_goal_start_pos':
                return np.array(self.o.items[0, K.HRL_FLAG_START_OFF:K.HRL_FLAG_START_OFF + 2])
        raise AttributeError(k)


class TimeLimit:
    """gym.wrappers.TimeLimit of gym <= 0.21, what gym.make wraps a registered env in (hrl_pybullet_envs/__init__.py:15)"""

    def __init__(self, env, max_episode_steps):
        self.env, self._max_episode_steps, self._elapsed_steps = env, max_episode_steps, None
        self.action_space, self.observation_space = env.action_space, env.observation_space

    unwrapped = property(lambda self: self.env.unwrapped)

    def seed(self, s=None):
        return self.env.seed(s)

    def reset(self):
        self._elapsed_steps = 0
        return self.env.reset()

    def step(self, a):
        ob, r, d, i = self.env.step(a)
        self._elapsed_steps += 1
        if self._elapsed_steps >= self._max_episode_steps:
            i['TimeLimit.truncated'] = not d
            d = True
        return ob, r, d, i

    def close(self):
        self.env.close()


def install():
    """puts the stand-in modules into sys.modules (the real ones must be absent: a box that has pybullet runs the generator for real)"""
    for m in ('pybullet', 'gym', 'pybullet_envs', 'hrl_pybullet_envs'):
        assert m not in sys.modules, m
    registry = {}

    def register(id, entry_point=None, max_episode_steps=None, **kw):   # noqa: A002
        registry[id] = (entry_point, max_episode_steps)

    def make(id, **kw):   # noqa: A002
        entry, limit = registry[id]
        env = entry(**kw)
        return TimeLimit(env, limit) if limit else env

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    envs = mod('gym.envs', register=register)
    mod('gym', make=make, envs=envs, __version__='0.21.0+standin')
    mod('pybullet', getAPIVersion=lambda: 'standin (tests/pybullet_standin.py: the fp64 CPU oracle, model %s)' % json.dumps(standin_model(), sort_keys=True),
        JOINT_REVOLUTE=JOINT_REVOLUTE, JOINT_FIXED=JOINT_FIXED)
    mod('pybullet_envs')
    h = mod('hrl_pybullet_envs', standin=True)
    for name in REGISTERED:   # hrl_pybullet_envs/__init__.py:9-16: `<ClassName>-v0`, max_episode_steps=2000
        cls = (lambda n: (lambda **kw: StandinEnv(n)))(name)
        setattr(h, name, cls)
        register(id=name + '-v0', entry_point=cls, max_episode_steps=2000)
    mod('hrl_pybullet_envs.envs')
    mod('hrl_pybullet_envs.envs.MjAnt', AntMjEnv=lambda: StandinEnv('AntMjEnv'))   # envs/MjAnt.py:31-34: constructed directly, never registered


def main():
    """python tests/pybullet_standin.py <script.py> [args...]: runs a script -- the generator -- with the stand-ins installed"""
    import runpy
    install()
    script = sys.argv[1]
    sys.argv = sys.argv[1:]
    runpy.run_path(script, run_name='__main__')


if __name__ == '__main__':
    main()
