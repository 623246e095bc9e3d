"""The robot attributes a trainer written against the reference reads between steps -- `env.robot.walk_target_x/y`,
`body_xyz`, `body_real_xyz`, `body_rpy`, `walk_target_dist`, `initial_z`, `joint_speeds`, `joints_at_limit`, `feet_contact`, `calc_potential()`
(upstream WalkerBase; point_bot.py:48-67; read in-tree at MjAnt.py:51-68) and `robot_body.pose().xyz()` / `.rpy()` /
`.get_position()` (read in-tree at ant_maze_bullet_env.py:67,125,130, ant_flagrun_env.py:104-105,128) -- as read-only views over the env's state tensors.  Single env: python / numpy values like the reference;
batched: one row per env (numpy, fp64, computed on the host from one copy of the state: informational, not the step path)."""
import numpy as np

from .. import _capi as K

_IS2 = 0.70710678118654752440
_L1, _L2 = 0.2 * np.sqrt(2.0), 0.4 * np.sqrt(2.0)  # assets/ant.xml:19,22


def quat_axes(q):
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    X = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y + w * z), 2 * (x * z - w * y)], 1)
    Y = np.stack([2 * (x * y - w * z), 1 - 2 * (x * x + z * z), 2 * (y * z + w * x)], 1)
    Z = np.stack([2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)], 1)
    return X, Y, Z


def quat_to_rpy(q):
    """pybullet getEulerFromQuaternion (as restated in csrc/step_core.h quat_to_rpy)."""
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    sarg = -2 * (x * z - w * y)
    roll = np.arctan2(2 * (y * z + w * x), w * w - x * x - y * y + z * z)
    pitch = np.arcsin(np.clip(sarg, -1, 1))
    yaw = np.arctan2(2 * (x * y + w * z), w * w + x * x - y * y - z * z)
    lo, hi = sarg <= -0.99999, sarg >= 0.99999
    roll = np.where(lo | hi, 0.0, roll)
    pitch = np.where(lo, -np.pi / 2, np.where(hi, np.pi / 2, pitch))
    yaw = np.where(lo, 2 * np.arctan2(x, -y), np.where(hi, 2 * np.arctan2(-x, y), yaw))
    return np.stack([roll, pitch, yaw], 1)


def ant_parts_centroid(qpos, n_static, static_sum):
    """xy of upstream's `parts` centroid: the mean over the 13 ant bodies (torso, and per leg the jointless
    capsule, the aux and the foot capsule, each at its capsule's midpoint) and the static bodies that share the dict."""
    qpos = np.asarray(qpos, np.float64)
    X, Y, Z = quat_axes(qpos[:, 3:7])
    s = np.zeros((qpos.shape[0], 3))
    for l in range(4):  # assets/ant.xml:15-58
        sx, sy = (1.0 if l in (0, 3) else -1.0), (1.0 if l < 2 else -1.0)
        sg = 1.0 if l in (1, 2) else -1.0
        qh, qa = qpos[:, 7 + 2 * l], qpos[:, 8 + 2 * l]
        ch, sh, ca, sa = np.cos(qh), np.sin(qh), np.cos(qa), np.sin(qa)
        e1x, e1y = (sx * ch - sy * sh) * _IS2, (sx * sh + sy * ch) * _IS2
        e1 = e1y[:, None] * Y + e1x[:, None] * X
        e2 = (sg * sa)[:, None] * Z + ca[:, None] * e1
        ph = 0.2 * (sy * Y + sx * X)
        pa = ph + _L1 * e1
        tip = pa + _L2 * e2
        s += 0.5 * ph + 0.5 * (ph + pa) + 0.5 * (pa + tip)
    n = 13.0 + n_static
    return np.stack([(13 * qpos[:, 0] + s[:, 0] + static_sum[0]) / n, (13 * qpos[:, 1] + s[:, 1] + static_sum[1]) / n], 1)


class PoseView:
    """upstream robot_bases.Pose_Helper: what `robot_body.pose()` returns"""

    def __init__(self, body):
        self._b = body

    def xyz(self):
        return self._b.get_position()

    def rpy(self):
        return self._b._r.body_rpy

    def orientation(self):
        return self._b.get_orientation()


class BodyView:
    """upstream robot_bases.BodyPart of the torso (`robot.robot_body`, `env.robot_body`): position, orientation (quaternion x y z w), velocity"""

    def __init__(self, robot):
        self._r = robot

    def pose(self):
        return PoseView(self)

    def get_position(self):
        return self._r.body_real_xyz

    current_position = get_position

    def get_orientation(self):
        return self._r._out(self._r._state()[:, 3:7])

    current_orientation = get_orientation

    def get_pose(self):
        return np.concatenate([np.atleast_2d(self.get_position()), np.atleast_2d(self.get_orientation())], 1)[() if self._r._e.num_envs > 1 else 0]

    def speed(self):
        return self._r._out(self._r._state()[:, K.HRL_QVEL_OFF:K.HRL_QVEL_OFF + 3])


_ANT_LO = np.deg2rad([-40.0, 30, -40, -100, -40, -100, -40, 30])   # assets/ant.xml:18-54, tree order
_ANT_HI = np.deg2rad([40.0, 100, 40, -30, 40, -30, 40, 100])


class RobotView:
    def __init__(self, env):
        self._e = env

    @property
    def robot_body(self):
        return BodyView(self)

    def _ant(self, what):
        if self._e._cfg.env_kind == K.HRL_POINT_GATHER:
            raise AttributeError(f'{what}: the PointBot has no joints (point_bot.py:10-74)')

    @property
    def joint_speeds(self):
        """upstream WalkerBase.calc_state: `j[1::2]`, the joint rates scaled by 0.1 (what the electricity cost multiplies, MjAnt.py:65)"""
        self._ant('joint_speeds')
        return self._out(0.1 * self._state()[:, K.HRL_QVEL_OFF + 6:K.HRL_QVEL_OFF + 14])

    @property
    def joints_at_limit(self):
        """upstream WalkerBase.calc_state: how many joints are beyond 99 % of their half range (MjAnt.py:40,68)"""
        self._ant('joints_at_limit')
        q = self._state()[:, 7:15]
        n = np.count_nonzero(np.abs(2.0 * (q - 0.5 * (_ANT_LO + _ANT_HI)) / (_ANT_HI - _ANT_LO)) > 0.99, axis=1)
        return int(n[0]) if self._e.num_envs == 1 else n

    @property
    def feet_contact(self):
        """the four feet's ground-contact flags as the last step left them (MjAnt.py:55-63; upstream WalkerBaseBulletEnv.step): kept by the kernel for the
        kinds whose observation shows them (AntMaze, AntFlagrun: bits 28..31 of aux[1])"""
        if self._e._cfg.env_kind not in (K.HRL_ANT_MAZE, K.HRL_ANT_FLAGRUN):
            raise AttributeError('feet_contact: kept only for AntMazeBulletEnv / AntFlagrunBulletEnv, whose observations contain the flags')
        bits = self._e._backend().aux[:, 1].cpu().numpy().astype(np.int64) >> 28
        return self._out(((bits[:, None] >> np.arange(4)) & 1).astype(np.float32))

    def calc_potential(self):
        """upstream WalkerBase.calc_potential: -walk_target_dist / dt, dt = the env step of 0.0165 s (MjAnt.py:51, ant_flagrun_env.py:119)"""
        m = self._e._cfg.model
        d = self.walk_target_dist
        return -d / (m.timestep * m.frame_skip)

    def _state(self):
        return self._e._backend().state.double().cpu().numpy()

    def _out(self, a):
        return a[0] if self._e.num_envs == 1 else a

    @property
    def body_real_xyz(self):  # torso position
        return self._out(self._state()[:, 0:3])

    @property
    def body_rpy(self):
        return self._out(quat_to_rpy(self._state()[:, 3:7]))

    @property
    def body_xyz(self):  # (centroid x, centroid y, torso z) for the ants; the cube's position for the point bot
        st = self._state()
        if self._e._cfg.env_kind == K.HRL_POINT_GATHER:
            return self._out(st[:, 0:3])
        c = self._e._cfg
        cen = ant_parts_centroid(st[:, :15], c.centroid_n_static if self._e._centroid_obs else 0,
                                 (c.centroid_static_sum[0], c.centroid_static_sum[1]) if self._e._centroid_obs else (0.0, 0.0))
        return self._out(np.concatenate([cen, st[:, 2:3]], 1))

    @property
    def initial_z(self):
        return self._out(self._state()[:, K.HRL_INITZ_OFF])

    def _target(self):
        return self._e._walk_target()  # [N, 2] numpy

    @property
    def walk_target_x(self):
        t = self._target()[:, 0]
        return float(t[0]) if self._e.num_envs == 1 else t

    @property
    def walk_target_y(self):
        t = self._target()[:, 1]
        return float(t[0]) if self._e.num_envs == 1 else t

    @property
    def walk_target_dist(self):
        b = self.body_xyz
        b = b[None] if self._e.num_envs == 1 else b
        d = np.linalg.norm(self._target() - b[:, 0:2], axis=1)
        return float(d[0]) if self._e.num_envs == 1 else d
