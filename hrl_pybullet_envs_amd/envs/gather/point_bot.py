"""PointBot -- mirror of hrl_pybullet_envs/envs/gather/point_bot.py:10-74.

In the reference this class loads player_cube.xml into pybullet, applies the planar force (:28-31) and computes the
8-d state (:48-67).  Here those live in the HIP step (csrc/step_core.h: point_substep, phase_point_state); the class
carries the robot's constants and tells GatherBulletEnv which kernel family simulates it."""
import numpy as np

from ... import _capi as K
from ..base import _make_box


class PointBot:
    start_pos = [0, 0, 0.5]                # point_bot.py:12
    env_kind = K.HRL_POINT_GATHER

    def __init__(self):
        act_dim, obs_dim = 2, 8            # point_bot.py:15-16
        self.action_space = _make_box(-1.0, 1.0, (act_dim,))
        self.observation_space = _make_box(-np.inf, np.inf, (obs_dim,))
        self.initial_z = 1                 # point_bot.py:18
        self.walk_target_x = 0
        self.walk_target_y = 0
        self._view = None                  # set by the env that simulates this robot: live pose of the batched state

    # point_bot.py:50-53: pose attributes calc_state leaves on the robot
    body_real_xyz = property(lambda self: self._view.body_real_xyz if self._view else np.array(self.start_pos, dtype=float))
    body_xyz = property(lambda self: self._view.body_xyz if self._view else np.array(self.start_pos, dtype=float))
    body_rpy = property(lambda self: self._view.body_rpy if self._view else np.zeros(3))
    robot_body = property(lambda self: self._view.robot_body if self._view else None)   # upstream BodyPart of the cube: pose().xyz() / .rpy(), get_position(), speed()

    def alive_bonus(self, z, pitch):
        return 1                           # point_bot.py:73-74: cannot die
