"""AntMjEnv / MjAnt -- mirror of hrl_pybullet_envs/envs/MjAnt.py:10-97 (flat ground, raw 29-d state) on the HIP step."""
import numpy as np

from .. import _capi as K
from .. import _lib
from .base import BatchedGymEnv, _make_box
from .robot_view import RobotView


class MjAnt(RobotView):
    """The robot of AntMjEnv (MjAnt.py:10-28).  In the reference it loads ant.xml and reads joints through pybullet; here the model lives in the
    kernel (csrc/host_cfg.h: build_devcfg) and this class is the reference's surface over the live batched state: the constants of :13-15,
    calc_state() = qpos | qvel (:17-25) and alive_bonus (:27-28)."""
    power = 2.5                                  # MjAnt.py:14
    env_kind = K.HRL_ANT_FLAT

    def __init__(self):
        super().__init__(None)                   # attached by the env that simulates it
        self.action_space = _make_box(-1.0, 1.0, (8,))               # :15 action_dim=8
        self.observation_space = _make_box(-np.inf, np.inf, (29,))   # :15 obs_dim=29

    def calc_state(self):
        if self._e is None:
            raise RuntimeError('MjAnt is not attached to an env (AntMjEnv().robot is)')
        return self._out(self._state()[:, :29])  # qpos (15) | qvel (14): the layout of include/hrl_envs.h is the reference's

    def alive_bonus(self, z, pitch):
        initial_z = 0.75 if self._e is None else self.initial_z
        return np.where(np.asarray(z) - initial_z > 0.26, 1, -1)[()]   # :27-28; a batch gives a batch


class AntMjEnv(BatchedGymEnv):
    def __init__(self, num_envs=1, device='cuda:0', seed=None):
        self._finish_init(_lib.default_config(K.HRL_ANT_FLAT), num_envs, device, seed)
        self._robot = MjAnt()
        self._robot._e = self

    @property
    def robot(self):
        return self._robot
