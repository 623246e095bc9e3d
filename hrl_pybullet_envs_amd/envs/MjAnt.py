"""AntMjEnv -- mirror of hrl_pybullet_envs/envs/MjAnt.py:31-97 (flat ground, raw 29-d state) on the HIP step."""
from .. import _capi as K
from .. import _lib
from .base import BatchedGymEnv


class AntMjEnv(BatchedGymEnv):
    def __init__(self, num_envs=1, device='cuda:0', seed=None):
        self._finish_init(_lib.default_config(K.HRL_ANT_FLAT), num_envs, device, seed)
