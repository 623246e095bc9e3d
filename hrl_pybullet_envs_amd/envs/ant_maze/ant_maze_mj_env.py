"""AntMazeMjEnv -- mirror of hrl_pybullet_envs/envs/ant_maze/ant_maze_mj_env.py:17-78 on the HIP step
(raw 29-d MuJoCo-style state + wall sensor + zero pit/moveable bins + t/1000)."""
import numpy as np

from ... import _capi as K
from ... import _lib
from ..base import BatchedGymEnv
from .ant_maze_bullet_env import AntMazeBulletEnv, PositionEncoding

_eval_target = [-2, 4]
_targets = ([2, -4], [2, 0], [2, 4], [0, 4], _eval_target)  # ant_maze_mj_env.py:13-14


class AntMazeMjEnv(BatchedGymEnv):
    def __init__(self, n_bins: int = 10, sensor_range: float = 5, sensor_span: float = 2 * np.pi, targets=_targets,
                 target_encoding=0, tol=1.5, inner_rew_weight=0, seed=None, debug=0, num_envs=1, device='cuda:0'):
        if isinstance(target_encoding, int):
            target_encoding = PositionEncoding(target_encoding)
        cfg = _lib.default_config(K.HRL_ANT_MAZE_MJ, n_bins=int(n_bins), sensor_range=float(sensor_range),
                                  sensor_span=float(sensor_span), targets=[tuple(t) for t in targets],
                                  target_encoding=int(target_encoding.value), tol=float(tol),
                                  inner_rew_weight=float(inner_rew_weight))
        self.n_bins, self.sensor_range, self.sensor_span = n_bins, float(sensor_range), sensor_span
        self.targets, self.tol, self.inner_rew_weight, self.target_encoding, self.debug = targets, tol, inner_rew_weight, target_encoding, debug
        self._finish_init(cfg, num_envs, device, seed)

    t = AntMazeBulletEnv.t                      # ant_maze_mj_env.py:38,70
    stadium_scene = AntMazeBulletEnv.stadium_scene   # :52-54: the same MazeScene
    scene = AntMazeBulletEnv.scene
    target = AntMazeBulletEnv.target            # :44,:91
    _walk_target = AntMazeBulletEnv._walk_target
    walk_target_x = AntMazeBulletEnv.walk_target_x
    walk_target_y = AntMazeBulletEnv.walk_target_y
