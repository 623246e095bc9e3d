"""PointGatherBulletEnv -- mirror of hrl_pybullet_envs/envs/gather/point_gather_env.py:7-24
(+ gather_base.py:11-191, point_bot.py:10-74) on the HIP step."""
import numpy as np

from .gather_base import GatherBulletEnv
from .point_bot import PointBot


class PointGatherBulletEnv(GatherBulletEnv):
    def __init__(self,
                 n_food=8,
                 n_poison=8,
                 world_size=(15, 15),
                 n_bins=5,
                 sensor_range=20.,
                 sensor_span=np.pi,
                 robot_coll_dist=1,
                 robot_object_spacing=2.,
                 dying_cost=-10,
                 render=False,
                 use_sensor=True,
                 respawn=True,
                 debug=False,
                 num_envs=1, device='cuda:0', seed=None):
        self.robot = PointBot()
        super().__init__(self.robot, n_food, n_poison, world_size, n_bins, sensor_range, sensor_span, robot_coll_dist,
                         robot_object_spacing, dying_cost, render, use_sensor, respawn, debug,
                         num_envs=num_envs, device=device, seed=seed)
