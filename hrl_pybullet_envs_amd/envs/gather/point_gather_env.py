"""PointGatherBulletEnv -- mirror of hrl_pybullet_envs/envs/gather/point_gather_env.py:7-24 (+ gather_base.py:11-191, point_bot.py:10-74) on the
HIP step.  The constructor is the reference's, argument for argument (tests/golden/constructor_signatures.json holds its inspect.signature), plus
this package's num_envs / device / seed; everything else is GatherBulletEnv with a PointBot as its robot."""
import numpy as np

from .gather_base import GatherBulletEnv
from .point_bot import PointBot


class PointGatherBulletEnv(GatherBulletEnv):
    def __init__(self, n_food=8, n_poison=8, world_size=(15, 15), n_bins=5, sensor_range=20., sensor_span=np.pi, robot_coll_dist=1,
                 robot_object_spacing=2., dying_cost=-10, render=False, use_sensor=True, respawn=True, debug=False,
                 num_envs=1, device='cuda:0', seed=None):
        forwarded = {k: v for k, v in locals().items() if k != 'self'}   # every argument goes to the base class under its own name
        self.robot = PointBot()                                           # the robot decides the kernel family (robot.env_kind)
        GatherBulletEnv.__init__(self, self.robot, **forwarded)
