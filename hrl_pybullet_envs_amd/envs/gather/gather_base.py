"""GatherBulletEnv -- mirror of hrl_pybullet_envs/envs/gather/gather_base.py:11-191: the robot-agnostic gather env
(PointGatherBulletEnv derives from it, point_gather_env.py:7-24) on the HIP step.

`robot` selects the kernel family through `robot.env_kind`: the in-tree PointBot (point_bot.py) is the one robot the
reference ever passes; an object with env_kind = HRL_ANT_GATHER gets the ant step with this class's defaults."""
import numpy as np

from ... import _capi as K
from ... import _lib
from ..base import BatchedGymEnv


class GatherBulletEnv(BatchedGymEnv):
    FOOD = 'food'
    POISON = 'poison'
    _gather_info = True
    _centroid_obs = False  # the gather envs drop the target terms (ant_gather_env.py:81): the step never forms the centroid

    def __init__(self,
                 robot,
                 n_food=8,
                 n_poison=8,
                 world_size=(15, 15),
                 n_bins=5,
                 sensor_range=20.,
                 sensor_span=np.pi,
                 robot_coll_dist=1,  # <= 0: pickup by contact with the item cubes (gather_base.py:103-106)
                 robot_object_spacing=2.,
                 dying_cost=-10,
                 render=False,
                 use_sensor=True,
                 respawn=True,
                 debug=False,
                 num_envs=1, device='cuda:0', seed=None):
        kind = getattr(robot, 'env_kind', None)
        if kind not in (K.HRL_POINT_GATHER, K.HRL_ANT_GATHER):
            raise TypeError('GatherBulletEnv needs a robot the batched step knows (PointBot, or env_kind = HRL_ANT_GATHER)')
        self._robot = robot
        cfg = _lib.default_config(kind, n_food=int(n_food), n_poison=int(n_poison),
                                  world_size=tuple(float(w) for w in world_size), n_bins=int(n_bins),
                                  sensor_range=float(sensor_range), sensor_span=float(sensor_span),
                                  robot_coll_dist=float(robot_coll_dist), robot_object_spacing=float(robot_object_spacing),
                                  dying_cost=float(dying_cost), use_sensor=int(bool(use_sensor)), respawn=int(bool(respawn)))
        cfg.centroid_static_sum[0] = -float(world_size[0]) / 2  # last wall loaded (sizeable_enclosed_scene.py:56-58)
        self.n_bins, self.sensor_span, self.sensor_range = n_bins, sensor_span, sensor_range
        self.use_sensor, self.dying_cost, self.robot_coll_dist = use_sensor, dying_cost, robot_coll_dist
        self.n_food, self.n_poison, self.world_size = n_food, n_poison, world_size
        self.spacing, self.respawn, self.debug = robot_object_spacing, respawn, debug
        self.walk_target_x = self.walk_target_y = 0
        self._finish_init(cfg, num_envs, device, seed)

    @property
    def stadium_scene(self):
        """`env.stadium_scene` (gather_base.py:57-60): the items as the reference's scene holds them -- `food`, `poison`, `all_items` --, read from the env's tensors."""
        from .gather_scene import GatherScene
        return GatherScene(self)

    scene = stadium_scene

    @property
    def robot(self):
        """The robot object handed to the constructor (gather_base.py:31); a PointBot reads its live pose attributes
        (body_real_xyz, body_xyz, body_rpy) from this env's state through `_view`."""
        if getattr(self, '_cfg', None) is not None and hasattr(self._robot, '_view'):
            from ..robot_view import RobotView
            self._robot._view = RobotView(self)
        return self._robot

    @robot.setter
    def robot(self, r):
        self._robot = r
