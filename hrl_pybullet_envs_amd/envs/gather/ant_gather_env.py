"""AntGatherBulletEnv -- mirror of hrl_pybullet_envs/envs/gather/ant_gather_env.py:12-200 on the HIP step."""
import numpy as np

from ... import _capi as K
from ... import _lib
from ..base import BatchedGymEnv


class AntGatherBulletEnv(BatchedGymEnv):
    FOOD = 'food'
    POISON = 'poison'
    _gather_info = True
    _centroid_obs = False  # the gather envs drop the target terms (ant_gather_env.py:81): the step never forms the centroid

    def __init__(self,
                 n_food=8,
                 n_poison=8,
                 world_size=(15, 15),
                 n_bins=10,
                 sensor_range=20.,
                 sensor_span=np.pi,
                 robot_coll_dist=1,
                 robot_object_spacing=2.,
                 dying_cost=-10,
                 use_sensor=True,
                 respawn=True,
                 render=False,  # accepted and ignored, as in the reference (ant_gather_env.py:28,30)
                 debug=False,
                 num_envs=1, device='cuda:0', seed=None):
        cfg = _lib.default_config(K.HRL_ANT_GATHER, n_food=int(n_food), n_poison=int(n_poison),
                                  world_size=tuple(float(w) for w in world_size), n_bins=int(n_bins),
                                  sensor_range=float(sensor_range), sensor_span=float(sensor_span),
                                  robot_coll_dist=float(robot_coll_dist), robot_object_spacing=float(robot_object_spacing),
                                  dying_cost=float(dying_cost), use_sensor=int(bool(use_sensor)), respawn=int(bool(respawn)))
        cfg.centroid_static_sum[0] = -float(world_size[0]) / 2  # last wall loaded (sizeable_enclosed_scene.py:56-58)
        self.n_bins, self.sensor_span, self.sensor_range = n_bins, sensor_span, sensor_range
        self.use_sensor, self.dying_cost, self.robot_coll_dist = use_sensor, dying_cost, robot_coll_dist
        self.n_food, self.n_poison, self.world_size = n_food, n_poison, world_size
        self.spacing, self.respawn, self.debug = robot_object_spacing, respawn, debug
        self._finish_init(cfg, num_envs, device, seed)

    @property
    def stadium_scene(self):
        """`env.stadium_scene` (ant_gather_env.py:57-60): the items as the reference's scene holds them -- `food`, `poison`, `all_items` --, read from the env's tensors."""
        from .gather_scene import GatherScene
        return GatherScene(self)

    scene = stadium_scene
