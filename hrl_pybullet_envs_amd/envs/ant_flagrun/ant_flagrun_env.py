"""AntFlagrunBulletEnv -- mirror of hrl_pybullet_envs/envs/ant_flagrun/ant_flagrun_env.py:11-204 on the HIP step
(the reference's pre-training env: chase a sequence of goals, +5000 per goal reached)."""
import numpy as np

from ... import _capi as K
from ... import _lib
from ..base import BatchedGymEnv


class AntFlagrunBulletEnv(BatchedGymEnv):
    ant_env_rew_weight = 1   # ant_flagrun_env.py:157-160
    path_rew_weight = 0
    dist_rew_weight = 0
    goal_reach_rew = 5000

    def __init__(self, size=10, tolerance=0.5, max_targets=100, max_target_dist=0, timeout=200, enclosed=True,
                 use_sensor=False, sensor_bins=8, sensor_span=np.pi, sensor_range=4,
                 switch_flag_on_collision=True, manual_goal_creation=False, seed=123, debug=False,
                 num_envs=1, device='cuda:0'):
        assert (max_target_dist == 0 and max_targets > 0) or (max_targets <= 0 and max_target_dist > 0), \
            'cannot have both max_targets and max_target_dist set at the same time'  # ant_flagrun_env.py:17-18
        cfg = _lib.default_config(K.HRL_ANT_FLAGRUN, flag_size=float(size), tol=float(tolerance), flag_max_targets=int(max_targets),
                                  flag_manual_goals=int(bool(manual_goal_creation)),
                                  flag_max_target_dist=float(max_target_dist),
                                  flag_timeout=int(timeout), flag_enclosed=int(bool(enclosed)), use_sensor=int(bool(use_sensor)),
                                  n_bins=int(sensor_bins), sensor_span=float(sensor_span), sensor_range=float(sensor_range),
                                  flag_switch_on_collision=int(bool(switch_flag_on_collision)),
                                  world_size=(float(size) + 2, float(size) + 2))
        cfg.centroid_static_sum[0] = -(float(size) + 2) / 2
        self.size, self.tol, self.max_targets, self.timeout, self.enclosed = size, tolerance, max_targets, timeout, enclosed
        self.max_target_dist, self.manual_goal_creation = max_target_dist, manual_goal_creation
        self.use_sensor, self.n_bins, self.sensor_span, self.sensor_range, self.debug = use_sensor, sensor_bins, sensor_span, sensor_range, debug
        self._finish_init(cfg, num_envs, device, seed)

    def set_goals(self, goals, mask=None):
        """manual_goal_creation: the reference's `env.goals = [...]; env.next_target()` (ant_flagrun_env.py:91-118).
        goals: [n_goals, 2] (one list for every env) or [num_envs, n_goals, 2], visited in the given order, at most 15.
        Returns the observation towards the first goal, like next_target()."""
        import torch
        if not self.manual_goal_creation:
            raise RuntimeError('set_goals needs manual_goal_creation=True; otherwise reset() draws the goals (ant_flagrun_env.py:149-152)')
        env = self._backend()
        g = torch.as_tensor(np.asarray(goals, dtype=np.float32), device=env.device)
        if g.dim() == 2:
            g = g.unsqueeze(0).expand(self.num_envs, -1, -1)
        obs = env.set_goals(g.contiguous(), mask)
        return obs[0].double().cpu().numpy() if self.num_envs == 1 else obs

    goal = property(lambda self: tuple(self._backend().items[0, 0:2].tolist()) if self.num_envs == 1 else self._backend().items[:, 0:2])
