"""AntMazeBulletEnv -- mirror of hrl_pybullet_envs/envs/ant_maze/ant_maze_bullet_env.py:17-178 on the HIP step."""
import numpy as np

from ... import _capi as K
from ... import _lib
from ...utils import PositionEncoding
from ..base import BatchedGymEnv
from ..upstream import walker_costs

_eval_target = [-2, 4]
_targets = ([2, -3], [2, 0], [2, 3], _eval_target)  # ant_maze_bullet_env.py:13-14


class AntMazeBulletEnv(BatchedGymEnv):
    def __init__(self, n_bins: int = 10, sensor_range: float = 5, sensor_span: float = 2 * np.pi, targets=_targets,
                 target_encoding=0, sense_target=False, sense_walls=True, done_at_target=True,
                 max_steps=-1, tol=1.5, inner_rew_weight=0, targ_dist_rew=False, seed=None, debug=0,
                 num_envs=1, device='cuda:0'):
        if isinstance(target_encoding, int):
            target_encoding = PositionEncoding(target_encoding)  # ValueError on anything but 0/1, as in the reference
        cfg = _lib.default_config(K.HRL_ANT_MAZE, n_bins=int(n_bins), sensor_range=float(sensor_range),
                                  sensor_span=float(sensor_span), targets=[tuple(t) for t in targets],
                                  target_encoding=int(target_encoding.value), sense_target=int(bool(sense_target)),
                                  sense_walls=int(bool(sense_walls)), done_at_target=int(bool(done_at_target)),
                                  max_steps=int(max_steps), tol=float(tol), inner_rew_weight=float(inner_rew_weight),
                                  targ_dist_rew=int(bool(targ_dist_rew)))
        cfg.walker_electricity_cost, cfg.walker_stall_torque_cost, cfg.walker_joints_at_limit_cost = walker_costs()   # upstream's class attributes, as they are now
        self.n_bins, self.sensor_range, self.sensor_span = n_bins, float(sensor_range), sensor_span
        self.targets, self.sense_walls, self.sense_target = targets, sense_walls, sense_target
        self.done_at_target, self.max_steps, self.tol = done_at_target, max_steps, tol
        self.inner_rew_weight, self.targ_dist_rew, self.target_encoding, self.debug = inner_rew_weight, targ_dist_rew, target_encoding, debug
        self._finish_init(cfg, num_envs, device, seed)

    def _sync_class_weights(self):
        """super().step() reads WalkerBaseBulletEnv.electricity_cost / stall_torque_cost / joints_at_limit_cost off the class every step (and an
        AntFlagrunBulletEnv.reset() anywhere in the process has set them to 0, ant_flagrun_env.py:133-135)."""
        c, w = self._cfg, walker_costs()
        if (c.walker_electricity_cost, c.walker_stall_torque_cost, c.walker_joints_at_limit_cost) != tuple(np.float32(x) for x in w):
            self._change_config(walker_electricity_cost=w[0], walker_stall_torque_cost=w[1], walker_joints_at_limit_cost=w[2])

    @property
    def stadium_scene(self):
        """`env.stadium_scene` / `env.scene` (ant_maze_bullet_env.py:59-61): the maze's bounding lines and a host-side `sense_walls`."""
        from .maze_scene import MazeScene
        return MazeScene()

    scene = stadium_scene

    # ant_maze_bullet_env.py:38,46,108-110: attributes a trainer reads between steps, over the state tensors
    @property
    def t(self):
        """Episode step counter (:78, :107)."""
        t = self._backend().aux[:, 0]
        return int(t[0]) if self.num_envs == 1 else t

    @property
    def target(self):
        """The episode's target (:46, :110): np.ndarray(2) for one env, [N, 2] tensor for a batch."""
        import torch
        env = self._backend()
        tab = torch.tensor([[self._cfg.targets[i][0], self._cfg.targets[i][1]] for i in range(self._cfg.n_targets)], device=env.device)
        tg = tab[env.aux[:, 3].long().clamp(0, self._cfg.n_targets - 1)]
        return tg[0].double().cpu().numpy() if self.num_envs == 1 else tg

    def _walk_target(self):
        import torch
        tg = self.target
        return tg[None].astype(np.float64) if self.num_envs == 1 else tg.double().cpu().numpy()

    walk_target_x = property(lambda self: self.robot.walk_target_x)
    walk_target_y = property(lambda self: self.robot.walk_target_y)
