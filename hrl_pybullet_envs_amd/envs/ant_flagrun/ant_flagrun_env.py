"""AntFlagrunBulletEnv -- mirror of hrl_pybullet_envs/envs/ant_flagrun/ant_flagrun_env.py:11-204 on the HIP step
(the reference's pre-training env: chase a sequence of goals, +5000 per goal reached)."""
import numpy as np

from ... import _capi as K
from ... import _lib
from ..base import BatchedGymEnv
from ..upstream import WalkerBaseBulletEnv, walker_costs


class AntFlagrunBulletEnv(BatchedGymEnv):
    """Class-level reward weights, as in the reference (ant_flagrun_env.py:157-160; `step` reads them off the CLASS, :169-186):
        r = ant_env_rew_weight * r_upstream + path_rew_weight * path_rew - dist_rew_weight * walk_target_dist (+ goal_reach_rew per goal reached).
    Set them on the class, before constructing an env or while it runs -- `AntFlagrunBulletEnv.path_rew_weight = 0.5` --: every step compares the
    class attributes with what the kernel holds (hrl_config.flag_*_rew_weight, flag_goal_reach_rew) and hands a change over in place
    (hrl_update_config); `env.set_reward_weights(...)` changes them for ONE env."""
    ant_env_rew_weight = 1
    path_rew_weight = 0
    dist_rew_weight = 0
    goal_reach_rew = 5000
    _goal_info = True

    def __init__(self, size=10, tolerance=0.5, max_targets=100, max_target_dist=0, timeout=200, enclosed=True,
                 use_sensor=False, sensor_bins=8, sensor_span=np.pi, sensor_range=4,
                 switch_flag_on_collision=True, manual_goal_creation=False, seed=123, debug=False,
                 num_envs=1, device='cuda:0', goal_capacity=15):
        assert (max_target_dist == 0 and max_targets > 0) or (max_targets <= 0 and max_target_dist > 0), \
            'cannot have both max_targets and max_target_dist set at the same time'  # ant_flagrun_env.py:17-18
        cfg = _lib.default_config(K.HRL_ANT_FLAGRUN, flag_size=float(size), tol=float(tolerance), flag_max_targets=int(max_targets),
                                  flag_manual_goals=int(bool(manual_goal_creation)),
                                  flag_goal_capacity=int(goal_capacity),  # longest `env.goals` list of a manual env (1..61)
                                  flag_max_target_dist=float(max_target_dist),
                                  flag_timeout=int(timeout), flag_enclosed=int(bool(enclosed)), use_sensor=int(bool(use_sensor)),
                                  n_bins=int(sensor_bins), sensor_span=float(sensor_span), sensor_range=float(sensor_range),
                                  flag_switch_on_collision=int(bool(switch_flag_on_collision)),
                                  world_size=(float(size) + 2, float(size) + 2),
                                  # the class attributes as they are NOW (the reference reads AntFlagrunBulletEnv.<name> in step, :169-186)
                                  flag_ant_env_rew_weight=float(type(self).ant_env_rew_weight), flag_path_rew_weight=float(type(self).path_rew_weight),
                                  flag_dist_rew_weight=float(type(self).dist_rew_weight), flag_goal_reach_rew=float(type(self).goal_reach_rew))
        if enclosed or use_sensor:  # ant_flagrun_env.py:59-64: the arena's floor and last wall join upstream's parts dict
            cfg.centroid_n_static = 2
            cfg.centroid_static_sum[0] = -(float(size) + 2) / 2
        else:                       # upstream stadium scene: only its `floor` at the origin
            cfg.centroid_n_static = 1
            cfg.centroid_static_sum[0] = 0.0
        self.size, self.tol, self.max_targets, self.timeout, self.enclosed = size, tolerance, max_targets, timeout, enclosed
        self.max_target_dist, self.manual_goal_creation = max_target_dist, manual_goal_creation
        self.goal_capacity, self._create_calls = int(goal_capacity), 0
        self.switch_flag_on_collision = switch_flag_on_collision
        self.use_sensor, self.n_bins, self.sensor_span, self.sensor_range, self.debug = use_sensor, sensor_bins, sensor_span, sensor_range, debug
        self._class_weights_seen = (float(type(self).ant_env_rew_weight), float(type(self).path_rew_weight), float(type(self).dist_rew_weight),
                                    float(type(self).goal_reach_rew)) + walker_costs()
        cfg.walker_electricity_cost, cfg.walker_stall_torque_cost, cfg.walker_joints_at_limit_cost = walker_costs()   # (reset() zeroes them on the class)
        self._finish_init(cfg, num_envs, device, seed)

    def reset(self, mask=None):
        """ant_flagrun_env.py:132-155.  Its first three lines assign 0 to upstream WalkerBaseBulletEnv's electricity / stall-torque / joints-at-limit
        costs ON THE CLASS (:133-135): this env, and every other walker env of the process from now on, steps without them."""
        WalkerBaseBulletEnv.electricity_cost = 0
        WalkerBaseBulletEnv.stall_torque_cost = 0
        WalkerBaseBulletEnv.joints_at_limit_cost = 0
        self._sync_class_weights()
        return super().reset(mask)

    def _sync_class_weights(self):
        """step() reads the four weights off AntFlagrunBulletEnv (:169-186) and super().step() the three costs off WalkerBaseBulletEnv, every time:
        a change of a class attribute reaches the kernel with the next step (weights set through set_reward_weights() belong to this env alone
        and stay until the CLASS attribute changes again)."""
        cls = type(self)
        now = (float(cls.ant_env_rew_weight), float(cls.path_rew_weight), float(cls.dist_rew_weight), float(cls.goal_reach_rew)) + walker_costs()
        if now != self._class_weights_seen:
            names = ('flag_ant_env_rew_weight', 'flag_path_rew_weight', 'flag_dist_rew_weight', 'flag_goal_reach_rew',
                     'walker_electricity_cost', 'walker_stall_torque_cost', 'walker_joints_at_limit_cost')
            self._change_config(**{name: v for name, v, seen in zip(names, now, self._class_weights_seen) if v != seen})
            self._class_weights_seen = now   # (after the library took them: a refused change is tried again at the next step)

    def set_reward_weights(self, ant_env_rew_weight=None, path_rew_weight=None, dist_rew_weight=None, goal_reach_rew=None):
        """Changes the reward weights of THIS env (None: keep), in place on a running env (hrl_update_config) -- what assigning to the class
        attribute does in the reference, whose step() reads it every time.  A path reward switched on mid-episode is measured from where the robot
        stood when it got its CURRENT goal, as in the reference: every flagrun env keeps set_target()'s `_goal_start_pos` / `_sq_dist_goal`
        (ant_flagrun_env.py:98-103) in its items record whatever the weight is."""
        new = dict(flag_ant_env_rew_weight=ant_env_rew_weight, flag_path_rew_weight=path_rew_weight, flag_dist_rew_weight=dist_rew_weight,
                   flag_goal_reach_rew=goal_reach_rew)
        self._change_config(**{k: float(v) for k, v in new.items() if v is not None})

    reward_weights = property(lambda self: dict(ant_env_rew_weight=self._cfg.flag_ant_env_rew_weight, path_rew_weight=self._cfg.flag_path_rew_weight,
                                                dist_rew_weight=self._cfg.flag_dist_rew_weight, goal_reach_rew=self._cfg.flag_goal_reach_rew))

    def _host_state(self):
        return {'create_calls': self._create_calls}

    def _load_host_state(self, host):
        self._create_calls = int(host.get('create_calls', self._create_calls))

    @property
    def _goal_start_pos(self):  # ant_flagrun_env.py:49,102
        p = self._backend().items[:, K.HRL_FLAG_START_OFF:K.HRL_FLAG_START_OFF + 2]
        return p[0].double().cpu().numpy() if self.num_envs == 1 else p

    @property
    def _sq_dist_goal(self):  # ant_flagrun_env.py:48,101
        d = self._backend().items[:, K.HRL_FLAG_SQDIST_OFF]
        return float(d[0]) if self.num_envs == 1 else d

    @property
    def stadium_scene(self):
        """`env.stadium_scene` / `env.scene` (ant_flagrun_env.py:57-64): the (size + 2)-sided arena when the env is enclosed or senses walls
        (its bounding lines and a host-side `sense_walls`); the open stadium of the upstream env has no bounding lines: None."""
        if not (self.enclosed or self.use_sensor):
            return None
        from ..sizeable_enclosed_scene import SizeableEnclosedScene
        return SizeableEnclosedScene((self.size + 2, self.size + 2))

    scene = stadium_scene

    # ---- the goal list (ant_flagrun_env.py:45,91-120) over the env's `items` / `aux` tensors (include/hrl_envs.h) ----
    def _listed(self):
        return self.manual_goal_creation and not self.max_target_dist > 0

    @property
    def goals(self):
        """`env.goals`: the goals still to come, in the reference's list order (next_target() pops the LAST one).
        One env: a list of (x, y); batched: ([N, n_max, 2] tensor, [N] lengths).  With max_targets < 1 the list is never
        read (goals are drawn near the robot, :113-114) and stays []."""
        import torch
        env = self._backend()
        if self.max_target_dist > 0:
            return [] if self.num_envs == 1 else (torch.zeros(self.num_envs, 0, 2, device=env.device), torch.zeros(self.num_envs, dtype=torch.int32, device=env.device))
        cur = (env.aux[:, 3] & 0xffff)
        if self._listed():
            pend = env.items[:, K.HRL_FLAG_PENDING_OFF:K.HRL_FLAG_PENDING_OFF + 2 * self.goal_capacity].reshape(self.num_envs, self.goal_capacity, 2)
            if self.num_envs == 1:
                return [tuple(g) for g in pend[0, :int(cur[0])].tolist()]
            return pend, cur
        # shared list (:91-96): goal k of the episode is a function of (seed, episode, k); `cur` of them are used up.
        # create_targets() fills the list front to back and pop() empties it from the back: goal number k sits at index
        # max_targets - k
        from ..._philox import flag_goal
        ep = env.aux[:, 2].cpu().numpy()
        ks = np.arange(self.max_targets, 0, -1)                   # list index i holds goal number max_targets - i
        allg = flag_goal(self._cfg.seed, self._cfg.flag_size, ep[:, None], ks[None, :])  # [N, max_targets, 2]
        left = (self.max_targets - cur.cpu().numpy()).clip(0)
        if self.num_envs == 1:
            return [tuple(map(float, g)) for g in allg[0, :int(left[0])]]
        return torch.from_numpy(allg).to(env.device), torch.from_numpy(left.astype(np.int32)).to(env.device)

    @goals.setter
    def goals(self, goals):
        """`env.goals = [...]` of a manual_goal_creation env: stores the list (one for every env: [n, 2]; or [N, n, 2]), at
        most `goal_capacity` goals (constructor argument of this package, default 15, up to 61); nothing else changes until
        next_target() / a goal is reached."""
        import torch
        if not self._listed():
            raise AttributeError('env.goals can only be assigned with manual_goal_creation=True and max_targets > 0 '
                                 '(otherwise reset() fills the list, or next_target() ignores it: ant_flagrun_env.py:113-116,150-153)')
        env = self._backend()
        g = torch.as_tensor(np.asarray(goals, dtype=np.float32).reshape(-1, 2) if len(goals) == 0 else np.asarray(goals, dtype=np.float32), device=env.device)
        if g.dim() == 2:
            g = g.unsqueeze(0).expand(self.num_envs, -1, -1)
        n = int(g.shape[1])
        if g.shape[0] != self.num_envs or g.shape[2] != 2 or n > self.goal_capacity:
            raise ValueError(f'goals must be [n <= goal_capacity = {self.goal_capacity}, 2] or [num_envs, n, 2] (build the env with a larger '
                             f'goal_capacity, up to {K.HRL_MAX_GOALS})')
        env.items[:, K.HRL_FLAG_PENDING_OFF:] = 0
        env.items[:, K.HRL_FLAG_PENDING_OFF:K.HRL_FLAG_PENDING_OFF + 2 * n] = g.reshape(self.num_envs, 2 * n)
        env.aux[:, 3] = (env.aux[:, 3] & ~0xffff) | n

    def next_target(self, mask=None):
        """`env.next_target()` (ant_flagrun_env.py:112-120): the last goal of the list -- the manual one, or the shared list reset() made --
        (or, with max_targets < 1, a goal near the robot) becomes the target; returns calc_state towards it.  One env: IndexError when the
        list is empty, as in the reference; batched: returns (obs, ok) with ok[i] = 0 for such envs (left unchanged)."""
        obs, ok = self._backend().next_target(mask)
        if self.num_envs == 1:
            if not bool(ok[0]):
                raise IndexError('pop from empty list')
            return obs[0].double().cpu().numpy()[:28]
        return obs, ok

    def set_goals(self, goals, mask=None):
        """`env.goals = [...]; env.next_target()` in one call (one kernel launch): goals [n_goals, 2] (one list for every env)
        or [num_envs, n_goals, 2], at most `goal_capacity`.  As in the reference the list is consumed from its BACK: goals[-1] becomes the
        target now, goals[-2] next, ... (`self.goals.pop()`, :116).  Returns the observation towards the new target."""
        import torch
        if not self._listed():
            raise RuntimeError('set_goals needs manual_goal_creation=True and max_targets > 0; otherwise reset() draws the goals '
                               '(ant_flagrun_env.py:150-153) or next_target() ignores the list (:113-114)')
        env = self._backend()
        g = torch.as_tensor(np.asarray(goals, dtype=np.float32), device=env.device)
        if g.dim() == 2:
            g = g.unsqueeze(0).expand(self.num_envs, -1, -1)
        obs = env.set_goals(g.contiguous(), mask)
        return obs[0].double().cpu().numpy() if self.num_envs == 1 else obs

    def create_targets(self, n):
        """ant_flagrun_env.py:91-96: `self.goals = [self.create_target() for _ in range(n)]` from the RandomState all parallel envs share.
        manual_goal_creation (the documented workflow reset(); create_targets(n); next_target()): n <= goal_capacity goals of the same
        counter-based stream the kernel draws its shared list from, keyed by (seed, how many times create_targets was called, k) -- every env
        and every process with the same seed gets the same list, and a later call a fresh one -- stored like `env.goals = [...]`.
        Otherwise the shared list is a function of (seed, episode, k) that reset() arms with max_targets goals: asking for exactly that is a
        no-op, any other n cannot be served."""
        n = int(n)
        if self._listed():
            if not 0 <= n <= self.goal_capacity:
                raise ValueError(f'create_targets({n}): a manual_goal_creation env holds at most goal_capacity = {self.goal_capacity} goals '
                                 f'(constructor argument of this package, up to {K.HRL_MAX_GOALS})')
            from ..._philox import flag_goal
            self._create_calls += 1
            ks = np.arange(n, 0, -1)  # list index i holds goal number n - i: pop() hands them out in draw order
            self.goals = flag_goal(self._cfg.seed, self._cfg.flag_size, (1 << 30) | self._create_calls, ks).astype(np.float32).reshape(n, 2)
            return
        if self.max_target_dist > 0:
            return  # max_targets < 1: next_target() never reads the list (:113-114)
        if n != self.max_targets:
            raise ValueError(f'create_targets({n}): the shared goal list has max_targets = {self.max_targets} goals per episode, armed by reset(); '
                             'build the env with another max_targets, or with manual_goal_creation=True to make lists of any length')

    @property
    def steps_since_goal_change(self):  # :43,171,192,200
        s = (self._backend().aux[:, 3] >> 16) & 0x7fff
        return int(s[0]) if self.num_envs == 1 else s

    @property
    def _rewarded(self):  # :51
        r = (self._backend().aux[:, 3] >> 31) & 1
        return bool(r[0]) if self.num_envs == 1 else r.bool()

    def _walk_target(self):
        env = self._backend()
        if self.max_target_dist > 0 or self.manual_goal_creation:
            return env.items[:, 0:2].double().cpu().numpy()
        from ..._philox import flag_goal
        aux = env.aux.cpu().numpy()
        return flag_goal(self._cfg.seed, self._cfg.flag_size, aux[:, 2], aux[:, 3] & 0xffff).astype(np.float64)

    @property
    def goal(self):  # :57
        t = self._walk_target()
        return (float(t[0, 0]), float(t[0, 1])) if self.num_envs == 1 else t

    walk_target_x = property(lambda self: self.robot.walk_target_x)
    walk_target_y = property(lambda self: self.robot.walk_target_y)
