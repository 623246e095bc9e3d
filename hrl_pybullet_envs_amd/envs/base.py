"""Host-side mirror of the reference's gym.Env interface on top of the batched HIP step.

The reference's boundary is the old-gym (<= 0.21) class API (hrl_pybullet_envs/__init__.py:11-16, README.md:24-34):
    reset() -> obs ;  step(a) -> (obs, rew, done, info) ;  seed(s) ;  observation_space / action_space
`num_envs == 1` (default) behaves like the reference object: numpy in, numpy/float/bool out, `info` a dict of
python numbers.  `num_envs > 1` is the batched form: torch tensors resident on the GPU, `done` envs auto-reset (the
returned obs is then the first observation of the next episode, as vector-env wrappers do).

The step limit.  In the reference the classes have none: `gym.make(id)` wraps them in `TimeLimit(2000)` (__init__.py:15).  Here the limit
lives in the kernel (`hrl_config.max_episode_steps`, `info['TimeLimit.truncated']`), switched on by whoever plays gym.make's part:
  * `hrl_pybullet_envs_amd.make(id, ...)`            -> 2000, as gym.make;
  * a class constructed directly with num_envs == 1  -> none, like the reference's object (so that a real `gym.make`, which wraps the object in
                                                        its own TimeLimit, does not meet a second limit that would turn its truncation flag off);
  * a class constructed directly with num_envs > 1   -> 2000: a batch is its own vector env, no wrapper of gym's can sit around it;
  * `env.max_episode_steps = n` at any time (0: none).
"""
import ctypes as C

import numpy as np

from .. import _capi as K
from .. import _lib


class Box:
    """Minimal stand-in for gym.spaces.Box (gym is optional: used when it is not installed)."""

    def __init__(self, low, high, shape, dtype=np.float32):
        self.low = np.full(shape, low, dtype=dtype)
        self.high = np.full(shape, high, dtype=dtype)
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return np.random.uniform(lo, hi).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return f'Box{self.shape}'


def _make_box(low, high, shape):
    try:
        from gym.spaces import Box as GymBox  # pragma: no cover - gym is absent in the build image
        return GymBox(low, high, shape=shape, dtype=np.float32)
    except Exception:
        return Box(low, high, shape)


class BatchedGymEnv:
    """Common machinery; subclasses fill `self._cfg` (hrl_config) from their reference constructor kwargs."""

    metadata = {'render.modes': ['rgb_array']}
    reward_range = (-float('inf'), float('inf'))
    REGISTERED_STEP_LIMIT = 2000  # hrl_pybullet_envs/__init__.py:15

    @property
    def max_episode_steps(self):
        return int(self._cfg.max_episode_steps)

    @max_episode_steps.setter
    def max_episode_steps(self, n):
        """The step limit of gym's TimeLimit, applied inside the kernel (0: none).  On a running env it takes effect with the next step; the
        simulation, the tensors handed out and the host buffers of step_host() stay as they are."""
        n = int(n)
        if n < 0:
            raise ValueError('max_episode_steps must be >= 0 (0: no limit)')
        if n == self._cfg.max_episode_steps:
            return
        self._change_config(max_episode_steps=n)   # a constant of the library handle: replaced in place (hrl_update_config), buffers and pinned host memory stay

    def _change_config(self, **fields):
        """Fields of hrl_config changed on a COPY, handed to the live handle (hrl_update_config) and committed to `self._cfg` only when the library
        took them: a refused change (it raises, with the library's reason) leaves host and device configs equal."""
        c = type(self._cfg).from_buffer_copy(self._cfg)
        for k, v in fields.items():
            setattr(c, k, v)
        if self._env is not None:
            self._env.update_config(c)
        self._cfg = c

    # what gym.make()'s TimeLimit wrapper and registration leave on the object a user of the reference holds (hrl_pybullet_envs/__init__.py:11-16):
    # `env._max_episode_steps`, `env._elapsed_steps` (gym.wrappers.TimeLimit), `env.spec.id` / `.max_episode_steps` (set by make())
    spec = None

    @property
    def _max_episode_steps(self):
        return self.max_episode_steps or None   # TimeLimit holds None when there is no limit

    @_max_episode_steps.setter
    def _max_episode_steps(self, n):
        self.max_episode_steps = 0 if n is None else n

    @property
    def _elapsed_steps(self):
        """steps of the running episode (TimeLimit's counter; the kernel's aux[0]): an int for one env, [N] for a batch"""
        t = self._backend().aux[:, 0].cpu().numpy()
        return int(t[0]) if self.num_envs == 1 else t

    def _finish_init(self, cfg, num_envs, device, seed):
        if num_envs < 1:
            raise ValueError('num_envs must be >= 1')
        cfg.num_envs = int(num_envs)
        cfg.max_episode_steps = self.REGISTERED_STEP_LIMIT if num_envs > 1 else 0   # see the module docstring
        cfg.auto_reset = 1 if num_envs > 1 else 0
        cfg.seed = 0 if seed is None else int(seed) & 0xFFFFFFFFFFFFFFFF
        self._cfg, self._device, self.num_envs = cfg, device, int(num_envs)
        L = _lib.lib()
        od, ad = L.hrl_obs_dim(C.byref(cfg)), L.hrl_act_dim(C.byref(cfg))
        self.observation_space = _make_box(-np.inf, np.inf, (od,))
        self.action_space = _make_box(-1.0, 1.0, (ad,))
        self._env = None
        self._last_a = None   # the action of the last step(): `rewards` recomputes the electricity term from it

    # lazily created so that constructing an env (e.g. to read its spaces) needs no GPU
    def _backend(self):
        if self._env is None:
            from ..vec_env import BatchedEnv
            self._env = BatchedEnv(self._cfg, self._device)
        return self._env

    def seed(self, seed=None):
        """ant_gather_env.py:63-66 + gather_scene.py:35-36 / ant_maze_bullet_env.py:99-102: reseeds the env's RNG, the simulation carries on.
        On a live env the seed is a constant of the library handle replaced in place (hrl_update_config): state, items, counters, the tensors
        handed out and the pinned host buffers stay as they are; the streams are counter-based functions of (seed, env id, counters), so the
        next respawn / reset / maze target / goal is drawn from the new seed's stream and nothing else moves."""
        self._change_config(seed=0 if seed is None else int(seed) & 0xFFFFFFFFFFFFFFFF)
        return [seed]

    def reset(self, mask=None):
        """`env.reset()` of the reference.  A batch may reset some of its envs only: `mask` [N] (bool / uint8 tensor), non-zero = reset;
        the others keep their state and their row of the returned observations."""
        obs = self._backend().reset(mask)
        if self.num_envs == 1:
            return obs[0].double().cpu().numpy()
        return obs

    def state_dict(self):
        """Checkpoint of the simulation (vec_env.BatchedEnv.state_dict): torch.save()-able, resumes bit for bit.  `host` holds what the
        Python object itself carries from call to call (AntFlagrun: how many goal lists create_targets() has drawn)."""
        sd = self._backend().state_dict()
        sd['host'] = self._host_state()
        return sd

    def load_state_dict(self, sd, strict=True):
        obs = self._backend().load_state_dict(sd, strict)
        self._load_host_state(sd.get('host', {}))
        return obs[0].double().cpu().numpy() if self.num_envs == 1 else obs

    def _host_state(self):
        return {}

    def _load_host_state(self, host):
        pass

    def _sync_class_weights(self):
        """The reference reads some reward weights off CLASS attributes in every step (upstream WalkerBaseBulletEnv's costs, AntFlagrunBulletEnv's
        weights): subclasses compare them with what the kernel holds and hand over a change (hrl_update_config).  Nothing by default."""

    def step(self, a):
        self._sync_class_weights()
        env = self._backend()
        self._last_a = a        # (what `rewards` recomputes the electricity term from)
        if self.num_envs == 1:
            # numpy in / numpy out like the reference: one launch + one synchronisation, the kernel reads the action from and
            # writes its outputs to pinned host memory (BatchedEnv.step_host)
            obs, rew, done, info = env.step_host(np.asarray(a, dtype=np.float32).reshape(1, -1))
            d = bool(done[0] != 0)
            out = {'food_rew': float(info[0, 0]), 'dead_rew': float(info[0, 1])} if self._gather_info else {}
            if d and int(info[0, 3]) >= self.max_episode_steps > 0:  # info[3] = length of the episode that just ended: the step limit was hit;
                out['TimeLimit.truncated'] = bool(env.host_final_obs()[1][0])  # gym.wrappers.TimeLimit: `not done` (the kernel's flag)
            if self._goal_info:  # ant_flagrun_env.py:191,199: `i['target'] = self.goal` on the steps in which next_target() ran
                g = env.host_goal()[0]
                if g[2] != 0:
                    out['target'] = (float(g[0]), float(g[1]))
            return obs[0].astype(np.float64), float(rew[0]), d, out
        return env.step(a)

    def close(self):
        if self._env is not None:
            self._env.close()
            self._env = None

    def render(self, mode='human', index=0, size=256):
        """`rgb_array`: a top-down picture of env `index` (arena, walls, maze box, food green / poison red, the robot),
        drawn on the host from the state record -- a debugging aid.  The reference renders through
        pybullet's GUI / camera (upstream MJCFBaseBulletEnv.render); `human` mode has no window here and returns None."""
        if mode != 'rgb_array':
            return None
        from .render import draw_env
        env = self._backend()
        st = env.state[index].cpu().numpy()
        items = env.items[index].cpu().numpy() if self._gather_info else None
        return draw_env(self._cfg, st, items, size)

    _gather_info = False
    _goal_info = False    # AntFlagrun: info['target']
    _centroid_obs = True  # kinds whose walk target distance is measured from upstream's parts centroid

    def _walk_target(self):
        """[N, 2] numpy: what `robot.walk_target_x/y` hold.  Default: the constructor's walk target (upstream's (1e3, 0))."""
        return np.tile(np.array([[self._cfg.walk_target[0], self._cfg.walk_target[1]]], np.float64), (self.num_envs, 1))

    @property
    def robot(self):
        from .robot_view import RobotView
        return RobotView(self)

    @property
    def robot_body(self):
        """`self.robot_body.pose().xyz()` / `.rpy()` as the reference's envs read the torso (ant_maze_bullet_env.py:67,125,130)"""
        return self.robot.robot_body

    @property
    def potential(self):
        """upstream WalkerBaseBulletEnv.potential: the potential the last step (or reset) left, what the next step's `progress` is measured from (MjAnt.py:50-52)"""
        p = self._backend().state[:, K.HRL_POTENTIAL_OFF].double().cpu().numpy()
        return float(p[0]) if self.num_envs == 1 else p

    @property
    def rewards(self):
        """The terms of the last step's locomotion reward, upstream's `self.rewards` (MjAnt.py:82-87: [alive, progress, joints_at_limit_cost,
        feet_collision_cost]; upstream WalkerBaseBulletEnv.step for AntMaze / AntFlagrun: [alive, progress, electricity_cost, joints_at_limit_cost,
        feet_collision_cost]).  alive and progress are the kernel's own values (info[:, 0:2] of these kinds); the cost terms are recomputed on the host
        from the state the step left and the action it was given.  One env: a list of floats; a batch: a list of [N] arrays (rows whose episode has
        just ended and was reset in place show the cost terms of the new episode's first state)."""
        kind = self._cfg.env_kind
        if kind in (K.HRL_ANT_GATHER, K.HRL_POINT_GATHER):
            raise AttributeError('rewards: the gather envs have no locomotion reward (ant_gather_env.py:118-119 returns food_reward + dead_rew)')
        info = self._backend().info[:, 0:2].double().cpu().numpy()
        r = self.robot
        jal = np.atleast_1d(r.joints_at_limit).astype(np.float64)
        mj = kind in (K.HRL_ANT_FLAT, K.HRL_ANT_MAZE_MJ)
        terms = [info[:, 0], info[:, 1]]
        if not mj:
            a = self._last_a
            a = np.zeros((self.num_envs, 8)) if a is None else (a.detach().double().cpu().numpy() if hasattr(a, 'detach') else np.asarray(a, np.float64)).reshape(self.num_envs, 8)
            js = np.atleast_2d(r.joint_speeds)
            c = self._cfg
            terms.append(c.walker_electricity_cost * np.abs(a * js).mean(axis=1) + c.walker_stall_torque_cost * np.square(a).mean(axis=1))
        terms += [(-0.1 if mj else self._cfg.walker_joints_at_limit_cost) * jal, np.zeros(self.num_envs)]
        return [float(t[0]) for t in terms] if self.num_envs == 1 else terms

    @property
    def unwrapped(self):
        return self
