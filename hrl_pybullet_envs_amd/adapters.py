"""API adapters around the reference-shaped envs.

The reference speaks old-gym (<= 0.21): reset() -> obs, step() -> (obs, rew, done, info) (README.md:24-34), one env per object; a trainer that wants
many of them wraps N objects in a vector env (gym.vector / Stable-Baselines3 VecEnv), which steps them one by one on the host.  Here the batch IS the
env (one kernel launch steps all N, tensors resident in HBM), so the vector-env contracts are thin views over `BatchedEnv`:

* `GymnasiumAdapter`      -- the Gymnasium 5-tuple on one env (numpy) or a batch (torch tensors);
* `GymnasiumVectorEnv`    -- shaped like `gymnasium.vector.VectorEnv` (same-step autoreset): `num_envs`, `single_observation_space`,
                             `single_action_space`, `reset(seed=, options=)`, `step()` -> 5-tuple, `info['final_observation']` /
                             `info['_final_observation']` / `info['final_info']`;
* `SB3VecEnv`             -- shaped like `stable_baselines3.common.vec_env.VecEnv`: numpy in / out, `step_async` launches the kernel on the
                             env's stream and returns at once, `step_wait` synchronises and hands over host arrays, `infos[i]['terminal_observation']`
                             and `infos[i]['TimeLimit.truncated']` for the envs that finished.

Neither gymnasium nor stable-baselines3 is imported (both are absent from the build image): the classes duck-type the contracts, and register
themselves as virtual subclasses when the packages are importable.  The terminal observation comes from the kernel (`hrl_buffers.final_obs`, ABI v6):
an env that ends is reset inside the same launch, so the returned `obs` row is already the next episode's first observation -- exactly the
vector-env convention -- and the observation the episode ended with is what `final_observation` / `terminal_observation` carry."""
import numpy as np


class GymnasiumAdapter:
    def __init__(self, env):
        self.env = env
        self.observation_space, self.action_space = env.observation_space, env.action_space
        self.num_envs = getattr(env, 'num_envs', 1)

    def reset(self, seed=None, options=None):
        if seed is not None:
            self.env.seed(seed)
        return self.env.reset(), {}

    def step(self, action):
        obs, rew, done, info = self.env.step(action)
        if self.num_envs == 1:
            truncated = bool(info.get('TimeLimit.truncated', False))
            return obs, rew, bool(done) and not truncated, truncated, info
        d = done.bool()
        if 'TimeLimit.truncated' in info:   # from the kernel: ended by the step limit alone (gym TimeLimit: `not done`)
            truncated = info['TimeLimit.truncated'].bool()
        else:                               # an old-gym shaped env without the flag: the episode length reached the limit
            limit = getattr(self.env, 'max_episode_steps', 0)
            truncated = d & (info['episode_length'] >= limit) if limit > 0 else d & False
        return obs, rew, d & ~truncated, truncated, info

    def close(self):
        self.env.close()


def _batched_space(space, n):
    """The batched counterpart of a single-env Box: shape (n, *shape)."""
    from .envs.base import _make_box
    lo, hi = float(np.min(space.low)), float(np.max(space.high))
    return _make_box(lo, hi, (n,) + tuple(space.shape))


class GymnasiumVectorEnv:
    """`gymnasium.vector.VectorEnv` contract over a batched env of this package (`AntGatherBulletEnv(num_envs=4096)` ...).

    reset(seed=None, options=None) -> (obs [N, D], {})
    step(actions [N, A])           -> (obs, reward, terminated, truncated, info); an env that ends is reset in the same step (gymnasium's
        AutoresetMode.SAME_STEP): `obs[i]` is then the first observation of its next episode and
            info['final_observation'][i] (alias 'final_obs')   the observation its episode ended with,
            info['_final_observation'][i] (alias '_final_obs') True,
            info['final_info'] = {'episode_return', 'episode_length', 'food_rew', 'dead_rew', '_mask'}   per-env values of the finished episodes.
    Tensors stay on the GPU (`numpy=True`: host numpy arrays instead, one synchronisation per step).

    ALIASING: `obs`, `reward` and the entries of `final_info` are the env's own output tensors, OVERWRITTEN IN PLACE by the next step -- a trainer that
    keeps them across steps (a rollout buffer) must `.clone()` them (with `numpy=True` every array is a fresh host copy).  `final_observation` is
    a new tensor every step: the rows of the envs that finished, zeros elsewhere (gymnasium hands out None / zeros for those rows; the kernel's
    persistent buffer would show the terminal observation of some EARLIER episode there)."""

    metadata = {'autoreset_mode': 'same_step', 'render_modes': ['rgb_array']}
    spec = None
    render_mode = None
    closed = False

    def __init__(self, env, numpy=False):
        if getattr(env, 'num_envs', 1) < 2:
            raise ValueError('GymnasiumVectorEnv wraps a batched env (num_envs >= 2); one env is what GymnasiumAdapter is for')
        self.env, self.num_envs, self._numpy = env, env.num_envs, bool(numpy)
        self.single_observation_space, self.single_action_space = env.observation_space, env.action_space
        self.observation_space = _batched_space(env.observation_space, self.num_envs)
        self.action_space = _batched_space(env.action_space, self.num_envs)

    @property
    def unwrapped(self):
        return self

    def _out(self, t):
        return t.cpu().numpy() if self._numpy else t

    def reset(self, *, seed=None, options=None):
        """options={'reset_mask': bool [N]} (gymnasium >= 1.0) resets the named envs only."""
        if seed is not None:
            self.env.seed(seed if isinstance(seed, int) else int(seed[0]))
        mask = None if not options else options.get('reset_mask')
        if mask is not None:
            import torch
            mask = torch.as_tensor(np.asarray(mask.cpu() if torch.is_tensor(mask) else mask, dtype=bool))
        return self._out(self.env.reset(mask) if mask is not None else self.env.reset()), {}

    def step(self, actions):
        import torch
        if not torch.is_tensor(actions):
            actions = torch.as_tensor(np.asarray(actions, dtype=np.float32))
        obs, rew, done, info = self.env.step(actions)
        d = done.bool()
        truncated = info['TimeLimit.truncated'].bool()
        final = torch.where(d[:, None], info['final_observation'], torch.zeros((), dtype=obs.dtype, device=obs.device))   # fresh, rows that did not finish zeroed
        out = {'final_observation': self._out(final), '_final_observation': self._out(d),
               'final_info': {'episode_return': self._out(info['episode_return']), 'episode_length': self._out(info['episode_length']),
                              'food_rew': self._out(info['food_rew']), 'dead_rew': self._out(info['dead_rew']), '_mask': self._out(d)}}
        out['final_obs'], out['_final_obs'] = out['final_observation'], out['_final_observation']
        if 'retargeted' in info:   # AntFlagrun: info['target'] of the envs whose step switched goals (ant_flagrun_env.py:191,199), gymnasium's masked form
            out['target'], out['_target'] = self._out(info['target'].clone()), self._out(info['retargeted'] != 0)
        return self._out(obs), self._out(rew), self._out(d & ~truncated), self._out(truncated), out

    def render(self):
        return self.env.render('rgb_array')

    def close(self, **kwargs):
        self.closed = True
        self.env.close()


class SB3VecEnv:
    """`stable_baselines3.common.vec_env.VecEnv` contract over a batched env of this package: numpy in / numpy out.

    step_async(actions) launches the step kernel on the env's HIP stream and returns; step_wait() synchronises once and returns
    (obs [N, D], rewards [N], dones [N] bool, infos: list of N dicts).  As in SB3's own VecEnvs an env that is done has been reset already and
    infos[i] holds 'terminal_observation' (the observation its episode ended with) and 'TimeLimit.truncated'; every info carries the gather
    envs' 'food_rew' / 'dead_rew' (ant_gather_env.py:119) when the env has them, and finished envs an SB3-Monitor-style
    'episode': {'r': return, 'l': length}."""

    def __init__(self, env):
        if getattr(env, 'num_envs', 1) < 2:
            raise ValueError('SB3VecEnv wraps a batched env (num_envs >= 2)')
        self.env, self.num_envs = env, env.num_envs
        self.observation_space, self.action_space = env.observation_space, env.action_space   # SB3: the spaces of ONE env
        self.render_mode = 'rgb_array'
        self._pending = False
        self._gather = bool(getattr(env, '_gather_info', False))

    def reset(self):
        return self.env.reset().cpu().numpy()

    def step_async(self, actions):
        import torch
        self._out = self.env.step(torch.as_tensor(np.asarray(actions, dtype=np.float32)))   # asynchronous: returns once the launch is queued
        self._pending = True

    def step_wait(self):
        assert self._pending, 'step_wait() without step_async()'
        self._pending = False
        obs, rew, done, info = self._out
        obs, rew, dones = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy().astype(bool)    # the first copy synchronises with the step
        if self._gather:   # (one list comprehension over .tolist(): per-element numpy indexing costs milliseconds per step at thousands of envs)
            infos = [{'food_rew': f, 'dead_rew': d} for f, d in zip(info['food_rew'].cpu().tolist(), info['dead_rew'].cpu().tolist())]
        else:
            infos = [{} for _ in range(self.num_envs)]
        if 'retargeted' in info:   # AntFlagrun: `i['target'] = self.goal` on the steps in which the goal was switched (ant_flagrun_env.py:191,199)
            sw = info['retargeted'].cpu().numpy() != 0
            if sw.any():
                tg = info['target'].cpu().numpy()
                for i in np.nonzero(sw)[0]:
                    infos[i]['target'] = (float(tg[i, 0]), float(tg[i, 1]))
        if dones.any():
            idx = np.nonzero(dones)[0]
            fin = info['final_observation'][done.bool()].cpu().numpy()
            trunc = info['TimeLimit.truncated'].cpu().numpy().astype(bool)
            ret, length = info['episode_return'].cpu().numpy(), info['episode_length'].cpu().numpy()
            for k, i in enumerate(idx):
                infos[i]['terminal_observation'] = fin[k]
                infos[i]['TimeLimit.truncated'] = bool(trunc[i])
                infos[i]['episode'] = {'r': float(ret[i]), 'l': int(length[i])}
        return obs, rew, dones, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        self.env.close()

    def seed(self, seed=None):
        self.env.seed(seed)
        return [seed] * self.num_envs

    def _indices(self, indices):
        return list(range(self.num_envs)) if indices is None else ([indices] if isinstance(indices, int) else list(indices))

    def get_attr(self, attr_name, indices=None):
        return [getattr(self.env, attr_name) for _ in self._indices(indices)]

    def set_attr(self, attr_name, value, indices=None):
        setattr(self.env, attr_name, value)

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        r = getattr(self.env, method_name)(*method_args, **method_kwargs)
        return [r for _ in self._indices(indices)]

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False for _ in self._indices(indices)]

    def get_images(self):
        return [self.env.render('rgb_array', index=i) for i in range(self.num_envs)]

    def render(self, mode='rgb_array'):
        return self.env.render('rgb_array')

    @property
    def unwrapped(self):
        return self


def _register_virtual_subclasses():
    """isinstance(x, gymnasium.vector.VectorEnv) / isinstance(x, VecEnv) hold where those packages exist; nothing happens where they do not."""
    try:
        from gymnasium.vector import VectorEnv  # pragma: no cover - absent in the build image
        VectorEnv.register(GymnasiumVectorEnv) if hasattr(VectorEnv, 'register') else None
    except Exception:
        pass
    try:
        from stable_baselines3.common.vec_env import VecEnv  # pragma: no cover
        VecEnv.register(SB3VecEnv)
    except Exception:
        pass


_register_virtual_subclasses()
