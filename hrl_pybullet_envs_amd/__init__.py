"""hrl_pybullet_envs_amd -- MI355X-native batched step for the hrl_pybullet_envs environments.

Same class names / constructor kwargs as the reference (hrl_pybullet_envs/__init__.py:3-16); `make(id)` resolves the
reference's registered ids with their 2000-step limit.  With gym installed the ids are also registered with `max_episode_steps=2000`
(gym.make then wraps the single-env object in its own TimeLimit, as it does with the reference's)."""
import types

from .envs.MjAnt import AntMjEnv
from .envs.ant_flagrun.ant_flagrun_env import AntFlagrunBulletEnv
from .envs.ant_maze.ant_maze_bullet_env import AntMazeBulletEnv
from .envs.ant_maze.ant_maze_mj_env import AntMazeMjEnv
from .envs.gather.ant_gather_env import AntGatherBulletEnv
from .envs.gather.point_gather_env import PointGatherBulletEnv

__all__ = ['AntGatherBulletEnv', 'AntMazeMjEnv', 'AntMazeBulletEnv', 'AntFlagrunBulletEnv', 'PointGatherBulletEnv', 'AntMjEnv', 'make']

_REGISTRY = {f'{c.__name__}-v0': c for c in (AntGatherBulletEnv, AntMazeMjEnv, AntMazeBulletEnv, AntFlagrunBulletEnv, PointGatherBulletEnv, AntMjEnv)}


def make(env_id, **kwargs):
    """gym.make() analogue for the ids the reference registers (`<ClassName>-v0`): constructs the class and, like gym.make's TimeLimit
    wrapper (`max_episode_steps=2000`, hrl_pybullet_envs/__init__.py:15), limits its episodes -- inside the kernel."""
    if env_id not in _REGISTRY:
        raise KeyError(f'unknown env id {env_id!r}; known: {sorted(_REGISTRY)}')
    env = _REGISTRY[env_id](**kwargs)
    env.max_episode_steps = env.REGISTERED_STEP_LIMIT
    cls = type(env)
    env.spec = types.SimpleNamespace(id=env_id, entry_point=f'{cls.__module__}:{cls.__name__}', max_episode_steps=env.REGISTERED_STEP_LIMIT,
                                     reward_threshold=None, nondeterministic=False, kwargs=dict(kwargs))   # gym's EnvSpec fields of the registration (:11-16)
    return env


def register_with(gym_module):
    """The reference's registration (hrl_pybullet_envs/__init__.py:11-16): the same ids, `<ClassName>-v0`, each with
    max_episode_steps=2000, so that `gym.make('AntGatherBulletEnv-v0')` of a user script resolves to this package when it
    is imported in place of hrl_pybullet_envs.  AntMjEnv is importable but, as in the reference (:9), not registered.
    Ids the registry already holds (the reference imported in the same process for an A/B run) are left alone."""
    ids = []
    registry = getattr(getattr(gym_module.envs, 'registry', None), 'env_specs', getattr(gym_module.envs, 'registry', None))
    for cls in (AntGatherBulletEnv, AntMazeMjEnv, AntMazeBulletEnv, AntFlagrunBulletEnv, PointGatherBulletEnv):
        env_id = f'{cls.__name__}-v0'
        try:
            if registry is not None and env_id in registry:
                continue
        except TypeError:
            pass
        gym_module.envs.register(id=env_id, entry_point=f'{cls.__module__}:{cls.__name__}', max_episode_steps=2000)
        ids.append(env_id)
    return ids


try:
    import gym as _gym
except Exception:  # gym is optional (absent in the build image; a broken install must not break this package): make() serves the same ids
    _gym = None
if _gym is not None:  # pragma: no cover
    try:
        register_with(_gym)
    except Exception as _e:  # duplicate ids on gym versions that reject them, API drift: make() stays the guaranteed path
        import warnings as _w
        _w.warn(f'hrl_pybullet_envs_amd: gym registration skipped ({type(_e).__name__}: {_e}); use hrl_pybullet_envs_amd.make(id)')
