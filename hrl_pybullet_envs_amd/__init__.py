"""hrl_pybullet_envs_amd -- MI355X-native batched step for the hrl_pybullet_envs environments.

Same class names / constructor kwargs as the reference (hrl_pybullet_envs/__init__.py:3-16); `make(id)` resolves the
reference's registered ids.  With gym installed the ids are also registered with `max_episode_steps=2000`."""
from .envs.MjAnt import AntMjEnv
from .envs.ant_flagrun.ant_flagrun_env import AntFlagrunBulletEnv
from .envs.ant_maze.ant_maze_bullet_env import AntMazeBulletEnv
from .envs.ant_maze.ant_maze_mj_env import AntMazeMjEnv
from .envs.gather.ant_gather_env import AntGatherBulletEnv
from .envs.gather.point_gather_env import PointGatherBulletEnv

__all__ = ['AntGatherBulletEnv', 'AntMazeMjEnv', 'AntMazeBulletEnv', 'AntFlagrunBulletEnv', 'PointGatherBulletEnv', 'AntMjEnv', 'make']

_REGISTRY = {f'{c.__name__}-v0': c for c in (AntGatherBulletEnv, AntMazeMjEnv, AntMazeBulletEnv, AntFlagrunBulletEnv, PointGatherBulletEnv, AntMjEnv)}


def make(env_id, **kwargs):
    """gym.make() analogue for the ids the reference registers (`<ClassName>-v0`)."""
    if env_id not in _REGISTRY:
        raise KeyError(f'unknown env id {env_id!r}; known: {sorted(_REGISTRY)}')
    return _REGISTRY[env_id](**kwargs)


def register_with(gym_module):
    """The reference's registration (hrl_pybullet_envs/__init__.py:11-16): the same ids, `<ClassName>-v0`, each with
    max_episode_steps=2000, so that `gym.make('AntGatherBulletEnv-v0')` of a user script resolves to this package when it
    is imported in place of hrl_pybullet_envs.  AntMjEnv is importable but, as in the reference (:9), not registered."""
    ids = []
    for cls in (AntGatherBulletEnv, AntMazeMjEnv, AntMazeBulletEnv, AntFlagrunBulletEnv, PointGatherBulletEnv):
        gym_module.envs.register(id=f'{cls.__name__}-v0', entry_point=f'{cls.__module__}:{cls.__name__}', max_episode_steps=2000)
        ids.append(f'{cls.__name__}-v0')
    return ids


try:
    import gym as _gym
except ImportError:  # gym is optional (absent in the build image): make() above serves the same ids
    _gym = None
if _gym is not None:  # pragma: no cover
    register_with(_gym)
