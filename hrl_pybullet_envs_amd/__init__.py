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


try:  # pragma: no cover - gym is not installed in the build image
    import gym
    for _name, _cls in _REGISTRY.items():
        gym.envs.register(id=_name.replace('-v0', 'AMD-v0'), entry_point=f'{_cls.__module__}:{_cls.__name__}',
                          max_episode_steps=2000)
except Exception:
    pass
