"""`import hrl_pybullet_envs_amd.alias` makes `import hrl_pybullet_envs` -- the reference's package name -- resolve to this package, module path for
module path (`hrl_pybullet_envs.envs.gather.ant_gather_env`, `...envs.ant_maze.ant_maze_bullet_env`, `...utils`, ...): a user script then runs
without touching its import lines.  Refuses when the real reference has already been imported in the process (an A/B run keeps both names)."""
import importlib
import pkgutil
import sys

import hrl_pybullet_envs_amd as _pkg

REFERENCE_NAME = 'hrl_pybullet_envs'


def install(name=REFERENCE_NAME):
    have = sys.modules.get(name)
    if have is not None and have is not _pkg:
        raise ImportError(f'{name!r} is already imported from {getattr(have, "__file__", "?")}: the alias would shadow the reference')
    sys.modules[name] = _pkg
    aliased = [name]
    for m in pkgutil.walk_packages(_pkg.__path__, _pkg.__name__ + '.'):
        tail = m.name[len(_pkg.__name__):]
        if tail.split('.')[1] not in ('envs', 'utils'):
            continue   # the reference's layout is envs/... and utils.py; this package's own machinery (library, bindings, adapters) has no counterpart
        sys.modules[name + tail] = importlib.import_module(m.name)
        aliased.append(name + tail)
    return aliased


install()
