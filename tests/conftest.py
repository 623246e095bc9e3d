import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


import pytest


@pytest.fixture(autouse=True)
def _upstream_class_attributes_as_shipped():
    """AntFlagrunBulletEnv.reset() assigns 0 to upstream WalkerBaseBulletEnv's cost weights ON THE CLASS (ant_flagrun_env.py:133-135), for the
    whole process -- in the reference and, through envs/upstream.py, here.  Every test starts from upstream's values."""
    from hrl_pybullet_envs_amd.envs.upstream import WalkerBaseBulletEnv as W
    W.electricity_cost, W.stall_torque_cost, W.foot_collision_cost, W.joints_at_limit_cost = -2.0, -0.1, -1.0, -0.1
    yield
