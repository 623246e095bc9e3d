"""tests/tools/fit_model.py -- fitting `hrl_model` to recorded reference steps -- on steps "recorded" from the oracle itself at moved
parameters: the search must find them again, i.e. the model choices nothing in the reference tree decides really are a matter of
configuration.  With real pybullet fixtures (tools/make_pybullet_golden.py) the same tool fits the real thing."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools'))
import fit_model  # noqa: E402
import orc  # noqa: E402
from hrl_pybullet_envs_amd import _capi as K  # noqa: E402

LO = np.radians([-40, 30, -40, -100, -40, -100, -40, 30])
HI = np.radians([40, 100, 40, -30, 40, -30, 40, 100])


def recorded_steps(true_model, n, seed):
    """n one-step records of standing / landing ants under the model `true_model`, in the format of tests/golden/pybullet_*.json"""
    rng = np.random.RandomState(seed)
    cfg = orc.default_config(K.HRL_ANT_GATHER, num_envs=n, seed=1, auto_reset=0, max_episode_steps=0, **{'model_' + k: v for k, v in true_model.items()})
    o = orc.OracleEnv(cfg, np.float64)
    o.reset()
    for t in range(40):   # let them come down and stand: contacts, limits, friction all in play
        o.step(rng.uniform(-1, 1, (n, 8)))
    steps = []
    q0, v0, it0 = o.state[:, :15].copy(), o.state[:, 15:29].copy(), o.items.copy()
    act = rng.uniform(-1, 1, (n, 8))
    o.step(act)
    for i in range(n):
        steps.append({'qpos': q0[i].tolist(), 'qvel': v0[i].tolist(), 'items': it0[i, :32].reshape(16, 2).tolist(), 'action': act[i].tolist(),
                      'qpos_after': o.state[i, :15].tolist(), 'qvel_after': o.state[i, 15:29].tolist()})
    return steps


def test_fit_recovers_moved_model_parameters():
    true = dict(density=850.0, contact_erp=0.6, friction_ground=0.55, joint_damping=1.0, joint_armature=1.0, solver_iters=8)   # (damping / armature: assets/ant.xml:8)
    steps = recorded_steps(true, 160, seed=3)
    rep = fit_model.Replay(K.HRL_ANT_GATHER, steps)
    fitted, before, after = fit_model.fit(rep, list(true), verbose=False)
    assert before[0] > 1e-3 and after[0] < 1e-7, (before, after)          # the default model is off by millimetres per step, the fitted one reproduces the records
    assert fitted['solver_iters'] == 8
    for k in ('density', 'contact_erp', 'friction_ground', 'joint_damping', 'joint_armature'):
        assert abs(fitted[k] - true[k]) <= 0.02 * abs(true[k]), (k, fitted[k], true[k])


def test_fit_of_the_default_model_stays_at_the_defaults():
    steps = recorded_steps({}, 64, seed=5)
    rep = fit_model.Replay(K.HRL_ANT_GATHER, steps)
    fitted, before, after = fit_model.fit(rep, ['density', 'contact_erp', 'restitution'], verbose=False)
    assert before[0] < 1e-12 and after[0] < 1e-12
    assert abs(fitted['density'] - 1000.0) < 1.0 and abs(fitted['contact_erp'] - 0.9) < 1e-3 and fitted['restitution'] < 1e-3
