"""The specification's fingerprint: sha256 of what 16 envs of every kind hold after 1, 10 and 80 steps of fixed pseudo-random actions under the DEFAULT
config (plus one config with every ABI-v7 model parameter switched on).  Every operation of the fp32 step is pinned (explicit fma, no contraction,
own transcendentals), so the bytes are the same on the CPU oracle, the host executor and the device, on any machine: a change of the fingerprint IS a
change of the specification -- deliberate (regenerate: `python tests/spec_fingerprint.py write`, and say so in the commit) or a bug.
Test infrastructure only (imports the oracle)."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, 'golden', 'spec_fingerprint.json')
STEPS, MARKS, N = 80, (1, 10, 80), 16
V7 = dict(model_linear_damping=0.04, model_angular_damping=0.04, model_restitution=0.3, model_max_contacts=8, model_joint_damping=1.0, model_joint_armature=1.0)


def cases():
    from hrl_pybullet_envs_amd import _capi as K
    kinds = [('flat', K.HRL_ANT_FLAT), ('gather', K.HRL_ANT_GATHER), ('maze', K.HRL_ANT_MAZE), ('point', K.HRL_POINT_GATHER), ('maze_mj', K.HRL_ANT_MAZE_MJ),
             ('flagrun', K.HRL_ANT_FLAGRUN)]
    out = [(name, kind, {}) for name, kind in kinds]
    out += [('gather+v7', K.HRL_ANT_GATHER, V7), ('point+v7', K.HRL_POINT_GATHER, {k: v for k, v in V7.items() if 'joint' not in k})]
    return out


def digest(env):
    h = hashlib.sha256()
    for name in ('state', 'items', 'aux', 'obs', 'rew', 'done', 'info', 'truncated'):
        h.update(np.ascontiguousarray(getattr(env, name)).tobytes())
    return h.hexdigest()


def run(make_env, name, kind, kw):
    """make_env(cfg) -> an object with reset(), step(actions) and the buffers of tests/orc.py::OracleEnv as numpy arrays"""
    import orc
    cfg = orc.default_config(kind, num_envs=N, seed=11, auto_reset=1, max_episode_steps=60, **kw)
    env = make_env(cfg)
    env.reset()
    rng = np.random.RandomState(len(name) * 7 + kind)      # (legacy generator: its stream is frozen across numpy versions)
    out = {'reset': digest(env)}
    for t in range(1, STEPS + 1):
        env.step(rng.uniform(-1, 1, (N, env.ad)).astype(np.float32))
        if t in MARKS:
            out[str(t)] = digest(env)
    return out


def fingerprint(make_env):
    return {name: run(make_env, name, kind, kw) for name, kind, kw in cases()}


if __name__ == '__main__':
    sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
    import orc
    fp = fingerprint(lambda cfg: orc.OracleEnv(cfg, np.float32))
    if sys.argv[1:] == ['write']:
        json.dump(fp, open(PATH, 'w'), indent=1)
        print('written', PATH)
    else:
        old = json.load(open(PATH))
        print('same' if old == fp else 'DIFFERENT: ' + ', '.join(k for k in fp if fp[k] != old.get(k)))
