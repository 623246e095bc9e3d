"""DRY RUN of the road that pins the rigid-body half of the oracle: generator -> replay -> fit, end to end, against stand-in packages.

`parity` can only turn green through tools/make_pybullet_golden.py (records the reference's own step() where pybullet + gym + the reference are
installed) -> tests/test_pybullet_golden.py (replays the records on the oracle) -> tests/tools/fit_model.py (fits the engine parameters nothing in
the reference tree decides).  None of the three can meet pybullet here (ModuleNotFoundError; nothing was denied), so whoever has it gets one shot.
This test makes sure that shot does not execute the recording code for the first time: it runs the GENERATOR ITSELF, unmodified, in a subprocess in
which `gym`, `pybullet`, `pybullet_envs` and `hrl_pybullet_envs` are tests/pybullet_standin.py's duck-typed stand-ins -- pybullet getters and the
reference classes' attributes answered by the fp64 CPU oracle under a PERTURBED hrl_model --, writing to a scratch directory; then
  (a) the replay of those records under the perturbed model deviates by ~0 in every quantity of every env (state order, quaternion convention, joint
      order, item order, target, potential, initial_z, feet flags, flagrun goal bookkeeping all line up; all six kinds -- the pybulletgym flavour, which the
      generator records where it imports and skips where it does not, included), and under the DEFAULT model the report is red;
  (b) the fit recovers the perturbation from the generator's own JSON;
  (c) tests/golden is untouched, stand-in records are marked and refused as fixtures.
It says nothing about pybullet (the stand-in restates its getters' tuple layouts from memory) and pins nothing: the rigid-body step stays
"parity unpinned" until real records are committed.  The reference call sites the records stand for: envs/gather/ant_gather_env.py:77-80,
envs/MjAnt.py:17-25 (state order), envs/ant_maze/ant_maze_bullet_env.py:117."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import pybullet_replay

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools'))
import fit_model  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
PERTURBED = {'density': 850.0, 'contact_erp': 0.6, 'friction_ground': 0.55, 'solver_iters': 8}
ANTS = ('AntGatherBulletEnv', 'AntMazeBulletEnv', 'AntFlagrunBulletEnv', 'AntMazeMjEnv', 'AntMjEnv')


def golden_digest():
    h = hashlib.sha256()
    for name in sorted(os.listdir(GOLDEN)):
        p = os.path.join(GOLDEN, name)
        if os.path.isfile(p):
            h.update(name.encode()); h.update(open(p, 'rb').read())
    return h.hexdigest()


def run_generator(out, *extra, model=PERTURBED):
    env = dict(os.environ, HRL_STANDIN_MODEL=json.dumps(model), PYTHONDONTWRITEBYTECODE='1', OMP_NUM_THREADS='1')
    return subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'pybullet_standin.py'), os.path.join(ROOT, 'tools', 'make_pybullet_golden.py'),
                           '--out', str(out), *extra], capture_output=True, text=True, env=env, cwd=str(out) if os.path.isdir(str(out)) else None)


@pytest.fixture(scope='module')
def records(tmp_path_factory):
    before = golden_digest()
    out = tmp_path_factory.mktemp('pin_road')
    p = run_generator(out, '--steps', '60', '--seeds', '3')
    assert p.returncode == 0, p.stderr[-2000:]
    files = sorted(str(out / f) for f in os.listdir(out))
    assert [os.path.basename(f) for f in files] == ['pybullet_AntFlagrunBulletEnv.json', 'pybullet_AntGatherBulletEnv.json', 'pybullet_AntMazeBulletEnv.json',
                                                    'pybullet_AntMazeMjEnv.json', 'pybullet_AntMjEnv.json', 'pybullet_PointGatherBulletEnv.json']   # (the last two of the ant files: the optional pybulletgym flavour)
    assert golden_digest() == before, 'the dry run touched tests/golden'
    return files


def test_generator_records_what_the_replay_needs(records):
    """every record carries the packed state in the order of include/hrl_envs.h and the task bookkeeping a step reads besides it"""
    for path in records:
        g = json.load(open(path))
        assert g['versions']['standin'] is True and 'standin' in g['versions']['pybullet_api']
        ant = 'Point' not in g['env_id']
        m = g['model']
        assert len(m['links']) == (13 if ant else 1) and len(m['joints']) == (12 if ant else 0)
        if ant:
            assert [j['name'] for j in m['joints'] if j['type'] == 0] == ['hip_1', 'ankle_1', 'hip_2', 'ankle_2', 'hip_3', 'ankle_3', 'hip_4', 'ankle_4']
            assert abs(m['total_mass'] - pybullet_replay.ant_total(850.0)) < 1e-9
        assert m['engine']['numSubSteps'] == 4 and m['engine']['numSolverIterations'] == 8 and abs(m['engine']['fixedTimeStep'] - 0.0165) < 1e-9
        assert len(g['episodes']) == 3 and g['obs_dim'] == {'AntGatherBulletEnv-v0': 46, 'AntMazeBulletEnv-v0': 38, 'PointGatherBulletEnv-v0': 18, 'AntFlagrunBulletEnv-v0': 28,
                                                            'AntMazeMjEnv-v0': 60, 'AntMjEnv': 29}[g['env_id']]
        n = 0
        for ep in g['episodes']:
            for r in ep['steps']:
                n += 1
                assert len(r['qpos']) == (15 if ant else 7) and len(r['qvel']) == (14 if ant else 6) and len(r['qpos_after']) == len(r['qpos'])
                assert abs(np.linalg.norm(r['qpos'][3:7]) - 1) < 1e-9   # quaternion x, y, z, w
                assert len(r['action']) == g['act_dim'] and len(r['obs']) == g['obs_dim'] and isinstance(r['done'], bool)
                t = r['task']
                assert 'initial_z' in t
                if 'Gather' in g['env_id']:
                    assert len(r['items']) == 16 and len(r['items_after']) == 16 and set(r['info']) == {'food_rew', 'dead_rew'}
                if 'Maze' in g['env_id']:
                    assert len(r['target']) == 2 and r['walk_target'] == r['target'] and 'potential' in t and 't' in t and len(t['feet_contact']) == 4
                if 'Flagrun' in g['env_id']:
                    assert {'potential', 'steps_since_goal_change', '_rewarded', '_sq_dist_goal', '_goal_start_pos', 'n_goals_pending', 'next_goal'} <= set(t)
        assert n >= 60   # (episodes end early where the ant falls over: AntMaze / AntFlagrun start at z 0.25, ant_maze_bullet_env.py:27)


def test_replay_agrees_with_the_engine_that_made_the_records_and_is_red_for_another(records):
    rep, med = pybullet_replay.replay(records, PERTURBED)
    for name, r in rep.items():
        assert r['done_flips'] == 0 and r['steps'] >= 60, (name, r)
        for k in ('qpos', 'qvel', 'obs', 'rew'):
            assert r[k]['max'] < 1e-9, (name, k, r[k])   # same engine, same state: the replay reconstructs EVERYTHING a step reads
        assert r['feet_flips'] == 0 and (r['walk_target_dist'] is None or r['walk_target_dist']['max'] < 1e-9)   # ... and upstream's own bookkeeping after the step
        if name in ANTS:
            assert abs(r['density_estimate'] - 850.0) < 1e-6 and r['density_hypothesis'] == 1000
            assert r['engine_equals_build'] == {'fixedTimeStep': True, 'numSubSteps': True, 'numSolverIterations': False, 'gravityAccelerationZ': True, 'contactERP': False}
    rep0, med0 = pybullet_replay.replay(records)   # the build's default specification against an engine that is NOT it: the report must say so
    over = {k: v for k, v in med0.items() if v > pybullet_replay.TOL[k[1]]}
    for name in ANTS:
        assert (name, 'qpos') in over and (name, 'qvel') in over, (name, med0)
    assert all(rep0[n]['steps'] == rep[n]['steps'] for n in rep)


def test_fit_recovers_the_engine_from_the_generators_own_json(records):
    path = [f for f in records if 'AntGather' in f][0]
    kind, steps = fit_model.load_steps(path, 300)
    assert len(steps) >= 100
    fitted, before, after = fit_model.fit(fit_model.Replay(kind, steps), list(PERTURBED), verbose=False)
    assert before[0] > 1e-3 and after[0] < 1e-7, (before, after)
    assert fitted['solver_iters'] == 8
    for k in ('density', 'contact_erp', 'friction_ground'):
        assert abs(fitted[k] - PERTURBED[k]) <= 0.02 * PERTURBED[k], (k, fitted[k])
    rep, med = pybullet_replay.replay([path], fitted)   # ... and the replay under the fitted model is green
    assert all(v <= pybullet_replay.TOL[k[1]] for k, v in med.items()), med


def test_standin_records_never_become_fixtures(records, tmp_path):
    before = golden_digest()
    p = run_generator(GOLDEN, '--steps', '2', '--seeds', '1', '--no-optional')   # the generator's default --out, with the stand-ins installed
    assert p.returncode != 0 and 'scratch directory' in p.stderr, p.stderr[-500:]
    assert golden_digest() == before and not [f for f in os.listdir(GOLDEN) if f.startswith('pybullet_')]
    # a stand-in record that found its way there would be refused by the replay test (it checks `versions.standin` before anything else)
    import test_pybullet_golden as T
    src = open(os.path.join(ROOT, 'tests', 'test_pybullet_golden.py')).read()
    assert "get('standin')" in src and T.FILES == []
