"""The rigid-body half of the oracle against the REAL reference -- when its fixtures exist.

tests/golden/pybullet_<env>.json are written by tools/make_pybullet_golden.py on a machine that has pybullet, gym and the reference installed
(none of which the build or GPU images have: ModuleNotFoundError, nothing denied).  Until someone commits them the replay test below SKIPS and the
rigid-body step stays "parity unpinned" (DESIGN.md 6); it never passes on fabricated data.  With the files present it
  * settles SURVEY Appendix A.4 from the recorded link masses (density 1000 kg/m^3, what this build assumes, against MuJoCo's 5),
  * checks the recorded engine parameters against the ones the build restates from memory,
  * replays every recorded step on the fp64 oracle from the identical (qpos, qvel, items, action, task bookkeeping) and reports the deviation of
    qpos', qvel', obs, reward and done per quantity (written to profiles/pybullet_deviation.json), failing where it exceeds the tolerance SURVEY 8d
    states (tests/pybullet_replay.py).
The whole road -- generator, this replay, the fit -- runs end to end in tests/test_pin_road_dry_run.py against stand-in packages backed by the oracle."""
import glob
import json
import os
import subprocess
import sys

import pytest

import pybullet_replay
from pybullet_replay import TOL, ant_masses

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden', 'pybullet_*.json')))


def test_the_generator_refuses_to_run_without_pybullet_and_writes_nothing():
    """In this image pybullet / gym are absent: the script must say so and leave tests/golden alone (no fabricated fixtures, ever)."""
    try:
        import pybullet  # noqa: F401
        pytest.skip('pybullet is importable here: run tools/make_pybullet_golden.py and commit its output instead')
    except ImportError:
        pass
    before = sorted(os.listdir(os.path.join(ROOT, 'tests', 'golden')))
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'make_pybullet_golden.py'), '--steps', '2', '--seeds', '1'], capture_output=True, text=True)
    assert p.returncode != 0 and 'nothing written' in p.stderr and 'pybullet' in p.stderr
    assert sorted(os.listdir(os.path.join(ROOT, 'tests', 'golden'))) == before


def test_mass_model_hypotheses_are_far_apart():
    """What the recorded masses will decide between: 182.2 kg at 1000 kg/m^3 (this build) against 0.91 kg at MuJoCo's 5."""
    m1000, m5 = ant_masses(1000.0), ant_masses(5.0)
    assert abs(m1000[0] + 4 * (m1000[1] + m1000[2]) - 182.2) < 0.1 and abs(m5[0] + 4 * (m5[1] + m5[2]) - 0.911) < 0.001


@pytest.mark.skipif(not FILES, reason='no tests/golden/pybullet_*.json: run tools/make_pybullet_golden.py where pybullet + gym + the reference are installed '
                                      '(the rigid-body step stays parity-unpinned until then)')
def test_oracle_against_recorded_pybullet_steps():
    for path in FILES:   # records of the dry run's stand-in packages (tests/pybullet_standin.py: the oracle itself) are not fixtures
        assert not (json.load(open(path)).get('versions') or {}).get('standin'), f'{path} was recorded from the stand-in packages of the dry run, not from pybullet: remove it'
    report, worst = pybullet_replay.replay(FILES)
    os.makedirs(os.path.join(ROOT, 'profiles'), exist_ok=True)
    with open(os.path.join(ROOT, 'profiles', 'pybullet_deviation.json'), 'w') as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))
    over = {k: v for k, v in worst.items() if v > TOL[k[1]]}
    assert not over, (f'median one-step deviation from pybullet beyond the stated tolerance {TOL}: {over} (full report: profiles/pybullet_deviation.json; '
                      'the engine parameters nothing in the reference tree decides are hrl_model fields: tests/tools/fit_model.py <fixture> fits them)')
    assert all(r.get('density_hypothesis', 1000) == 1000 for r in report.values()), 'Bullet used another density than this build assumes (SURVEY A.4)'
