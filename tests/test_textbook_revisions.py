"""oracle/textbook_ref.c is the independent statement of the rigid-body specification: it changes only when the MODEL changes, every revision is in
the ledger tests/golden/textbook_revisions.json with the change that forced it, and (from round 6 on) a commit that revises it touches neither the
kernels nor the optimised oracle -- the textbook form lands first, the optimised forms are then made to agree with it."""
import hashlib
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LEDGER = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'textbook_revisions.json')))


def test_the_textbook_reference_is_the_ledgers_last_revision():
    sha = hashlib.sha256(open(os.path.join(ROOT, 'oracle', 'textbook_ref.c'), 'rb').read()).hexdigest()
    revs = LEDGER['revisions']
    assert sha == revs[-1]['sha256'], ('oracle/textbook_ref.c changed: add a revision (sha256, why: the MODEL change that forced it) to '
                                       'tests/golden/textbook_revisions.json -- in a commit that touches neither csrc/ nor oracle/orc_impl.h')
    assert len({r['sha256'] for r in revs}) == len(revs) and all(r['why'] and len(r['sha256']) == 64 for r in revs)


def test_revisions_since_the_rule_do_not_share_a_commit_with_the_kernels():
    if not os.path.isdir(os.path.join(ROOT, '.git')):
        pytest.skip('no git history here (the GPU box gets a snapshot)')
    since = LEDGER['rule_since']
    try:
        log = subprocess.run(['git', '-C', ROOT, 'log', '--format=%H', f'{since}..HEAD', '--', 'oracle/textbook_ref.c'], capture_output=True, text=True, check=True).stdout.split()
    except (subprocess.CalledProcessError, FileNotFoundError):
        pytest.skip('git log unavailable')
    for c in log:
        files = subprocess.run(['git', '-C', ROOT, 'show', '--name-only', '--format=', c], capture_output=True, text=True, check=True).stdout.split()
        shared = [f for f in files if f.startswith('hrl_pybullet_envs_amd/csrc/') or f == 'oracle/orc_impl.h']
        assert not shared, f'commit {c[:7]} revises oracle/textbook_ref.c together with {shared}: the textbook form goes first, in a commit of its own'
