"""tests/golden/spec_fingerprint.json: sha256 of the buffers of 16 envs of every kind after the reset and 1, 10 and 80 steps (tests/spec_fingerprint.py).
The fp32 CPU oracle and the host executor of the kernel phases must reproduce it here, the device does in tests/test_gpu_parity.py: the three
implementations agree not only with one another (the parity tests) but with a committed record of the specification -- a spec change has to show up
in the fixture's diff."""
import json

import numpy as np

import emu_env
import orc
import spec_fingerprint as S


def test_oracle_reproduces_the_committed_fingerprint():
    want = json.load(open(S.PATH))
    got = S.fingerprint(lambda cfg: orc.OracleEnv(cfg, np.float32))
    assert got == want, [k for k in want if got.get(k) != want[k]]
    digests = [v for k, d in want.items() for m, v in d.items() if not (k.endswith('+v7') and m == 'reset')]   # (a reset does not depend on the engine parameters)
    assert len(set(digests)) == len(digests)   # every digest is its own: nothing is stuck


def test_host_executor_reproduces_the_committed_fingerprint():
    want = json.load(open(S.PATH))
    got = S.fingerprint(lambda cfg: emu_env.EmuEnv(cfg))
    assert got == want, [k for k in want if got.get(k) != want[k]]
