"""The reference's fixtures replayed on the product's wave phases under the host executor (tests/golden_replay.py): no GPU needed.
The same replay through the C-ABI on the device: tests/test_gpu_golden.py."""
import numpy as np

import emu_env
import golden_replay
import orc
from hrl_pybullet_envs_amd import _capi as K


class EmuSide:
    def __init__(self, cfg):
        self.e = emu_env.EmuEnv(cfg)
        self.n = cfg.num_envs

    def set(self, qpos, qvel, items=None, aux3=None, initial_z=None):
        e = self.e
        e.state[:, 0:15] = qpos; e.state[:, 15:29] = qvel
        if initial_z is not None:
            e.state[:, K.HRL_INITZ_OFF] = initial_z
        if items is not None:
            e.items[:, :items.shape[1]] = items
        if aux3 is not None:
            e.aux[:, 3] = aux3

    def observe(self):
        return self.e.observe().copy()


def test_reference_fixtures_on_the_wave_phases():
    golden_replay.check(golden_replay.replay_all(EmuSide, orc.default_config))


def test_observe_writes_nothing_but_the_observation():
    """hrl_observe: state, items, counters untouched; masked rows keep their observation; equal to the oracle's make_obs bit for bit."""
    for kind in (K.HRL_ANT_GATHER, K.HRL_ANT_MAZE, K.HRL_POINT_GATHER, K.HRL_ANT_FLAT, K.HRL_ANT_MAZE_MJ, K.HRL_ANT_FLAGRUN):
        cfg = orc.default_config(kind, num_envs=16, seed=5, auto_reset=1)
        o, e = orc.OracleEnv(cfg, np.float32), emu_env.EmuEnv(cfg)
        o.reset(); e.reset()
        rng = np.random.RandomState(kind)
        for t in range(5):
            a = rng.uniform(-1, 1, (16, o.ad)).astype(np.float32)
            o.step(a); e.step(a)
        for env in (o, e):   # a teleport, as ant_maze_bullet_env.py:117 does
            env.state[:, 0:2] += np.float32(0.37); env.state[:, 5] = np.float32(0.1); env.state[:, 6] = np.float32(np.sqrt(1 - 0.01))
        before = {k: getattr(e, k).copy() for k in ('state', 'items', 'aux', 'rew', 'done', 'info', 'obs')}
        mask = np.ones(16, np.uint8); mask[3] = 0
        o.observe(mask); e.observe(mask)
        assert np.array_equal(o.obs, e.obs, equal_nan=True)
        for k in ('state', 'items', 'aux', 'rew', 'done', 'info'):
            assert np.array_equal(getattr(e, k), before[k], equal_nan=True), (kind, k)
        assert np.array_equal(e.obs[3], before['obs'][3], equal_nan=True) and not np.array_equal(e.obs[0], before['obs'][0])
