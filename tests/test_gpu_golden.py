"""The reference's own numbers on the DEVICE: every case of tests/golden/{sense_walls, food_sensor, maze_step, target_vec_spot, pointbot_state}.json
(values the reference's in-tree Python produced: tests/golden/make_golden.py) is written into the env's records with `hrl_set_state`, observed with
`hrl_observe` -- the reference's teleport + calc_state + _get_obs (ant_maze_bullet_env.py:117-121) -- through the C-ABI, and compared with the
reference's outputs at 2e-5 (fp32 device pipeline against float64 numpy; readings that flip at a bin edge / quadrant boundary are counted, <= 2 per
fixture).  The chain device == fp32 oracle == fixtures (tests/test_gpu_parity.py, tests/test_oracle_golden.py) is closed here directly."""
import numpy as np
import pytest
import torch

import golden_replay
from hrl_pybullet_envs_amd import _capi as K

pytestmark = pytest.mark.gpu


class GpuSide:
    def __init__(self, cfg):
        from hrl_pybullet_envs_amd.vec_env import BatchedEnv
        self.g = BatchedEnv(cfg, 'cuda:0')
        self.n = cfg.num_envs

    def set(self, qpos, qvel, items=None, aux3=None, initial_z=None):
        g = self.g
        if initial_z is not None:
            g.state[:, K.HRL_INITZ_OFF] = initial_z
        if items is not None:
            g.items[:, :items.shape[1]] = torch.from_numpy(items).cuda()
        if aux3 is not None:
            g.aux[:, 3] = torch.from_numpy(aux3).cuda()
        g.set_state(torch.from_numpy(qpos), torch.from_numpy(qvel), observe=False)   # hrl_set_state

    def observe(self):
        obs = self.g.observe().cpu().numpy().copy()   # hrl_observe
        self.g.close()
        return obs


def test_reference_fixtures_on_the_device():
    from hrl_pybullet_envs_amd import _lib
    golden_replay.check(golden_replay.replay_all(GpuSide, _lib.default_config))


def test_set_state_returns_the_observation_of_the_new_state():
    """BatchedEnv.set_state() = teleport + observation, as the reference's reset does after resetBasePositionAndOrientation
    (ant_maze_bullet_env.py:117-121); state, items and counters are what was written, nothing else moved; equal to the oracle's observation."""
    import orc
    from hrl_pybullet_envs_amd import _lib
    from hrl_pybullet_envs_amd.vec_env import BatchedEnv
    for kind in (K.HRL_ANT_GATHER, K.HRL_ANT_MAZE, K.HRL_POINT_GATHER, K.HRL_ANT_FLAT, K.HRL_ANT_MAZE_MJ, K.HRL_ANT_FLAGRUN):
        n = 64
        g = BatchedEnv(_lib.default_config(kind, num_envs=n, seed=5, auto_reset=1), 'cuda:0')
        o = orc.OracleEnv(orc.default_config(kind, num_envs=n, seed=5, auto_reset=1), np.float32)
        g.reset(); o.reset()
        rng = np.random.RandomState(kind)
        for t in range(5):
            a = rng.uniform(-1, 1, (n, o.ad)).astype(np.float32)
            g.step(torch.from_numpy(a).cuda()); o.step(a)
        qpos, qvel = g.get_state()
        qpos, qvel = qpos.cpu().numpy(), qvel.cpu().numpy()
        assert np.array_equal(qpos, o.state[:, :15]) and np.array_equal(qvel, o.state[:, 15:29])
        qpos[:, 0:2] += np.float32(0.37); qpos[:, 3:7] = np.array([0, 0, 0.1, np.sqrt(1 - 0.01)], np.float32)
        items0, aux0, rew0 = g.items.clone(), g.aux.clone(), g.reward.clone()
        obs = g.set_state(torch.from_numpy(qpos), torch.from_numpy(qvel)).cpu().numpy()
        o.state[:, :15] = qpos; o.state[:, 15:29] = qvel
        o.observe()
        assert np.array_equal(obs, o.obs, equal_nan=True), kind
        assert np.array_equal(g.state.cpu().numpy(), o.state) and torch.equal(g.items, items0) and torch.equal(g.aux, aux0) and torch.equal(g.reward, rew0)
        mask = torch.ones(n, dtype=torch.uint8); mask[3] = 0
        g.state[:, 0] += 0.5; o.state[:, 0] += np.float32(0.5)
        obs2 = g.observe(mask).cpu().numpy(); o.observe(mask.numpy())
        assert np.array_equal(obs2, o.obs, equal_nan=True) and np.array_equal(obs2[3], obs[3], equal_nan=True)
        g.close()
