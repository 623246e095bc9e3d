"""N > 1 path on CPU: two gloo ranks, each steps its contiguous shard of envs and all-gathers episode returns.
The stepping engine here is the lock-step host build of the product's kernel phases (tests/emu); on the GPU box the
same dist.py code runs over RCCL.  Results must be invariant to the number of ranks (RNG keyed by global env id)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, total, steps, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import emu_env
    import orc
    from hrl_pybullet_envs_amd import _capi as K
    from hrl_pybullet_envs_amd.dist import all_gather_returns, init_distributed, shard_range
    r, w, _ = init_distributed(world, backend='gloo')
    off, cnt = shard_range(total, r, w)
    env = emu_env.EmuEnv(orc.default_config(K.HRL_ANT_GATHER, num_envs=cnt, seed=4, auto_reset=1, env_id_offset=off, max_episode_steps=20))
    env.reset()
    acts = np.random.RandomState(0).uniform(-1, 1, (steps, total, 8)).astype(np.float32)
    for t in range(steps):
        env.step(acts[t, off:off + cnt])
    counts = [shard_range(total, k, w)[1] for k in range(w)]  # uneven when total % world != 0: padded inside
    gathered = all_gather_returns(torch.from_numpy(env.info[:, 2].copy()), w, counts=counts)
    dist.barrier()
    if r == 0:
        np.save(os.path.join(out_dir, 'gathered.npy'), gathered.numpy())
    np.save(os.path.join(out_dir, f'state{r}.npy'), env.state)
    np.save(os.path.join(out_dir, f'final{r}.npy'), np.concatenate([env.final_obs, env.truncated[:, None].astype(np.float32)], axis=1))
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize('total', [12, 13])  # 13: uneven shards (7 + 6), the gather pads to the largest
def test_two_rank_sharding_matches_single_process(tmp_path, total):
    steps, world = 30, 2
    mp.spawn(_worker, args=(world, _free_port(), total, steps, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import emu_env
    import orc
    from hrl_pybullet_envs_amd import _capi as K
    full = emu_env.EmuEnv(orc.default_config(K.HRL_ANT_GATHER, num_envs=total, seed=4, auto_reset=1, max_episode_steps=20))
    full.reset()
    acts = np.random.RandomState(0).uniform(-1, 1, (steps, total, 8)).astype(np.float32)
    for t in range(steps):
        full.step(acts[t])
    st = np.concatenate([np.load(tmp_path / f'state{r}.npy') for r in range(world)])
    assert np.array_equal(st, full.state)
    fin = np.concatenate([np.load(tmp_path / f'final{r}.npy') for r in range(world)])   # terminal observations (the 20-step limit has passed) + last truncation flags
    assert np.array_equal(fin[:, :-1], full.final_obs) and np.array_equal(fin[:, -1], full.truncated) and np.any(full.final_obs != 0)
    assert np.array_equal(np.load(tmp_path / 'gathered.npy'), full.info[:, 2])
