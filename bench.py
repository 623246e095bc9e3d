#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched AntGather step on N MI355X (BASELINE.json metric).

A "step" is one hrl_step() over the shard's envs: one HIP kernel launch that advances every env by one env step
(4 physics substeps + observation/reward).  Workload at N=1: AntGatherBulletEnv-v0, 4096 envs (BASELINE.json
configs[2], the config the metric is quoted on); weak scaling: every rank owns 4096 envs, RNG keyed by global id.
Inputs (state, items, pre-generated U(-1,1) actions) are resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--envs 4096] [--kind gather|flat|maze|point|maze_mj|flagrun]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from hrl_pybullet_envs_amd import _capi as K  # noqa: E402
from hrl_pybullet_envs_amd import _lib  # noqa: E402
from hrl_pybullet_envs_amd.dist import ReturnGatherer, init_distributed  # noqa: E402
from hrl_pybullet_envs_amd.vec_env import BatchedEnv  # noqa: E402

KINDS = {'flat': K.HRL_ANT_FLAT, 'gather': K.HRL_ANT_GATHER, 'maze': K.HRL_ANT_MAZE, 'point': K.HRL_POINT_GATHER,
         'maze_mj': K.HRL_ANT_MAZE_MJ, 'flagrun': K.HRL_ANT_FLAGRUN}
NAMES = {'flat': 'AntMjEnv (flat ground)', 'gather': 'AntGatherBulletEnv-v0', 'maze': 'AntMazeBulletEnv-v0',
         'point': 'PointGatherBulletEnv-v0', 'maze_mj': 'AntMazeMjEnv-v0', 'flagrun': 'AntFlagrunBulletEnv-v0'}
# algorithmic HBM bytes per env-step, fp32, state read once + written once (SURVEY.md 8d / BASELINE.md 4)
ALG_BYTES = {'gather': 581, 'flat': 385, 'maze': 429, 'point': 317,
             'maze_mj': 148 + 8 + 116 + 240 + 5, 'flagrun': 148 + 4 + 116 + 112 + 5}  # same accounting: read state+act(+target/goal index), write state+obs+rew+done
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def cpu_baseline(kind, seconds_budget=15.0):
    """Times the CPU oracle (a port: the pybullet reference is not installable here) on ONE host core, N = 1 env,
    the like-for-like of the reference's single-process loop (README.md:29-34).  Bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import orc
    L = orc.lib()
    L.orc_bench_f32.restype = C.c_double
    cfg = orc.default_config(KINDS[kind], num_envs=1, seed=0, auto_reset=1)
    cs = C.c_double()
    dt = L.orc_bench_f32(C.byref(cfg), 2000, 1, C.byref(cs))  # calibrate
    steps = max(2000, int(2000 / dt * seconds_budget))
    dt = L.orc_bench_f32(C.byref(cfg), steps, 1, C.byref(cs))
    out = {'value': steps / dt, 'unit': 'env-steps/s', 'cores': 1, 'kind': 'port',
           'sample': f'CPU oracle (oracle/liborc.so, fp32), {NAMES[kind]}, 1 env x {steps} random-action steps, '
                     f'1 thread, {dt:.1f} s'}
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    ncpu = max(1, min(ncpu, 16))  # the GPU box gives one GPU's job a 16-core share
    cfg_n = orc.default_config(KINDS[kind], num_envs=4096, seed=0, auto_reset=1)
    L.orc_bench_f32(C.byref(cfg_n), 2, ncpu, C.byref(cs))  # start the OpenMP team
    dtn = L.orc_bench_f32(C.byref(cfg_n), 40, ncpu, C.byref(cs))  # calibrate
    nsteps = max(40, int(40 / dtn * 6.0))
    dtn = L.orc_bench_f32(C.byref(cfg_n), nsteps, ncpu, C.byref(cs))
    out['all_cores'] = {'value': 4096 * nsteps / dtn, 'cores': ncpu,
                        'sample': f'same oracle, 4096 envs x {nsteps} steps, OpenMP over {ncpu} threads, {dtn:.1f} s'}
    out['reference'] = reference_probe()
    return out


def reference_probe():
    """the pybullet reference's README loop can only be timed where pybullet, gym and the reference
    package are importable; that is never the case on the build/GPU images (no network), so this reports why not."""
    missing = []
    for mod in ('gym', 'pybullet', 'pybullet_envs', 'hrl_pybullet_envs'):
        try:
            __import__(mod)
        except Exception as e:  # noqa: BLE001
            missing.append(f'{mod} ({type(e).__name__})')
    if missing:
        return 'unavailable on this box: ' + ', '.join(missing)
    import gym  # pragma: no cover
    import numpy as np  # pragma: no cover
    env = gym.make('AntGatherBulletEnv-v0'); env.seed(0); env.reset()  # README.md:24-34  # pragma: no cover
    t0, n = time.perf_counter(), 0  # pragma: no cover
    for _ in range(1000):  # pragma: no cover
        _, _, d, _ = env.step(np.random.uniform(-1, 1, 8)); n += 1
        if d:
            env.reset()
    return {'value': n / (time.perf_counter() - t0), 'unit': 'env-steps/s', 'cores': 1, 'kind': 'reference'}  # pragma: no cover


def profiled_traffic(kind, n):
    """HBM bytes per launch of k_step from the committed rocprofv3 PMC summary (profiles/), scaled to n envs.
    bench.py cannot collect counters itself; None when no summary for this kernel is committed."""
    path = os.path.join(ROOT, 'profiles', 'pmc_summary.json')
    try:
        with open(path) as f:
            d = json.load(f)[kind]
        return (d['fetch_bytes_per_env'] + d['write_bytes_per_env']) * n
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=200)
    ap.add_argument('--envs', type=int, default=4096, help='envs per GPU')
    ap.add_argument('--kind', default='gather', choices=sorted(KINDS))
    ap.add_argument('--gather-every', type=int, default=100, help='all-gather episode returns every K steps (N>1)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--backend', default=None, help='torch.distributed backend (default nccl = RCCL); gloo only to rehearse N>1 on a box with fewer GPUs')
    args = ap.parse_args()

    rank, world, local_rank = init_distributed(args.gpus, backend=args.backend)
    if args.backend == 'gloo':
        local_rank = local_rank % max(1, torch.cuda.device_count())
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    n = args.envs
    cfg = _lib.default_config(KINDS[args.kind], num_envs=n, seed=0, auto_reset=1, env_id_offset=rank * n)
    env = BatchedEnv(cfg, dev)
    env.reset()
    # pre-generated actions, resident in HBM: [T, N, A] ~ U(-1, 1), seed 0 (+rank)
    T = 256
    gen = torch.Generator(device=dev).manual_seed(rank)
    actions = torch.rand(T, n, env.act_dim, device=dev, generator=gen) * 2 - 1
    gatherer = ReturnGatherer(env, world) if world > 1 else None

    def run(k0, k):
        for t in range(k0, k0 + k):
            env.step(actions[t % T])
            if gatherer is not None and (t + 1) % args.gather_every == 0:
                gatherer.launch()

    run(0, args.warmup)
    torch.cuda.synchronize(dev)
    if world > 1:
        torch.distributed.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()  # the kernels are launched on torch's current stream, which is where these events are recorded
    run(args.warmup, args.steps)
    ev1.record()
    torch.cuda.synchronize(dev)
    if world > 1:
        torch.distributed.barrier()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if world > 1:
        tt = torch.tensor([wall], device=dev if torch.distributed.get_backend() == 'nccl' else 'cpu', dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        wall = float(tt.item())
    assert bool(torch.isfinite(env.state).all()), 'non-finite state after the timed region'
    if world > 1:  # all collectives are done: leave the group before rank 0 spends ~25 s of host time on the CPU baseline
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()

    if rank == 0:
        total_steps = world * n * args.steps
        launch_s = dev_ms / 1e3 / args.steps  # one kernel per step: HIP-event time / launches
        achieved = ALG_BYTES[args.kind] * n / launch_s / 1e9
        out = {
            'metric': 'env-steps/sec, AntGatherBulletEnv-v0 @4096 envs, 1/2/4/8 MI355X' if args.kind == 'gather' and n == 4096
            else f'env-steps/sec, {NAMES[args.kind]} @{n} envs/GPU',
            'value': total_steps / wall, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': wall / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'{NAMES[args.kind]}, {n} envs per GPU, U(-1,1) actions pre-generated on device, auto-reset, '
                                   f'max_episode_steps 2000', 'envs_per_gpu': n, 'global_envs': world * n,
                       'substeps_per_step': 4, 'parallelism': f'env-sharded x{world}, no data-path collective; '
                                                              f'RCCL all-gather of episode returns every {args.gather_every} steps'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': profiled_traffic(args.kind, n), 'kernel': 'k_step', 'kernel_avg_us': launch_s * 1e6,
                         'algorithmic_bytes_per_launch': ALG_BYTES[args.kind] * n,
                         'note': 'VALU-issue/latency-bound by construction (~9.4e3 VALU wave-instructions, ~1e5 flop per env-step, ~170 flop/B): see DESIGN.md 5'},
        }
        if not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.kind)
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
