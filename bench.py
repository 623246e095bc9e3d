#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched AntGather step on N MI355X (BASELINE.json metric).

A "step" is one hrl_step() over the shard's envs: one HIP kernel launch that advances every env by one env step
(4 physics substeps + observation/reward).  Workload at N=1: AntGatherBulletEnv-v0, 4096 envs (BASELINE.json
configs[2], the config the metric is quoted on); weak scaling: every rank owns 4096 envs, RNG keyed by global id.
Inputs (state, items, pre-generated U(-1,1) actions) are resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--envs 4096]
                    [--kind gather|flat|maze|point|maze_mj|flagrun|mixed] [--backend nccl|gloo]

`--kind mixed` is BASELINE.json configs[4] per GPU: the first half of the shard AntGather, the second half PointGather
(two hrl_step launches per step on two HIP streams, so the kernels overlap on the chip), one JSON line.
`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process starts the N ranks itself
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`) BEFORE anything touches
the GPU, passes their output through and exits with their status.  Under torch.distributed.run it is a rank.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

KIND_NAMES = ['flat', 'gather', 'maze', 'point', 'maze_mj', 'flagrun', 'mixed']
NAMES = {'flat': 'AntMjEnv (flat ground)', 'gather': 'AntGatherBulletEnv-v0', 'maze': 'AntMazeBulletEnv-v0',
         'point': 'PointGatherBulletEnv-v0', 'maze_mj': 'AntMazeMjEnv-v0', 'flagrun': 'AntFlagrunBulletEnv-v0',
         'mixed': 'AntGatherBulletEnv-v0 + PointGatherBulletEnv-v0 mixed batch'}
# algorithmic HBM bytes per env-step, fp32, state read once + written once (SURVEY.md 8d / BASELINE.md 4)
ALG_BYTES = {'gather': 581, 'flat': 385, 'maze': 429, 'point': 317,
             'maze_mj': 148 + 8 + 116 + 240 + 5, 'flagrun': 148 + 4 + 116 + 112 + 5}  # same accounting: read state+act(+target/goal index), write state+obs+rew+done
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MAX_CLOCK_GHZ = 2.4     # MI355X_MICROARCH.md chip table
N_SIMD = 256 * 4        # 256 CUs x 4 SIMD-32
VALU_ISSUE_CYCLES = 2   # a wave64 VALU instruction occupies its SIMD-32 for 2 cycles


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=200)
    ap.add_argument('--envs', type=int, default=4096, help='envs per GPU')
    ap.add_argument('--kind', default='gather', choices=sorted(KIND_NAMES))
    ap.add_argument('--gather-every', type=int, default=100, help='all-gather episode returns every K steps (N>1)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--model', action='append', default=[], metavar='FIELD=VALUE', help='experiments only: an hrl_model field other than the default (e.g. '
                    '--model linear_damping=0 --model angular_damping=0); the line then says so in config.model_overrides and is not the headline')
    ap.add_argument('--backend', default=None, help='torch.distributed backend (default nccl = RCCL); gloo only to rehearse N>1 on a box with fewer GPUs')
    ap.add_argument('--settle', type=int, default=300, help='untimed steps before --warmup, whatever the caller passes for --warmup: the timed window then '
                    'sees ants that stand (the first ~50 steps after a reset have fewer contacts and shorter launches); 0 = time the post-reset transient')
    ap.add_argument('--watchdog', type=float, default=900.0, help='seconds a rank may go WITHOUT PROGRESS (a new stage, or another 64 launches queued) before '
                    'it reports the stage it hangs in and exits 3 (0 = off)')
    ap.add_argument('--stall-in', default=None, help=argparse.SUPPRESS)  # test hook: sleep forever on entering this stage (tests/test_gpu_envs.py)
    return ap.parse_args()


class Watchdog:
    """A rank that stops making progress (a collective some rank never joined, a rendezvous that never completes) would otherwise sit until the
    driver's limit with nothing on its output.  A STALL detector, not a budget: every new stage (`at`) and every heartbeat of the stepping loop
    (`beat`) moves the deadline `seconds` ahead, so a long run that keeps going is never cut short; a daemon thread that finds the deadline passed
    prints which stage the rank was in and ends THIS process with status 3 (os._exit: no interpreter teardown that could block on the GPU; never
    a re-exec).  torch.distributed.run then ends the others."""

    def __init__(self, seconds, rank, stall_in=None):
        import threading
        self.stage, self.rank, self.seconds, self.stall_in = 'start', rank, seconds, stall_in
        self._done = threading.Event()
        self._deadline = time.monotonic() + seconds
        if seconds > 0:
            threading.Thread(target=self._watch, daemon=True).start()

    def _watch(self):
        while True:
            left = self._deadline - time.monotonic()
            if left <= 0:
                break
            if self._done.wait(min(left, 1.0)):
                return
        sys.stderr.write(f'bench.py watchdog: rank {self.rank} made no progress in stage "{self.stage}" for {self.seconds:.0f} s, exiting 3\n')
        sys.stderr.flush()
        os._exit(3)

    def beat(self):
        self._deadline = time.monotonic() + self.seconds

    def at(self, stage):
        self.stage = stage
        self.beat()
        if self.stall_in == stage:   # test hook: a rank that hangs here
            while True:
                time.sleep(1.0)

    def done(self):
        self._done.set()


def count_gpus(kfd_root=None):
    """GPUs of this machine WITHOUT touching the GPU: the KFD topology nodes whose `simd_count` is not 0 (CPU nodes have 0), narrowed by
    ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES when set.  None when the topology cannot be read (no amdgpu driver here: nothing to decide from)."""
    root = kfd_root or os.environ.get('HRL_KFD_ROOT') or '/sys/class/kfd/kfd/topology/nodes'
    try:
        nodes = sorted(os.listdir(root))
    except OSError:
        return None
    n = 0
    for node in nodes:
        try:
            with open(os.path.join(root, node, 'properties')) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
        except OSError:
            continue
        n += int(props.get('simd_count', '0')) > 0
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(',') if x.strip() != '']))
    return n


def preflight(ranks, backend):
    """Before anything touches the GPU or a rendezvous starts: a job of `ranks` RCCL ranks on a machine with fewer GPUs would sit in
    init_process_group / its first collective until the watchdog ends it 900 s later.  One line and exit code 2 instead.  (gloo rehearsals share
    the card on purpose; an unreadable topology decides nothing.)"""
    have = count_gpus()
    if backend == 'gloo' or have is None or have >= ranks:
        return
    try:   # a second opinion before refusing a run: torch's own count (does not initialise the GPU on this image); whoever sees enough devices wins
        import warnings
        import torch
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')   # (a machine without a usable driver warns here; the count it returns is what is asked for)
            if torch.cuda.device_count() >= ranks:
                return
    except Exception:  # noqa: BLE001
        pass
    print(f'bench.py: --gpus {ranks} asks for {ranks} ranks (one GPU each), this machine shows {have} GPU(s) '
          f'(KFD topology, ROCR_/HIP_VISIBLE_DEVICES): not started', file=sys.stderr, flush=True)
    sys.exit(2)


def spawn_ranks(args):
    """Parent of an N-rank run: nothing in this process has touched the GPU (no HIP call, no torch.cuda.*), so the
    ranks are plain child processes; the parent never re-executes itself."""
    preflight(args.gpus, args.backend)
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')  # dmabuf IPC: RCCL across processes needs it on this driver
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def cpu_baseline(kind, seconds_budget=15.0):
    """Times the CPU oracle (a port: the pybullet reference is not installable here) on ONE host core, N = 1 env,
    the like-for-like of the reference's single-process loop (README.md:29-34).  Bounded sample."""
    from hrl_pybullet_envs_amd import _capi as K
    kinds = {'flat': K.HRL_ANT_FLAT, 'gather': K.HRL_ANT_GATHER, 'maze': K.HRL_ANT_MAZE, 'point': K.HRL_POINT_GATHER,
             'maze_mj': K.HRL_ANT_MAZE_MJ, 'flagrun': K.HRL_ANT_FLAGRUN}
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import orc
    L = orc.lib()
    L.orc_bench_f32.restype = C.c_double
    if kind == 'mixed':  # half the sample on each env type; value = env-steps of both / time of both
        a, p = cpu_baseline('gather', seconds_budget / 2), cpu_baseline('point', seconds_budget / 2)
        va, vp = a['value'], p['value']
        out = {'value': 2.0 / (1.0 / va + 1.0 / vp), 'unit': 'env-steps/s', 'cores': 1, 'kind': 'port',
               'sample': 'equal numbers of AntGather and PointGather env-steps, one after the other on 1 thread: ' + a['sample'] + ' | ' + p['sample'],
               'all_cores': {'value': 2.0 / (1.0 / a['all_cores']['value'] + 1.0 / p['all_cores']['value']), 'cores': a['all_cores']['cores'],
                             'sample': a['all_cores']['sample'] + ' | ' + p['all_cores']['sample']},
               'reference': a['reference']}
        return out
    cfg = orc.default_config(kinds[kind], num_envs=1, seed=0, auto_reset=1)
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
    cfg_n = orc.default_config(kinds[kind], num_envs=4096, seed=0, auto_reset=1)
    L.orc_bench_f32(C.byref(cfg_n), 2, ncpu, C.byref(cs))  # start the OpenMP team
    dtn = L.orc_bench_f32(C.byref(cfg_n), 40, ncpu, C.byref(cs))  # calibrate
    nsteps = max(40, int(40 / dtn * seconds_budget * 0.4))
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


def pmc_summary(kind):
    """Committed rocprofv3 PMC summary of k_step<kind> (profiles/pmc_summary.json, tools/summarize_profile.py).
    bench.py cannot collect counters itself; {} when no summary for this kernel is committed."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_summary.json')) as f:
            return json.load(f).get(kind, {})
    except Exception:
        return {}


def roofline(kind, n, launch_s):
    """HBM roofline (the contractual one) + the VALU-issue roofline (the one that binds), SURVEY 8d "report both".

    hbm:  achieved = algorithmic bytes per launch / measured launch time, against 8 TB/s.
    valu: a wave64 VALU instruction holds its SIMD-32 for 2 cycles, so a launch needs at least
          insts_per_env x envs / 1024 SIMDs x 2 cycles of VALU issue per SIMD; frac = that / (launch time x 2.4 GHz),
          i.e. the fraction of the chip's peak VALU issue rate the launch used.  insts_per_env, wave cycles and the
          clock the chip held come from the committed PMC pass of the same kernel (profiles/)."""
    s = pmc_summary(kind)
    from hrl_pybullet_envs_amd.build import kernel_source_hash
    stale = bool(s) and s.get('source_sha256') != kernel_source_hash()
    if stale:  # the committed counters describe other code than the one benched: report none rather than stale ones
        s = {}
    alg = ALG_BYTES[kind] * n
    achieved = alg / launch_s / 1e9
    # HBM-side bytes per launch from the committed counter passes (2 x FETCH_SIZE + WRITE_SIZE: the gfx950 correction, calibrated on this access pattern,
    # profiles/r5_traffic_calibration.txt) -- only for the env count they were collected at: the fetch side has a fixed part per launch (the kernel's text,
    # fetched again by each of the eight L2s), so a per-env figure does not scale to another batch size
    same_n = s.get('envs_per_launch') == n
    traffic = (s['fetch_bytes_per_env'] + s['write_bytes_per_env']) * n if 'fetch_bytes_per_env' in s and same_n else None
    out = {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
           'traffic': traffic, 'kernel': f'k_step<{kind}>', 'kernel_avg_us': launch_s * 1e6, 'algorithmic_bytes_per_launch': alg,
           'note': 'the path is VALU-issue/latency bound (~170 flop/B, serial recursion): `valu` is the roofline that binds, DESIGN.md 5'}
    if 'fetch_bytes_per_env' in s and not same_n:
        out['traffic_note'] = f"counters were collected at {s.get('envs_per_launch')} envs per launch, not {n}: not scaled (fixed part per launch)"
    out['pmc_stale'] = stale  # True: profiles/pmc_summary.json was collected from different kernel sources; traffic / valu withheld
    if 'valu_insts_per_env' in s:
        per_simd = s['valu_insts_per_env'] * n / N_SIMD * VALU_ISSUE_CYCLES
        v = {'insts_per_env': s['valu_insts_per_env'], 'issue_cycles_per_simd': per_simd,
             'frac': per_simd / (launch_s * MAX_CLOCK_GHZ * 1e9), 'peak': 'one wave64 VALU instruction per 2 cycles per SIMD-32 at 2.4 GHz',
             'source': f"profiles/{s.get('tag', '?')}_pmc.json (SQ_INSTS_VALU per wave)"}
        if 'wave_cycles_per_env' in s and 'kernel_us_profiled' in s:
            v['held_clock_ghz'] = s['wave_cycles_per_env'] / (s['kernel_us_profiled'] * 1e3)  # lower bound: a wave lives at most the launch
            v['frac_at_held_clock'] = per_simd / (launch_s * v['held_clock_ghz'] * 1e9)
        out['valu'] = v
        # What four waves on one SIMD can issue was MEASURED (tools/micro/simd_share.hip, profiles/r3_simd_share_microbench.txt): VALU-only
        # streams 830 - 908 instructions per microsecond per SIMD, streams that mix in scalar instructions 370 - 620 VALU per microsecond.
        per_us = s['valu_insts_per_env'] * n / N_SIMD / (launch_s * 1e6)
        out['issue'] = {'valu_per_us_per_simd': per_us, 'measured_ceiling_valu_only': [830.0, 908.0], 'frac_of_measured_ceiling': per_us / 908.0,
                        'measured_ceiling_solver_row_mix': [598.0, 624.0], 'source': 'profiles/r3_simd_share_microbench.txt'}
        if 'insts_per_env' in s:
            out['issue']['all_insts_per_us_per_simd'] = s['insts_per_env'] * n / N_SIMD / (launch_s * 1e6)
            out['issue']['insts_per_env'] = s['insts_per_env']
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args))
    if 'WORLD_SIZE' in os.environ:   # a rank of the driver's own torch.distributed.run: the same check, before the rendezvous (every rank leaves with 2 at once);
        #                              the ranks of THIS node are what its GPUs must cover (LOCAL_WORLD_SIZE, set by torch.distributed.run)
        preflight(int(os.environ.get('LOCAL_WORLD_SIZE') or os.environ['WORLD_SIZE']), args.backend)

    import torch

    from hrl_pybullet_envs_amd import _capi as K
    from hrl_pybullet_envs_amd import _lib
    from hrl_pybullet_envs_amd.dist import ReturnGatherer, init_distributed
    from hrl_pybullet_envs_amd.vec_env import BatchedEnv
    kinds = {'flat': K.HRL_ANT_FLAT, 'gather': K.HRL_ANT_GATHER, 'maze': K.HRL_ANT_MAZE, 'point': K.HRL_POINT_GATHER,
             'maze_mj': K.HRL_ANT_MAZE_MJ, 'flagrun': K.HRL_ANT_FLAGRUN}

    wd = Watchdog(args.watchdog, int(os.environ.get('RANK', '0')), args.stall_in)
    wd.at('init_process_group')
    rank, world, local_rank = init_distributed(args.gpus, backend=args.backend)
    wd.at('build envs')
    if args.backend == 'gloo':
        local_rank = local_rank % max(1, torch.cuda.device_count())
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    n = args.envs
    T = 256  # pre-generated actions, resident in HBM: [T, N, A] ~ U(-1, 1), seed = rank
    gen = torch.Generator(device=dev).manual_seed(rank)
    main_stream = torch.cuda.current_stream(dev)
    if args.kind == 'mixed':  # BASELINE.json configs[4]: first half of the shard AntGather, second half PointGather
        assert n % 2 == 0, '--envs must be even for --kind mixed'
        parts = [('gather', n // 2, rank * n), ('point', n // 2, rank * n + n // 2)]
    else:
        parts = [(args.kind, n, rank * n)]
    envs = []
    for kname, cnt, off in parts:
        cfg = _lib.default_config(kinds[kname], num_envs=cnt, seed=0, auto_reset=1, env_id_offset=off)
        for kv in args.model:
            k, v = kv.split('=', 1)
            setattr(cfg.model, k, type(getattr(cfg.model, k))(float(v)))
        env = BatchedEnv(cfg, dev)
        env.reset()
        acts = torch.rand(T, cnt, env.act_dim, device=dev, generator=gen) * 2 - 1
        stream = main_stream if len(parts) == 1 else torch.cuda.Stream(device=dev)  # one HIP stream per sub-shard: the launches overlap
        envs.append((kname, env, acts, stream))
    torch.cuda.synchronize(dev)
    gatherer = ReturnGatherer([(e, s) for _, e, _, s in envs], world, counts=[n] * world) if world > 1 else None
    gather_every = max(1, min(args.gather_every, args.steps))  # a short run (the driver's --steps 20) still gathers inside the timed region
    gather_phase = gather_every // 2   # launched in the MIDDLE of its interval: a collective queued behind the window's last step would be waited for, exposed, by the
    #                                    synchronisation that ends the window (one all-gather latency in a 20-launch window); queued mid-window it overlaps the steps that follow

    def run(k0, k):
        for t in range(k0, k0 + k):
            for _, env, acts, stream in envs:
                with torch.cuda.stream(stream):
                    env.step(acts[t % T])
            if gatherer is not None and (t + 1) % gather_every == gather_phase:
                gatherer.launch()
            if (t & 63) == 63:
                wd.beat()   # the launch queue is bounded: the host gets here only as fast as the GPU works the launches off

    ranks_seen = None
    if world > 1:  # every rank reports in before anything is timed: the line shows the job really was `world` ranks on `world` devices
        wd.at('all_gather of (rank, device)')
        ranks_seen = [None] * world
        torch.distributed.all_gather_object(ranks_seen, (rank, local_rank, torch.cuda.get_device_name(dev)))
    wd.at('settle')
    run(0, args.settle)   # untimed, before the caller's warmup: the ants come to stand (the driver passes --warmup 5)
    wd.at('warmup')
    run(args.settle, args.warmup)
    n_warm_gathers = gatherer.k if gatherer is not None else 0
    torch.cuda.synchronize(dev)   # set-up work is done before the clock starts; settle, warm-up and timed launches are the SAME launch: the shipped default, no diagnostic output bound
    if world > 1:
        wd.at('barrier before the timed region')
        torch.distributed.barrier()
        torch.cuda.synchronize(dev)
    wd.at('timed region')
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in envs]  # made before the clock starts (hipEventCreate is host time, not step time)
    t0 = time.perf_counter()
    # HIP events on the stream each kernel is launched on, around launches 2..K of the timed region: an event recorded on an IDLE stream is
    # stamped at once and the first launch arrives ~20 us later (host-to-GPU dispatch latency) -- in the driver's 20-launch window that would add
    # 1 us to every launch's "duration" (profiles/r5_driverlike_kernel_trace.txt).  The wall clock (ms_per_step, value) covers all K.
    first = 1 if args.steps > 1 else 0
    run(args.settle + args.warmup, first)
    for (e0, _), (_, _, _, stream) in zip(ev, envs):
        e0.record(stream)
    run(args.settle + args.warmup + first, args.steps - first)
    for (e0, e1), (_, _, _, stream) in zip(ev, envs):
        e1.record(stream)
    torch.cuda.synchronize(dev)
    wall_own = time.perf_counter() - t0   # this rank's K steps, all its streams drained
    if world > 1:
        wd.at('barrier after the timed region')
        torch.distributed.barrier()
    wall = time.perf_counter() - t0       # the contract's bracket: barrier + synchronize on both sides (`value`); the closing barrier itself -- an all-reduce and
    #                                       a synchronisation -- is in it, `ms_per_step_before_barrier` shows the steps without it
    wd.at('reductions after the timed region')
    dev_ms = [e0.elapsed_time(e1) for e0, e1 in ev]
    if world > 1:
        tt = torch.tensor([wall, wall_own], device=dev if torch.distributed.get_backend() == 'nccl' else 'cpu', dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        wall, wall_own = float(tt[0].item()), float(tt[1].item())
    for _, env, _, _ in envs:
        assert bool(torch.isfinite(env.state).all()), 'non-finite state after the timed region'
    gathered_ok = None
    rccl = None
    if gatherer is not None:
        g = gatherer.latest()
        gathered_ok = bool(g is not None and g.numel() == world * n and torch.isfinite(g).all())
        times = gatherer.gather_times_us(first=n_warm_gathers)   # the collectives of the timed region (launch index >= the warmup's count), HIP events on the side stream
        t_first = args.settle + args.warmup
        expected = sum(1 for t in range(t_first, t_first + args.steps) if (t + 1) % gather_every == gather_phase)
        assert len(times) == min(expected, gatherer.KEEP), (len(times), expected)
        tmax = torch.tensor([max(times) if times else 0.0], device=dev if torch.distributed.get_backend() == 'nccl' else 'cpu', dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        rccl = {'backend': torch.distributed.get_backend(), 'world_size': torch.distributed.get_world_size(),
                'ranks_seen': sorted(r for r, _, _ in ranks_seen), 'devices': [f'{r}:cuda:{lr}:{nm}' for r, lr, nm in sorted(ranks_seen)],
                'gather_count': len(times), 'gather_every': gather_every, 'gather_bytes_per_rank': 4 * n,
                'gather_us_max': float(tmax.item()), 'gather_us_median_rank0': float(sorted(times)[len(times) // 2]) if times else None,
                'collective': 'all_gather_into_tensor of the episode returns on a side stream, off the step path'}
    # solver rows per env: the regime the launch time belongs to, in the line itself.  Counted in an UNTIMED pass over the launches that follow the window
    # (the counter is a per-env read-modify-write the shipped configuration does not do, so it is not bound inside the window; the regime is the settled
    # one either way: an env's row count persists from step to step).  No collective in it.
    rows_steps = max(1, min(args.steps, 200))
    for _, env, _, _ in envs:
        env.count_solver_rows()
    t_rows = args.settle + args.warmup + args.steps
    for t in range(t_rows, t_rows + rows_steps):
        for _, env, acts, stream in envs:
            with torch.cuda.stream(stream):
                env.step(acts[t % T])
    torch.cuda.synchronize(dev)
    if world > 1:  # all collectives are done: leave the group before rank 0 spends ~25 s of host time on the CPU baseline
        wd.at('final barrier')
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    wd.at('report')

    if rank == 0:
        total_steps = world * n * args.steps
        # dominant kernel: the ant step (for mixed: k_step<gather> over its half of the shard, overlapped with the point kernel)
        dom = 0
        dom_kind, dom_env = envs[dom][0], envs[dom][1]
        launch_s = dev_ms[dom] / 1e3 / (args.steps - first)  # one kernel per step per stream: HIP-event time on that stream / the launches between the events
        if args.kind == 'gather' and n == 4096:
            metric = 'env-steps/sec, AntGatherBulletEnv-v0 @4096 envs, 1/2/4/8 MI355X'
        elif args.kind == 'mixed':
            metric = f'env-steps/sec, AntGatherBulletEnv-v0 + PointGatherBulletEnv-v0 mixed batch @{n} envs/GPU'
        else:
            metric = f'env-steps/sec, {NAMES[args.kind]} @{n} envs/GPU'
        shard = f'{NAMES[args.kind]}, {n} envs per GPU'
        if args.kind == 'mixed':
            shard += f' ({n // 2} AntGather + {n // 2} PointGather, two launches per step on two HIP streams)'
        out = {
            'metric': metric,
            'value': total_steps / wall, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': wall / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            # True: at least 200 untimed steps (settle + warmup) lie between the reset and the timed window -- the ants stand, the contact count and
            # with it the launch time have levelled off (launches are ~5 % shorter in the first 50 steps after a reset)
            'steady_state': args.settle + args.warmup >= 200, 'settle_steps': args.settle,
            # mean constraint rows per env and env step (joint limits + 3 per contact, summed over the 4 substeps) over the `solver_rows_steps` untimed launches
            # right after the timed window: the sweeps are serial in the rows, so this is the regime the launch time belongs to (hrl_buffers.solver_rows)
            'solver_rows_per_env_step': {k: float(env.solver_rows.sum().item()) / (env.num_envs * rows_steps) for k, env, _, _ in envs},
            # the env with the most rows: a launch lasts as long as its slowest env's chain (one env of 4096 that lies against a wall with 12
            # contacts, 23 rows per substep instead of 20, makes every launch 3 - 4 us longer)
            'solver_rows_max_env': {k: float(env.solver_rows.max().item()) / rows_steps for k, env, _, _ in envs},
            'solver_rows_steps': rows_steps,
            'config': {'workload': f'{shard}, U(-1,1) actions pre-generated on device, auto-reset, max_episode_steps 2000',
                       'envs_per_gpu': n, 'global_envs': world * n,
                       'substeps_per_step': 4,
                       **({'model_overrides': list(args.model)} if args.model else {}),
                       'parallelism': 'one GPU, one process, no collective' if world == 1 else
                                      f'env-sharded x{world} (one process per GPU), no data-path collective; all-gather of episode returns '
                                      f'every {gather_every} steps on a side stream (see `rccl`)'},
            'roofline': roofline(dom_kind, dom_env.num_envs, launch_s),
        }
        if world > 1:   # the slowest rank's own K steps, before the barrier that closes the timed bracket (that barrier is inside `value` / `ms_per_step`)
            out['ms_per_step_before_barrier'] = wall_own / args.steps * 1e3
        if args.kind == 'mixed':
            out['roofline']['streams_ms_per_step'] = {k: ms / (args.steps - first) for (k, _, _, _), ms in zip(envs, dev_ms)}
        if gathered_ok is not None:
            out['config']['returns_gathered_ok'] = gathered_ok
        if rccl is not None:
            out['rccl'] = rccl
        if not args.no_cpu_baseline and world == 1:  # the CPU leg belongs to the N = 1 line only
            out['cpu_baseline'] = cpu_baseline(args.kind)
        print(json.dumps(out), flush=True)
    wd.done()


if __name__ == '__main__':
    main()
