"""Multi-GPU: one process per GPU, envs sharded by contiguous global id, no data-path collective.

Envs never interact (one Bullet world per env object in the reference, sizeable_enclosed_scene.py:23), so the step
needs no exchange.  RNG streams are keyed by GLOBAL env id (hrl_config.env_id_offset), which makes results
independent of the number of ranks.  The only collective is an all-gather of per-env episode returns
(RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests), issued on a side stream off the step path.
"""
import collections
import os

import torch
import torch.distributed as dist


def shard_range(total_envs, rank, world):
    """Contiguous shard of `total_envs` for `rank`: (offset, count); the first `total % world` ranks get one extra.
    (Uneven shards are fine for stepping; `all_gather_returns` pads them to the largest shard.)"""
    base, extra = divmod(total_envs, world)
    count = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return offset, count


def init_distributed(expected_world=None, backend=None, force=False):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run).  Returns (rank, world, local_rank).
    A world of one needs no process group; `force` creates it anyway (a one-rank RCCL communicator: what the `-m gpu`
    test of the collective path runs on a one-GPU box)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if expected_world is not None and expected_world != world:
        raise RuntimeError(f'--gpus {expected_world} but WORLD_SIZE={world}: launch with '
                           f'python -m torch.distributed.run --nproc-per-node {expected_world} ... '
                           f'(bench.py spawns the ranks itself when WORLD_SIZE is unset)')
    if (world > 1 or force) and not dist.is_initialized():
        # dmabuf IPC between the ranks' processes (RCCL over xGMI needs it on this driver); only effective if nothing has touched the GPU yet,
        # which holds for bench.py and for anything that calls this first.  A launcher's own export wins.
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank


def all_gather_returns(local, world, out=None, counts=None):
    """local: [N_local] float32 -> all ranks' values concatenated in rank order.

    `all_gather_into_tensor` needs equal sizes on every rank.  `counts` (the per-rank shard sizes, e.g. from
    `shard_range`) makes uneven shards legal: every rank pads to max(counts), the padding is trimmed after the
    collective.  Without `counts` the shards must be equal, which is checked by the shape of `out` only -- so callers
    with uneven shards MUST pass counts (an unequal all_gather_into_tensor hangs on RCCL)."""
    if not dist.is_initialized():
        if world != 1:
            raise ValueError(f'all_gather_returns(world={world}) without a process group: call init_distributed() first')
        return local.clone() if out is None else out.copy_(local)
    if dist.get_world_size() != world:  # a collective sized for another world errors out -- or, on RCCL, hangs when only some ranks call it
        raise ValueError(f'all_gather_returns(world={world}) but the process group has {dist.get_world_size()} ranks')
    if counts is not None and len(set(counts)) > 1:
        if len(counts) != world or local.numel() != counts[dist.get_rank()]:
            raise ValueError(f'counts {counts} do not describe this rank\'s shard of {local.numel()} values')
        m = max(counts)
        padded = torch.zeros(m, dtype=local.dtype, device=local.device)
        padded[:local.numel()] = local
        full = all_gather_returns(padded, world)
        res = torch.cat([full[r * m:r * m + c] for r, c in enumerate(counts)])
        return res if out is None else out.copy_(res)
    if out is None:
        out = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)
    elif out.numel() != world * local.numel():
        raise ValueError(f'out has {out.numel()} elements, expected world * N_local = {world * local.numel()}')
    if local.is_cuda and dist.get_backend() != 'nccl':
        # rehearsal path (gloo with device tensors): stage through the host; RCCL gathers device-to-device
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, local.detach().cpu().contiguous())
        out.copy_(host)
        return out
    dist.all_gather_into_tensor(out, local.contiguous())
    return out


class ReturnGatherer:
    """All-gather of the running episode returns (info[:, 2]) every K steps, off the step path.

    16 KiB per rank at 4096 envs: latency-bound, so only the collective runs on a side stream; `latest()` waits for it.
    The snapshot of info[:, 2] is taken ON THE STEP STREAM of each env (a 16 KiB device copy, ordered before the next
    hrl_step that overwrites `info`), then the side stream waits for that snapshot: the gathered values always belong
    to one step.  Snapshot and result buffers are double-buffered: a launch only waits for the collective of TWO
    launches ago (long done), so the step streams never stall behind a collective in flight.
    `envs` is one BatchedEnv or a list of (env, stream) pairs (mixed shards step on their own streams).
    `counts` = every rank's shard size (e.g. from `shard_range`); None = equal shards.  Uneven shards without `counts`
    would make `all_gather_into_tensor` hang on RCCL, so the sizes are exchanged once here when a group exists."""

    KEEP = 4096   # collectives whose timing events are kept

    def __init__(self, envs, world, counts=None):
        if not isinstance(envs, (list, tuple)):
            envs = [(envs, None)]
        self.parts, self.world = list(envs), world
        if dist.is_initialized() and dist.get_world_size() != world:
            raise ValueError(f'ReturnGatherer(world={world}) but the process group has {dist.get_world_size()} ranks')
        if not dist.is_initialized() and world != 1:
            raise ValueError(f'ReturnGatherer(world={world}) without a process group: call init_distributed() first')
        dev = self.parts[0][0].device
        n = sum(e.num_envs for e, _ in self.parts)
        if counts is None and dist.is_initialized():
            sizes = [None] * world
            dist.all_gather_object(sizes, n)
            counts = [int(c) for c in sizes]
        if counts is not None:
            counts = [int(c) for c in counts]
            if len(counts) != world or (dist.is_initialized() and counts[dist.get_rank()] != n):
                raise ValueError(f'counts {counts} do not describe a world of {world} with {n} envs on this rank')
        self.counts = counts
        total = sum(counts) if counts is not None else world * n
        self.cpu = torch.device(dev).type == 'cpu'   # host tensors (the gloo tests on a box without a GPU): no streams, the collective runs in place
        self.stream = None if self.cpu else torch.cuda.Stream(device=dev)
        self.local = [torch.empty(n, dtype=torch.float32, device=dev) for _ in range(2)]
        self.out = [torch.empty(total, dtype=torch.float32, device=dev) for _ in range(2)]
        self.done_event = [None, None]
        self.k = 0          # launches so far
        self.last = None    # buffer index of the latest launch
        self._timed = collections.deque(maxlen=self.KEEP)  # (launch index, start, end) HIP events around the latest collectives on the side stream: `gather_times_us()`

    def launch(self):
        b = self.k & 1
        self.k += 1
        if self.cpu:
            import time
            off = 0
            for env, _ in self.parts:
                self.local[b][off:off + env.num_envs].copy_(env.info[:, 2])
                off += env.num_envs
            t0 = time.perf_counter()
            all_gather_returns(self.local[b], self.world, self.out[b], counts=self.counts)
            self._timed.append((self.k - 1, t0, time.perf_counter()))
            self.last = b
            return
        if self.done_event[b] is not None:  # the collective of two launches ago read self.local[b]: order the new snapshot after it
            for env, st in self.parts:
                (st or torch.cuda.current_stream(env.device)).wait_event(self.done_event[b])
        off, ready = 0, []
        for env, st in self.parts:
            st = st or torch.cuda.current_stream(env.device)
            with torch.cuda.stream(st):
                self.local[b][off:off + env.num_envs].copy_(env.info[:, 2])
                ev = torch.cuda.Event()
                ev.record(st)
            ready.append(ev)
            off += env.num_envs
        with torch.cuda.stream(self.stream):
            for ev in ready:
                self.stream.wait_event(ev)
            t0 = torch.cuda.Event(enable_timing=True)
            t0.record(self.stream)
            all_gather_returns(self.local[b], self.world, self.out[b], counts=self.counts)
            self.done_event[b] = torch.cuda.Event(enable_timing=True)
            self.done_event[b].record(self.stream)
            self._timed.append((self.k - 1, t0, self.done_event[b]))
        self.last = b

    def gather_times_us(self, first=0):
        """Duration in microseconds of the collectives with launch index >= `first` (0-based; of the latest KEEP launches -- older ones are
        forgotten, which an index, unlike a slice of this list, does not get wrong), from HIP events on the side stream (waits for those)."""
        out = []
        for k, t0, t1 in self._timed:
            if k < first:
                continue
            if self.cpu:
                out.append((t1 - t0) * 1e6)
                continue
            t1.synchronize()
            out.append(t0.elapsed_time(t1) * 1e3)
        return out

    def latest(self):
        """The result of the most recent launch (waits for it); None before the first launch."""
        if self.last is None:
            return None
        if not self.cpu:
            self.done_event[self.last].synchronize()
        return self.out[self.last]
