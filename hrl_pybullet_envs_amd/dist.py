"""Multi-GPU: one process per GPU, envs sharded by contiguous global id, no data-path collective.

Envs never interact (one Bullet world per env object in the reference, sizeable_enclosed_scene.py:23), so the step
needs no exchange.  RNG streams are keyed by GLOBAL env id (hrl_config.env_id_offset), which makes results
independent of the number of ranks.  The only collective is an all-gather of per-env episode returns
(RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests), issued on a side stream off the step path.
"""
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


def init_distributed(expected_world=None, backend=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if expected_world is not None and expected_world != world:
        raise RuntimeError(f'--gpus {expected_world} but WORLD_SIZE={world}: launch with '
                           f'python -m torch.distributed.run --nproc-per-node {expected_world} ... '
                           f'(bench.py spawns the ranks itself when WORLD_SIZE is unset)')
    if world > 1 and not dist.is_initialized():
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
    if world == 1:
        return local.clone() if out is None else out.copy_(local)
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
    to one step.  `envs` is one BatchedEnv or a list of (env, stream) pairs (mixed shards step on their own streams)."""

    def __init__(self, envs, world):
        if not isinstance(envs, (list, tuple)):
            envs = [(envs, None)]
        self.parts, self.world = list(envs), world
        dev = self.parts[0][0].device
        n = sum(e.num_envs for e, _ in self.parts)
        self.stream = torch.cuda.Stream(device=dev)
        self.local = torch.empty(n, dtype=torch.float32, device=dev)
        self.out = torch.empty(world * n, dtype=torch.float32, device=dev)
        self.done_event = None

    def launch(self):
        if self.done_event is not None:  # the previous collective still reads self.local: order the new snapshot after it
            for env, st in self.parts:
                (st or torch.cuda.current_stream(env.device)).wait_event(self.done_event)
        off, ready = 0, []
        for env, st in self.parts:
            st = st or torch.cuda.current_stream(env.device)
            with torch.cuda.stream(st):
                self.local[off:off + env.num_envs].copy_(env.info[:, 2])
                ev = torch.cuda.Event()
                ev.record(st)
            ready.append(ev)
            off += env.num_envs
        with torch.cuda.stream(self.stream):
            for ev in ready:
                self.stream.wait_event(ev)
            all_gather_returns(self.local, self.world, self.out)
            self.done_event = torch.cuda.Event()
            self.done_event.record(self.stream)

    def latest(self):
        if self.done_event is not None:
            self.done_event.synchronize()
        return self.out
