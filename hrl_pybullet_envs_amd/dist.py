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
    """Contiguous shard of `total_envs` for `rank`: (offset, count); the first `total % world` ranks get one extra."""
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
                           f'python -m torch.distributed.run --nproc-per-node {expected_world} ...')
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


def all_gather_returns(local, world, out=None):
    """local: [N_local] float32 -> [world * N_local] in rank order (equal shard sizes)."""
    if world == 1:
        return local.clone() if out is None else out.copy_(local)
    if out is None:
        out = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)
    if local.is_cuda and dist.get_backend() != 'nccl':
        # rehearsal path (gloo with device tensors): stage through the host; RCCL gathers device-to-device
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, local.detach().cpu().contiguous())
        out.copy_(host)
        return out
    dist.all_gather_into_tensor(out, local.contiguous())
    return out


class ReturnGatherer:
    """All-gather of the running episode returns (info[:, 2]) on a side stream, every K steps.

    16 KiB per rank at 4096 envs: latency-bound, so it is kept off the step stream; `latest()` waits for it."""

    def __init__(self, env, world):
        self.env, self.world = env, world
        self.stream = torch.cuda.Stream(device=env.device)
        self.local = torch.empty(env.num_envs, dtype=torch.float32, device=env.device)
        self.out = torch.empty(world * env.num_envs, dtype=torch.float32, device=env.device)
        self.done_event = None

    def launch(self):
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.env.device))
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            self.local.copy_(self.env.info[:, 2])
            all_gather_returns(self.local, self.world, self.out)
            self.done_event = torch.cuda.Event()
            self.done_event.record(self.stream)

    def latest(self):
        if self.done_event is not None:
            self.done_event.synchronize()
        return self.out
