"""Row sharding of an aircraft batch over the GPUs of one node (one process per GPU).

Aircraft never interact in the Control/Heading/Tracking tasks, so the batch is split by rows with
NO data-path collective; torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo"
in the CPU tests) is used only for the timing barrier and the max-over-ranks reduction.  Because
the kernels key their counter-based RNG by the GLOBAL row index (`row0 + i`), a sharded run
reproduces the unsharded trajectories bit for bit.
"""
import os

import torch


def shard_rows(n_total, world_size, rank):
    """Contiguous block partition: returns (row0, n_local); the first `n_total % world_size` ranks get one extra row."""
    base, rem = divmod(int(n_total), int(world_size))
    n_local = base + (1 if rank < rem else 0)
    row0 = rank * base + min(rank, rem)
    return row0, n_local


def env_world():
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')))


def init_distributed(backend, device=None):
    """Initialise torch.distributed from the torchrun environment (no-op for WORLD_SIZE=1)."""
    import torch.distributed as dist
    rank, _, world = env_world()
    if world == 1:
        return None
    if not dist.is_initialized():
        kw = {}
        if backend == 'nccl' and device is not None:
            kw['device_id'] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def max_over_ranks(value, dist, device='cpu'):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
