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


def gather_over_ranks(value, dist, device='cpu'):
    """One float per rank -> the list of all ranks' values, in rank order (a timing diagnostic: launch skew between the ranks)."""
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist is None:
        return [float(t.item())]
    parts = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return [float(p.item()) for p in parts]


def live_group(dist):
    """What the LIVE process group says about itself (not what the command line asked for): world size and backend string;
    (1, None) without a group."""
    if dist is None or not dist.is_initialized():
        return 1, None
    return int(dist.get_world_size()), str(dist.get_backend())


# ---------------------------------------------------------------------------------------------------
# SingleCombat: partition by ENV (both aircraft of an engagement stay on one rank, so physics, reward and
# terminations need no exchange).  The one real exchange of the self-play setup is between the rank that
# hosts an engagement and the rank that hosts its OPPONENT POLICY (runner/selfplay_F16sim_runner.py:49-55,
# 90-100): the opponent half of the observations is all-gathered, the opponent policies act on their slice
# of envs, and the opponent actions come back the same way.  Payload at 1e5 engagements: 15 floats x 4 B x
# 1e5 = 6 MB in total per step — latency-bound on xGMI, one collective per direction, no per-env messages.
# ---------------------------------------------------------------------------------------------------
def split_ego_opponent(x, num_envs, num_agents=2):
    """[n, k] rows (2k = ego, 2k+1 = enemy) -> (ego [E, k], opponent [E, k]) views; obs[:, :A//2] / obs[:, A//2:] of the runner."""
    v = x.reshape(num_envs, num_agents, -1)
    return v[:, 0], v[:, 1]


def all_gather_opponent(x_local, dist, envs_per_rank=None):
    """All-gather a per-env tensor [E_local, k] (opponent observations, or opponent actions on the way back)
    over the ranks -> [E_total, k] in global env order.  Equal shards use ONE all_gather_into_tensor; ragged
    shards (E_total % world != 0) fall back to all_gather with padding."""
    x_local = x_local.contiguous()
    if dist is None:
        return x_local
    if x_local.is_cuda and dist.get_backend() == 'gloo':
        # functional-check path only (ranks sharing one GPU in the tests): gloo has no all_gather for device tensors
        return all_gather_opponent(x_local.cpu(), dist, envs_per_rank).to(x_local.device)
    world = dist.get_world_size()
    if envs_per_rank is None:
        sizes = torch.tensor([x_local.shape[0]], dtype=torch.int64, device=x_local.device)
        allsz = [torch.zeros_like(sizes) for _ in range(world)]
        dist.all_gather(allsz, sizes)
        envs_per_rank = [int(t.item()) for t in allsz]
    if len(set(envs_per_rank)) == 1:
        out = torch.empty((world * x_local.shape[0],) + tuple(x_local.shape[1:]), dtype=x_local.dtype, device=x_local.device)
        dist.all_gather_into_tensor(out, x_local)
        return out
    cap = max(envs_per_rank)
    pad = torch.zeros((cap,) + tuple(x_local.shape[1:]), dtype=x_local.dtype, device=x_local.device)
    pad[:x_local.shape[0]] = x_local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:m] for p, m in zip(parts, envs_per_rank)], dim=0)


def merge_actions(ego_actions, opponent_actions):
    """(ego [E, k], opponent [E, k]) -> the [2E, k] action rows the env consumes (np.concatenate(..., axis=1) of the runner)."""
    return torch.stack((ego_actions, opponent_actions), dim=1).reshape(2 * ego_actions.shape[0], -1)
