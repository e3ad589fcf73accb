"""world_size-2 `gloo` test of the multi-GPU path's host logic on CPU.

The N>1 path has no data-path collective: rows are split with sharding.shard_rows, every rank
steps its own shard, and torch.distributed only carries the barrier + max-over-ranks of bench.py.
What must hold by construction is that a sharded run reproduces the unsharded one bit for bit —
the RNG is keyed by the GLOBAL row — which is checked here with the CPU oracle standing in for
the per-rank engine (the GPU tests check HIP == oracle with a non-zero row0)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, steps, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port), OMP_NUM_THREADS='1')
    from neuralplane_amd import sharding
    from oracle.f16_oracle import Oracle
    d = sharding.init_distributed('gloo')
    assert d is not None and sharding.env_world() == (rank, rank, world)
    row0, n = sharding.shard_rows(n_total, world, rank)
    o = Oracle('heading')
    st = Oracle.new_state(n)
    acts = np.random.RandomState(3).uniform(-1, 1, (steps, n_total, 4)).astype(np.float32)
    o.reset(st, seed=11, call_idx=0, row0=row0)
    d.barrier()
    for t in range(steps):
        obs, rew, dn, bd, tm = o.step(st, acts[t, row0:row0 + n], seed=11, call_idx=t + 1, row0=row0)
    d.barrier()
    worst = sharding.max_over_ranks(float(rank + 1), d)      # the reduction bench.py uses for the step time
    assert worst == float(world)
    parts = [None] * world
    d.all_gather_object(parts, (row0, st['s'], obs, rew, bd))
    if rank == 0:
        q.put(parts)
    d.barrier()
    d.destroy_process_group()


def test_two_rank_sharded_run_equals_unsharded_run():
    n_total, steps, world = 101, 12, 2          # odd size: ragged shards
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    parts = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    parts.sort(key=lambda x: x[0])
    s = np.concatenate([p[1] for p in parts])
    obs = np.concatenate([p[2] for p in parts])
    rew = np.concatenate([p[3] for p in parts])
    bad = np.concatenate([p[4] for p in parts])

    sys.path.insert(0, ROOT)
    from oracle.f16_oracle import Oracle
    o = Oracle('heading')
    st = Oracle.new_state(n_total)
    acts = np.random.RandomState(3).uniform(-1, 1, (steps, n_total, 4)).astype(np.float32)
    o.reset(st, seed=11, call_idx=0, row0=0)
    for t in range(steps):
        o_obs, o_rew, _, o_bad, _ = o.step(st, acts[t], seed=11, call_idx=t + 1, row0=0)
    assert np.array_equal(s, st['s']) and np.array_equal(obs, o_obs) and np.array_equal(rew, o_rew) and np.array_equal(bad, o_bad)
