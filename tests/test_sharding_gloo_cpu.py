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
import pytest
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


# ---------------------------------------------------------------------------------------------------
# SingleCombat: shard by env + the opponent-observation exchange of the self-play setup
# ---------------------------------------------------------------------------------------------------
def _combat_worker(rank, world, port, e_total, steps, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port), OMP_NUM_THREADS='1')
    from neuralplane_amd import sharding
    from oracle.f16_oracle import CombatOracle
    d = sharding.init_distributed('gloo')
    env0, e_loc = sharding.shard_rows(e_total, world, rank)
    o = CombatOracle()
    st = o.new_state(e_loc)
    obs = torch.from_numpy(o.combat_reset(st, seed=21, call_idx=0, env0=env0))
    W = torch.linspace(-1, 1, 15 * 4).reshape(15, 4)      # a fixed linear "policy" for both sides
    for t in range(steps):
        ego_obs, opp_obs = sharding.split_ego_opponent(obs, e_loc)
        # opponent policy lives on rank (world - 1): every rank contributes its opponent observations, the
        # host of the opponent policy acts on ALL of them, and the actions travel back the same way
        opp_all = sharding.all_gather_opponent(opp_obs, d)
        assert opp_all.shape == (e_total, 15)
        act_all = torch.tanh(opp_all @ W) if rank == world - 1 else torch.zeros(e_total, 4)
        d.broadcast(act_all, src=world - 1)
        opp_act = act_all[env0:env0 + e_loc]
        ego_act = torch.tanh(ego_obs @ W)
        a = sharding.merge_actions(ego_act, opp_act)
        o_np, rew, dn, bd, tm = o.combat_step(st, a.numpy(), pid_first=(t == 0), seed=21, call_idx=t + 1, env0=env0)
        obs = torch.from_numpy(o_np)
    parts = [None] * world
    d.all_gather_object(parts, (env0, st['s'], o_np, rew, st['blood']))
    if rank == 0:
        q.put(parts)
    d.barrier()
    d.destroy_process_group()


@pytest.mark.parametrize('e_total', [21, 20], ids=['ragged_padded_all_gather', 'equal_all_gather_into_tensor'])
def test_two_rank_combat_shards_with_opponent_exchange_equal_single_process_run(e_total):
    steps, world = 6, 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_combat_worker, args=(r, world, port, e_total, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    parts = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    parts.sort(key=lambda x: x[0])
    s, obs, rew, blood = (np.concatenate([p[k] for p in parts]) for k in (1, 2, 3, 4))

    sys.path.insert(0, ROOT)
    from neuralplane_amd import sharding
    from oracle.f16_oracle import CombatOracle
    o = CombatOracle()
    st = o.new_state(e_total)
    ob = torch.from_numpy(o.combat_reset(st, seed=21, call_idx=0, env0=0))
    W = torch.linspace(-1, 1, 15 * 4).reshape(15, 4)
    for t in range(steps):
        ego_obs, opp_obs = sharding.split_ego_opponent(ob, e_total)
        a = sharding.merge_actions(torch.tanh(ego_obs @ W), torch.tanh(opp_obs @ W))
        o_np, o_rew, _, _, _ = o.combat_step(st, a.numpy(), pid_first=(t == 0), seed=21, call_idx=t + 1, env0=0)
        ob = torch.from_numpy(o_np)
    assert np.array_equal(s, st['s']) and np.array_equal(obs, o_np) and np.array_equal(rew, o_rew) and np.array_equal(blood, st['blood'])


# ---------------------------------------------------------------------------------------------------
# The self-play loop bench.py --task combat steps (neuralplane_amd/selfplay.py::OpponentExchange): opponent
# observations all-gathered to the rank hosting the opponent policy, opponent actions all-gathered back
# ---------------------------------------------------------------------------------------------------
def _selfplay_loop(o, st, e_loc, env0, e_total, d, steps, lag, split=False):
    from neuralplane_amd import sharding
    from neuralplane_amd.selfplay import OpponentExchange
    W_ego = torch.linspace(-1, 1, 15 * 4).reshape(15, 4)
    W_opp = torch.linspace(1, -1, 15 * 4).reshape(15, 4)

    def opp_policy(obs, env_ids):     # depends on the GLOBAL env index: a misrouted slice cannot go unnoticed
        return torch.tanh(obs @ W_opp + (env_ids.to(obs.dtype) % 7)[:, None] * 0.01)

    ex = OpponentExchange(e_loc, env0, e_total, d, 'cpu', opponent_policy=opp_policy, lag=lag)
    obs = torch.from_numpy(o.combat_reset(st, seed=21, call_idx=0, env0=env0))
    for t in range(steps):
        if split and d is not None:
            # the split layout of SingleCombatEnv.step_split (the two halves as separate contiguous arrays, nothing copied by the
            # exchange); the oracle engine takes interleaved rows, so the halves are cut and joined here, outside the exchange
            oe, oo = (h.contiguous() for h in sharding.split_ego_opponent(obs, e_loc))
            a = sharding.merge_actions(*ex.actions_split(oe, oo, lambda x: torch.tanh(x @ W_ego)))
        else:
            a = ex.actions(obs, lambda x: torch.tanh(x @ W_ego))
        o_np, rew, dn, bd, tm = o.combat_step(st, a.numpy(), pid_first=(t == 0), seed=21, call_idx=t + 1, env0=env0)
        obs = torch.from_numpy(o_np)
    return o_np, rew


def _selfplay_worker(rank, world, port, e_total, steps, lag, q, split=False):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port), OMP_NUM_THREADS='1')
    from neuralplane_amd import sharding
    from oracle.f16_oracle import CombatOracle
    d = sharding.init_distributed('gloo')
    env0, e_loc = sharding.shard_rows(e_total, world, rank)
    o = CombatOracle()
    st = o.new_state(e_loc)
    o_np, rew = _selfplay_loop(o, st, e_loc, env0, e_total, d, steps, lag, split)
    parts = [None] * world
    d.all_gather_object(parts, (env0, st['s'], o_np, rew, st['blood']))
    if rank == 0:
        q.put(parts)
    d.barrier()
    d.destroy_process_group()


@pytest.mark.parametrize('e_total,lag,split', [(21, 0, False), (20, 0, False), (20, 1, False), (21, 1, False), (20, 0, True), (21, 1, True)],
                         ids=['ragged_lag0', 'equal_lag0', 'equal_lag1', 'ragged_lag1', 'equal_lag0_split', 'ragged_lag1_split'])
def test_two_rank_selfplay_exchange_equals_single_process_run(e_total, lag, split):
    steps, world = 6, 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_selfplay_worker, args=(r, world, port, e_total, steps, lag, q, split)) for r in range(world)]
    for p in procs:
        p.start()
    parts = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    parts.sort(key=lambda x: x[0])
    s, obs, rew, blood = (np.concatenate([p[k] for p in parts]) for k in (1, 2, 3, 4))

    sys.path.insert(0, ROOT)
    from oracle.f16_oracle import CombatOracle
    o = CombatOracle()
    st = o.new_state(e_total)
    o_np, o_rew = _selfplay_loop(o, st, e_total, 0, e_total, None, steps, lag)
    assert np.array_equal(s, st['s']) and np.array_equal(obs, o_np) and np.array_equal(rew, o_rew) and np.array_equal(blood, st['blood'])
