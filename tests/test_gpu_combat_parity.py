"""GPU parity of the fused SingleCombat macro-step (np_f16_combat_step, one launch per env.step) against the CPU
oracle and the fixtures recorded from the reference's components.  All calls go through the C ABI
(neuralplane_amd.core.F16CombatBatch -> libneuralplane_hip.so).

Bar: HIP == oracle BIT-EXACT for states, controls, controller state, blood, counters, the three masks,
observations and rewards — free-running, because the closed attitude loop amplifies any last-bit difference
(tests/test_combat_oracle_golden.py explains) — and HIP vs the reference's plain recording per env.step within
the tolerances of the CPU test.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.f16_oracle import MODE_PWL, CombatOracle  # noqa: E402  (the checker; test infrastructure)

STATE_FLOORS = np.array([100, 100, 100, .1, .1, .1, 10, .1, .1, .1, .1, .1], np.float32)


def _batch(num_envs, solver=None, seed=0, env0=0, tables=False, variant='auto'):
    from neuralplane_amd.core import F16CombatBatch
    from neuralplane_amd.envs.utils.utils import parse_config
    b = F16CombatBatch(num_envs, parse_config('selfplay'), 'cuda:0', seed=seed, solver=solver, env0=env0, aero_1d_tables=tables)
    b.set_kernel_variant(variant)   # 'auto' picks the four-wave latency kernel at these sizes; 'dual8' / 'dual4': np_combat_lat.hip
    return b


def _load(b, st):
    b.s.copy_(torch.from_numpy(st['s'].T.copy()))
    b.u.copy_(torch.from_numpy(st['u'].T.copy()))
    b.pid.copy_(torch.from_numpy(st['pid'].T.copy()))
    b.blood.copy_(torch.from_numpy(st['blood']))
    b.step_count.copy_(torch.from_numpy(st['step_count']))
    b.flags.copy_(torch.from_numpy(np.stack([st['done'], st['bad'], st['timeout']])))


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return bool(np.all((a == b) | (np.isnan(a.astype(np.float64)) & np.isnan(b.astype(np.float64)))))


def _check(b, obs, rew, flags, st, o_obs, o_rew, what, rows=slice(None)):
    assert _same(b.s.cpu().numpy().T[rows], st['s']), f'{what}: state'
    assert _same(b.u.cpu().numpy().T[rows], st['u']), f'{what}: controls'
    assert _same(b.pid.cpu().numpy().T[rows], st['pid']), f'{what}: controller state'
    assert _same(b.blood.cpu().numpy()[rows], st['blood']), f'{what}: blood'
    assert np.array_equal(b.step_count.cpu().numpy()[rows], st['step_count']), f'{what}: step_count'
    f = flags.cpu().numpy()
    assert np.array_equal(f[0][rows], st['done']) and np.array_equal(f[1][rows], st['bad']) and \
        np.array_equal(f[2][rows], st['timeout']), f'{what}: masks'
    assert _same(obs.cpu().numpy()[rows], o_obs), f'{what}: obs'
    if rew is not None:
        assert _same(rew.cpu().numpy()[rows], o_rew), f'{what}: reward'


def _fixture_state(o, d, n):
    st = o.new_state(n // 2)
    st['s'][:], st['u'][:], st['blood'][:], st['step_count'][:] = d['s_init'], d['u_init'], d['blood_init'], d['step_count_init']
    st['done'][:] = 0
    st['bad'][:] = 0
    st['timeout'][:] = 0
    return st


@pytest.mark.parametrize('variant', ['latency', 'throughput', 'pair', 'dual8', 'dual4'])
@pytest.mark.parametrize('tables', [False, True], ids=['mlp', 'aero_1d_tables'])
def test_combat_fixture_free_running_bit_exact_vs_oracle(golden_dir, tables, variant):
    if tables and variant in ('pair', 'dual8', 'dual4'):
        pytest.skip('the two-set variants evaluate the MLP numerics only (the table mode falls back to the single-set kernels)')
    """The 48 recorded env.steps (Crash, Timeout, both Shutdown outcomes, pairwise auto-resets with injected draws)."""
    d = np.load(f'{golden_dir}/combat_kat.npz')
    K, n = d['actions'].shape[:2]
    o = CombatOracle(mode=MODE_PWL if tables else 0)
    st = _fixture_state(o, d, n)
    b = _batch(n // 2, tables=tables, variant=variant)
    _load(b, st)
    fired = np.zeros(3, np.int64)
    for k in range(K):
        obs, rew, flags = b.step(torch.from_numpy(d['actions'][k]).cuda(), rand_u=d['rand_u'][k])
        o_obs, o_rew, dn, bd, tm = o.combat_step(st, d['actions'][k], rand_u=d['rand_u'][k], pid_first=(k == 0))
        _check(b, obs, rew, flags, st, o_obs, o_rew, f'fixture step {k}')
        fired += np.array([int(dn.sum()), int(bd.sum()), int(tm.sum())])
    assert fired[0] >= 2 and fired[1] >= 4 and fired[2] >= 2


def test_combat_fixture_vs_reference_teacher_forced(golden_dir):
    """HIP vs the reference's own (plain ATen) recording, each env.step started from the recorded state."""
    d = np.load(f'{golden_dir}/combat_kat.npz')
    K, n = d['actions'].shape[:2]
    b = _batch(n // 2)
    st = _fixture_state(CombatOracle, d, n)
    for k in range(K):
        _load(b, st)
        b.pid_first = (k == 0)
        obs, rew, flags = b.step(torch.from_numpy(d['actions'][k]).cuda(), rand_u=d['rand_u'][k])
        s = b.s.cpu().numpy().T
        ref = d[f's_{k}']
        assert np.max(np.abs(s - ref) / np.maximum(np.abs(ref), STATE_FLOORS)) < 3e-3, k   # P, Q, R: measured 1.8e-3 (the rate PID amplifies fp32 noise; profiles/r03_parity.json)
        assert np.max(np.abs(s[:, :9] - ref[:, :9]) / np.maximum(np.abs(ref[:, :9]), STATE_FLOORS[:9])) < 1e-4, k
        assert np.array_equal(flags.cpu().numpy(), d[f'flags_{k}']), k
        assert np.allclose(obs.cpu().numpy(), d[f'obs_{k}'], rtol=0, atol=1e-4), k
        assert np.allclose(rew.cpu().numpy(), d[f'reward_{k}'], rtol=0, atol=2e-6), k
        assert np.allclose(b.blood.cpu().numpy(), d[f'blood_{k}'], rtol=0, atol=1e-4), k
        st = dict(s=d[f's_{k}'], u=d[f'u_{k}'], pid=d[f'pid_{k}'], blood=d[f'blood_{k}'], step_count=d[f'step_count_{k}'],
                  done=d[f'flags_{k}'][0], bad=d[f'flags_{k}'][1], timeout=d[f'flags_{k}'][2])


@pytest.mark.parametrize('solver,variant', [('euler', 'latency'), ('euler', 'throughput'), ('euler', 'pair'), ('euler', 'dual8'), ('euler', 'dual4'), ('rk4', 'auto')])
def test_combat_free_running_production_rng_bit_exact_vs_oracle(solver, variant):
    """reset + 40 env.steps (200 FDM steps) with the in-kernel Philox reset draws, hazard-rich demands, a ragged last
    workgroup and a non-zero first env (shard offset)."""
    num_envs, steps, seed, env0 = 333, 40, 77, 3_000_000_000
    n = 2 * num_envs
    b = _batch(num_envs, solver=solver, seed=seed, env0=env0, variant=variant)
    o = CombatOracle(solver=solver)
    st = o.new_state(num_envs)
    rng = np.random.RandomState(8)
    obs = b.reset()
    o_obs = o.combat_reset(st, seed=seed, call_idx=0, env0=env0)   # new_state: every flag set -> every env drawn
    _check(b, obs, None, b.flags, st, o_obs, None, 'reset')
    # a short fuse: some pairs close together / low on blood / near the step limit (same edits on both sides)
    st['s'][11, :3] = st['s'][10, :3] + np.float32([120, 50, -30])
    st['blood'][20:30] = np.float32(0.4)
    st['step_count'][40:44] = 1980
    _load(b, st)
    total = np.zeros(3, np.int64)
    counts = np.zeros(9, np.uint32)
    b.termination_counts(reset=True)
    for t in range(steps):
        a = rng.uniform(-1.4, 1.4, (n, 4)).astype(np.float32)
        a[:, 0] = rng.uniform(0, 1.2, n)
        obs, rew, flags = b.step(torch.from_numpy(a).cuda())
        o_obs, o_rew, dn, bd, tm = o.combat_step(st, a, pid_first=(t == 0), seed=seed, call_idx=t + 1, env0=env0, term_counts=counts)
        _check(b, obs, rew, flags, st, o_obs, o_rew, f'{solver}: step {t}')
        total += np.array([int(dn.sum()), int(bd.sum()), int(tm.sum())])
    assert total[1] > 0 and total[2] > 0
    got = b.termination_counts()
    assert list(got.values()) == counts.tolist() and got['crash'] > 0 and got['timeout'] > 0 and got['shutdown_bad'] > 0


def test_combat_sharding_by_env_is_invariant():
    """Engagements are independent: a shard [e0, e1) of the batch reproduces the same rows bit for bit."""
    E, seed = 700, 5
    rng = np.random.RandomState(2)
    acts = rng.uniform(-1.2, 1.2, (6, 2 * E, 4)).astype(np.float32)
    full = _batch(E, seed=seed)
    full.reset()
    lo, hi = 257, 600   # odd boundaries: pairs must stay together
    part = _batch(hi - lo, seed=seed, env0=lo)
    part.reset()
    for k in range(6):
        of, rf, ff = full.step(torch.from_numpy(acts[k]).cuda())
        op, rp, fp = part.step(torch.from_numpy(acts[k][2 * lo:2 * hi]).cuda())
        assert torch.equal(of[2 * lo:2 * hi], op) and torch.equal(rf[2 * lo:2 * hi], rp) and torch.equal(ff[:, 2 * lo:2 * hi], fp)
    assert torch.equal(full.s[:, 2 * lo:2 * hi], part.s) and torch.equal(full.blood[2 * lo:2 * hi], part.blood)


def test_singlecombat_env_surface_and_vec_wrapper():
    """The reference-shaped surface: constructor, attributes, shapes, GPUVecEnv [E, 2, .] reshape, checkpoint."""
    from neuralplane_amd.envs.env_wrappers import GPUVecEnv
    from neuralplane_amd.envs.singlecombat_env import SingleCombatEnv
    env = SingleCombatEnv(num_envs=6, config='selfplay', random_seed=3, device='cuda:0')
    assert (env.num_agents, env.n, env.num_observation, env.num_actions) == (2, 12, 15, 4)
    assert env.observation_space.shape == (15,) and env.action_space.shape == (4,)
    obs = env.reset()
    assert obs.shape == (12, 15) and obs.device.type == 'cuda' and torch.isfinite(obs).all()
    assert env.s.shape == (12, 12) and env.u.shape == (12, 5) and env.blood.shape == (12,) and bool((env.blood == 100).all())
    # pair symmetry of the observation: relative slots are mirrored between the two aircraft of an env
    o = obs.view(6, 2, 15)
    assert torch.equal(o[:, 0, 9], -o[:, 1, 9]) and torch.equal(o[:, 0, 10], -o[:, 1, 10])
    assert torch.equal(o[:, 0, 13], o[:, 1, 13]) and torch.equal(o[:, 0, 14], -o[:, 1, 14])
    a = torch.zeros(12, 4, device='cuda')
    a[:, 0] = 0.5
    out = env.step(a)
    assert len(out) == 6 and out[0].shape == (12, 15) and out[1].shape == (12,) and all(x.dtype == torch.bool for x in out[2:5])
    assert bool((env.step_count == 5).all())                      # 5 FDM steps per env.step
    sd = env.state_dict()
    ref = [env.step(a) for _ in range(3)]
    env2 = SingleCombatEnv(num_envs=6, config='selfplay', random_seed=3, device='cuda:0')
    env2.load_state_dict(sd)
    for r in ref:
        got = env2.step(a)
        assert all(torch.equal(x, y) for x, y in zip(r[:5], got[:5]))
    with pytest.raises(ValueError):
        env.step(torch.zeros(11, 4, device='cuda'))
    vec = GPUVecEnv([lambda: SingleCombatEnv(num_envs=4, config='selfplay', random_seed=0, device='cuda:0')])
    o0 = vec.reset()
    assert o0.shape == (4, 2, 15) and isinstance(o0, np.ndarray)
    o1, r1, d1, b1, t1, _ = vec.step(np.zeros((4, 2, 4), np.float32))
    assert o1.shape == (4, 2, 15) and r1.shape == (4, 2, 1) and d1.shape == b1.shape == t1.shape == (4, 2, 1) and d1.dtype == np.bool_


@pytest.mark.parametrize('E', [100_000, 25_000, 12_500], ids=['1e5_one_gpu_pair', '25000_share_of_4_gpus_dual4', '12500_share_of_8_gpus_dual8'])
def test_combat_full_size_sampled_blocks_vs_oracle(E):
    """BASELINE.json config 5 size (1e5 engagements = 2e5 aircraft on one GPU, and its shares of four and eight GPUs, which the automatic
    choice gives to the dual4 / dual8 kernels): 3 env.steps at full size, sampled workgroups compared bit for bit with the oracle, plus
    determinism of the whole batch."""
    seed = 11
    from neuralplane_amd import _lib
    assert _lib.dispatch_plan(2 * E, 256)['combat_latency'] == {100_000: 0, 25_000: 3, 12_500: 2}[E]
    n = 2 * E
    g = torch.Generator(device='cpu').manual_seed(4)
    acts = [(torch.rand((n, 4), generator=g) * 2.4 - 1.2) for _ in range(3)]
    runs = []
    for rep in range(2):
        b = _batch(E, seed=seed)
        b.reset()
        outs = [b.step(a.cuda()) for a in acts]
        runs.append((b, outs))
    (b, outs), (b2, outs2) = runs
    for (o1, r1, f1), (o2, r2, f2) in zip(outs, outs2):
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(f1, f2)
    assert torch.isfinite(outs[-1][0]).all() and torch.isfinite(outs[-1][1]).all()
    o = CombatOracle()
    for e0 in (0, 64 * (E // 142) + 13, E - 96):      # first, a middle (odd offset) and the last block of engagements
        cnt = 96
        st = o.new_state(cnt)
        o.combat_reset(st, seed=seed, call_idx=0, env0=e0)
        rows = slice(2 * e0, 2 * (e0 + cnt))
        for k, a in enumerate(acts):
            o_obs, o_rew, _, _, _ = o.combat_step(st, a.numpy()[rows], pid_first=(k == 0), seed=seed, call_idx=k + 1, env0=e0)
        _check(b, outs[-1][0], outs[-1][1], outs[-1][2], st, o_obs, o_rew, f'block at env {e0}', rows=rows)


@pytest.mark.parametrize('variant', ['latency', 'throughput', 'pair', 'dual8', 'dual4'])
def test_combat_hostile_inputs(variant):
    """NaN / inf demands and poisoned states: the per-row "non-finite target or measurement holds the previous PID output" rule,
    NaN-compares-false in Crash / Shutdown / Timeout and the pair exchange must agree with the oracle lane by lane."""
    E, seed = 96, 13
    n = 2 * E
    b = _batch(E, seed=seed, variant=variant)
    o = CombatOracle()
    st = o.new_state(E)
    rng = np.random.RandomState(4)
    b.reset()
    o.combat_reset(st, seed=seed, call_idx=0)
    s = st['s']
    s[0, 6] = 0.0
    s[3, 2] = np.nan                  # NaN altitude of an enemy aircraft: poisons its pair's geometry too
    s[4, 9] = np.inf
    s[7, 3] = 2.0e9
    s[8, 4] = np.float32(np.pi / 2)
    s[11, :3] = s[10, :3]             # co-located pair: R = 0
    s[12, 6] = 1.0e6
    st['blood'][14] = np.nan
    st['blood'][17] = -np.inf
    st['pid'][20:30] = rng.normal(0, 100, (10, 11)).astype(np.float32)
    st['pid'][30, 4] = np.nan         # last_out NaN: the hold path returns NaN
    _load(b, st)
    b.pid_first = False
    for t in range(5):
        a = rng.uniform(-1.5, 1.5, (n, 4)).astype(np.float32)
        a[40, 1] = np.nan
        a[41, 2] = np.inf
        a[42, 0] = -np.inf
        a[43] = [1e30, -1e30, 1e-40, -0.0]
        obs, rew, flags = b.step(torch.from_numpy(a).cuda())
        o_obs, o_rew, _, _, _ = o.combat_step(st, a, pid_first=False, seed=seed, call_idx=t + 1)
        _check(b, obs, rew, flags, st, o_obs, o_rew, f'hostile step {t}')


@pytest.mark.parametrize('variant,num_envs', [('latency', 97), ('pair', 97), ('pair', 640), ('throughput', 33), ('dual8', 97), ('dual4', 640)])
def test_split_layout_equals_interleaved_layout(variant, num_envs):
    """np_f16_combat_io.action_opp / obs_opp: the ego / opponent halves as separate contiguous per-env arrays give exactly the
    interleaved launch's results — observations, rewards, masks, every state array — through resets, a ragged last workgroup and
    caller-owned output buffers; the action halves may be row-strided views (a slice of a gathered buffer)."""
    seed = 9
    a, b = _batch(num_envs, seed=seed, variant=variant), _batch(num_envs, seed=seed, variant=variant)
    rng = np.random.RandomState(4)
    obs = a.reset()
    oe, oo = b.reset_split()
    assert torch.equal(obs.view(num_envs, 2, 15)[:, 0], oe) and torch.equal(obs.view(num_envs, 2, 15)[:, 1], oo)
    out = (torch.empty((num_envs, 15), device='cuda'), torch.empty((num_envs, 15), device='cuda'))
    gathered = torch.zeros((num_envs + 5, 6), device='cuda')     # the opponent actions as a slice of a larger, wider buffer
    for t in range(12):
        act = torch.from_numpy(rng.uniform(-1.3, 1.3, (2 * num_envs, 4)).astype(np.float32)).cuda()
        act[:, 0] = act[:, 0].abs()
        obs, rew, flags = a.step(act)
        pairs = act.view(num_envs, 2, 4)
        gathered[3:3 + num_envs, :4] = pairs[:, 1]
        ego = torch.zeros((num_envs, 6), device='cuda')
        ego[:, :4] = pairs[:, 0]
        oe, oo, rew2, flags2 = b.step_split(ego, gathered[3:3 + num_envs], out=out if t % 2 else None)
        v = obs.view(num_envs, 2, 15)
        assert torch.equal(v[:, 0], oe) and torch.equal(v[:, 1], oo), f'step {t}: observations'
        assert torch.equal(rew, rew2) and torch.equal(flags, flags2), f'step {t}: reward / masks'
        for k in ('s', 'u', 'pid', 'blood', 'step_count'):
            assert torch.equal(getattr(a, k), getattr(b, k)), f'step {t}: {k}'
    assert a.call_idx == b.call_idx


def test_opponent_exchange_runs_over_rccl():
    """sharding.all_gather_opponent on device tensors through the 'nccl' backend (= RCCL on ROCm), world size 1 — the only
    world a 1-GPU box offers: checks that the collective path (communicator creation, all_gather_into_tensor on the env's
    device buffers, no host staging) works on real hardware; the multi-rank logic is covered by the gloo test on CPU."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, sys, torch, torch.distributed as dist\n"
        "sys.path.insert(0, %r)\n"
        "from neuralplane_amd import sharding\n"
        "from neuralplane_amd.envs.singlecombat_env import SingleCombatEnv\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))\n"
        "env = SingleCombatEnv(num_envs=64, config='selfplay', random_seed=0, device='cuda:0')\n"
        "obs = env.reset()\n"
        "ego, opp = sharding.split_ego_opponent(obs, 64)\n"
        "allopp = sharding.all_gather_opponent(opp, dist)\n"
        "assert allopp.is_cuda and allopp.shape == (64, 15) and torch.equal(allopp, opp.contiguous())\n"
        "act = sharding.merge_actions(torch.zeros(64, 4, device='cuda'), torch.ones(64, 4, device='cuda') * 0.1)\n"
        "out = env.step(act)\n"
        "torch.cuda.synchronize(); dist.destroy_process_group(); print('RCCL_OK', tuple(out[0].shape))\n" % root)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and 'RCCL_OK (128, 15)' in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


# ---------------------------------------------------------------------------------------------------
# the sharded self-play loop of bench.py --task combat on the real kernels: two gloo ranks sharing this GPU
# ---------------------------------------------------------------------------------------------------
def _hip_selfplay_loop(e_loc, env0, e_total, d, steps, lag):
    from neuralplane_amd.envs.singlecombat_env import SingleCombatEnv
    from neuralplane_amd.selfplay import OpponentExchange
    dev = torch.device('cuda', 0)
    env = SingleCombatEnv(num_envs=e_loc, config='selfplay', random_seed=21, device='cuda:0', env0=env0)
    W_ego = torch.linspace(-1, 1, 15 * 4, device=dev).reshape(15, 4)
    W_opp = torch.linspace(1, -1, 15 * 4, device=dev).reshape(15, 4)

    def opp_policy(obs, env_ids):       # elementwise in the env dimension and keyed by the GLOBAL env index
        return torch.tanh((obs[:, :, None] * W_opp[None]).sum(1) + (env_ids.to(obs.dtype) % 7)[:, None] * 0.01)

    ex = OpponentExchange(e_loc, env0, e_total, d, dev, opponent_policy=opp_policy, lag=lag)
    obs = env.reset()
    for _ in range(steps):
        a = ex.actions(obs, lambda x: torch.tanh((x[:, :, None] * W_ego[None]).sum(1)))
        obs, rew = env.step(a)[:2]
    torch.cuda.synchronize()
    return env.s.cpu().numpy(), obs.cpu().numpy(), rew.cpu().numpy(), env.blood.cpu().numpy()


def _hip_selfplay_worker(rank, world, port, e_total, steps, lag, q):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, 'tests'))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from neuralplane_amd import sharding
    torch.cuda.set_device(0)
    d = sharding.init_distributed('gloo')
    env0, e_loc = sharding.shard_rows(e_total, world, rank)
    res = _hip_selfplay_loop(e_loc, env0, e_total, d, steps, lag)
    parts = [None] * world
    d.all_gather_object(parts, (env0,) + res)
    if rank == 0:
        q.put(parts)
    d.barrier()
    d.destroy_process_group()


@pytest.mark.parametrize('lag', [0, 1])
def test_two_rank_hip_selfplay_loop_equals_single_process_run(lag):
    """OpponentExchange (two all-gathers per step on a side stream) around the HIP combat kernel, 2 ranks x ragged env shards,
    against the unsharded run in this process: states, observations, rewards and blood bit for bit after 8 env.steps."""
    import socket
    import torch.multiprocessing as mp
    e_total, steps, world = 301, 8, 2
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_hip_selfplay_worker, args=(r, world, port, e_total, steps, lag, q)) for r in range(world)]
    for p in procs:
        p.start()
    parts = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    parts.sort(key=lambda x: x[0])
    got = [np.concatenate([p[k] for p in parts]) for k in (1, 2, 3, 4)]
    want = _hip_selfplay_loop(e_total, 0, e_total, None, steps, lag)
    for g_, w_ in zip(got, want):
        assert np.array_equal(g_, w_)


def test_combat_obs_is_the_observation_of_the_current_state_and_changes_nothing():
    """SingleCombatEnv.obs() (singlecombat_env.py:64-138): no noise in this env, so it must repeat the observation the last step
    returned, and leave state, blood, controller state, counters and flags alone."""
    from neuralplane_amd.envs.singlecombat_env import SingleCombatEnv
    env = SingleCombatEnv(num_envs=96, config='selfplay', random_seed=4, device='cuda:0')
    env.reset()
    g = torch.Generator(device='cuda').manual_seed(1)
    for _ in range(30):
        out = env.step(torch.rand((env.n, 4), device='cuda', generator=g) * 2 - 1)
    b = env._batch
    before = {k: getattr(b, k).clone() for k in ('s', 'u', 'pid', 'blood', 'step_count', 'flags')}
    obs = env.obs()
    assert torch.equal(obs, out[0])
    for k, v in before.items():
        assert torch.equal(getattr(b, k), v), k
    with pytest.raises(RuntimeError, match='fused'):
        env.reward()
    env.update_recent_s(env.s)
    assert env.recent_s[0] is not None
