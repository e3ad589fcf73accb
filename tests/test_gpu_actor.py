"""GPU tests of the fused low-level controller (np_actor_forward, neuralplane_amd/actor.py::FusedActor) through the C ABI:
bit-exact against the oracle's restatement, close to the reference's PPOActor recording, and inside PlanningEnv."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.f16_oracle import ActorOracle  # noqa: E402  (the checker; test infrastructure)


def _sd(d):
    return {k[4:]: d[k] for k in d.files if k.startswith('sd::')}


def _same(a, b):
    return bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


@pytest.mark.parametrize('tile', ['32', '64'])
@pytest.mark.parametrize('n', [1, 31, 32, 33, 63, 64, 65, 1000])
def test_fused_actor_bit_exact_vs_oracle(golden_dir, n, tile, monkeypatch):
    """Both tilings of the controller (32-row tiles for small batches, 64-row tiles above the per-CU bound of np_dispatch.h; NP_ACTOR_TILE forces one)."""
    from neuralplane_amd.actor import FusedActor, pack_ppo_actor
    monkeypatch.setenv('NP_ACTOR_TILE', tile)
    d = np.load(f'{golden_dir}/actor_kat.npz')
    w = pack_ppo_actor(_sd(d))
    fa, o = FusedActor(w, 'cuda:0'), ActorOracle(w)
    rng = np.random.RandomState(n)
    h = (rng.normal(0, 0.5, (n, 1, 128))).astype(np.float32)
    h_o = h[:, 0].copy()
    h_t = torch.from_numpy(h).cuda()
    for t in range(5):
        obs = (rng.normal(0, 1, (n, 22)) * rng.uniform(0.1, 30, (1, 22))).astype(np.float32)
        masks = (rng.uniform(0, 1, (n, 1)) > 0.2).astype(np.float32)
        if t == 3:
            obs[0, 5] = np.float32(1e20)      # saturating inputs: LayerNorm keeps them finite
        a_t, _, h_t = fa(torch.from_numpy(obs).cuda(), h_t, torch.from_numpy(masks).cuda())
        a_o, h_o = o.forward(obs, h_o, masks)
        assert a_t.shape == (n, 4) and h_t.shape == (n, 1, 128)
        assert _same(a_t.cpu().numpy(), a_o), f'actions differ at call {t}'
        assert _same(h_t.cpu().numpy()[:, 0], h_o), f'rnn state differs at call {t}'


def test_fused_actor_tilings_agree_at_the_switch_over(golden_dir, monkeypatch):
    """n on both sides of the automatic switch: the two kernels return the same bits."""
    from neuralplane_amd.actor import FusedActor
    d = np.load(f'{golden_dir}/actor_kat.npz')
    fa = FusedActor(_sd(d), 'cuda:0')
    rng = np.random.RandomState(5)
    for n in (16384, 16385):
        obs = torch.from_numpy((rng.normal(0, 1, (n, 22)) * rng.uniform(0.1, 30, (1, 22))).astype(np.float32)).cuda()
        h = torch.from_numpy(rng.normal(0, 0.5, (n, 1, 128)).astype(np.float32)).cuda()
        m = torch.from_numpy((rng.uniform(0, 1, (n, 1)) > 0.2).astype(np.float32)).cuda()
        res = {}
        for tile in ('', '32', '64'):
            if tile:
                monkeypatch.setenv('NP_ACTOR_TILE', tile)
            else:
                monkeypatch.delenv('NP_ACTOR_TILE', raising=False)
            a, _, h2 = fa(obs, h, m)
            res[tile] = (a.cpu().numpy(), h2.cpu().numpy())
        for tile in ('32', '64'):
            assert _same(res[''][0], res[tile][0]) and _same(res[''][1], res[tile][1]), (n, tile)


def test_fused_actor_close_to_reference_recording(golden_dir):
    from neuralplane_amd.actor import FusedActor
    d = np.load(f'{golden_dir}/actor_kat.npz')
    fa = FusedActor(_sd(d), 'cuda:0')
    steps, n = d['obs'].shape[:2]
    h = torch.zeros((n, 1, 128), device='cuda')
    for t in range(steps):
        a, _, h = fa(torch.from_numpy(d['obs'][t]).cuda(), h, torch.from_numpy(d['masks'][t]).cuda())
    assert np.max(np.abs(a.cpu().numpy() - d['actions'][-1])) < 5e-5
    assert np.max(np.abs(h.cpu().numpy() - d['rnn'][-1])) < 5e-5
    with pytest.raises(ValueError):
        FusedActor(np.zeros(10, np.float32), 'cuda:0')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        FusedActor(_sd(d), 'cpu')


def test_planning_env_with_fused_actor_eager_and_graph(golden_dir):
    """PlanningEnv(controller=FusedActor): the whole macro-step is native (1 + 50 x 3 launches); the HIP-graph replay
    gives the eager results bit for bit, and the low-level actions equal what the oracle's actor computes from the same
    low-level observations."""
    from neuralplane_amd.actor import FusedActor, pack_ppo_actor
    from neuralplane_amd.envs.planning_env import PlanningEnv
    d = np.load(f'{golden_dir}/actor_kat.npz')
    w = pack_ppo_actor(_sd(d))
    n = 200
    envs = [PlanningEnv(num_envs=n, config='tracking', model='F16', random_seed=5, device='cuda:0', controller=FusedActor(w, 'cuda:0'))
            for _ in range(3)]
    envs[0].use_inner_loop = envs[1].use_inner_loop = False   # [0] launch by launch, [1] the same as a HIP graph, [2] np_planning_inner_loop
    envs[1].enable_graph()
    g = torch.Generator(device='cpu').manual_seed(2)
    for k in range(3):
        a = (torch.rand((n, 3), generator=g) * 2 - 1).cuda()
        outs = [e.step(a) for e in envs]
        for other in (1, 2):
            for x, y in zip(outs[0][:5], outs[other][:5]):
                assert torch.equal(x, y), f'macro-step {k}, path {other}'
            assert torch.equal(envs[0].model.s, envs[other].model.s) and torch.equal(envs[0].ego_rnn_states, envs[other].ego_rnn_states)
    # one more inner iteration by hand: oracle actor on the env's low-level observation
    env = envs[0]
    tgt3 = torch.stack((env.model.s[:, 4], env.model.s[:, 5], env.model.s[:, 6]))
    ll = env._batch.lowlevel_obs(tgt3)
    act, _, _ = env.controller(ll, env.ego_rnn_states, torch.ones((n, 1), device='cuda'))
    a_o, _ = ActorOracle(w).forward(ll.cpu().numpy(), env.ego_rnn_states.cpu().numpy()[:, 0], np.ones(n, np.float32))
    assert _same(act.cpu().numpy(), a_o)


@pytest.mark.parametrize('n,groups', [(10_037, 0), (9_001, 3), (700, 2), (64, 4)])
def test_planning_inner_loop_row_groups_equal_the_launch_by_launch_path(golden_dir, n, groups):
    """np_planning_inner_loop: the 50 iterations enqueued by one call, as one or several row groups on their own streams (two
    automatically for 8 192 < n <= 16 384) — states, observation, reward, flags, recurrent state and termination statistics equal
    the launch-by-launch path bit for bit; ragged last group, more groups than 64-row blocks."""
    from neuralplane_amd.actor import FusedActor, pack_ppo_actor
    from neuralplane_amd.envs.planning_env import PlanningEnv
    w = pack_ppo_actor(_sd(np.load(f'{golden_dir}/actor_kat.npz')))
    envs = [PlanningEnv(num_envs=n, config='tracking', model='F16', random_seed=11, device='cuda:0', controller=FusedActor(w, 'cuda:0'))
            for _ in range(2)]
    envs[0].use_inner_loop = False
    envs[1].loop_groups = groups
    for e in envs:
        e.termination_reasons()           # switches the per-aircraft tracking on
    g = torch.Generator(device='cpu').manual_seed(n)
    for k in range(3):
        a = (torch.rand((n, 3), generator=g) * 2 - 1).cuda()
        outs = [e.step(a) for e in envs]
        for x, y in zip(outs[0][:5], outs[1][:5]):
            assert torch.equal(x, y), f'macro-step {k}'
        assert torch.equal(envs[0].model.s, envs[1].model.s) and torch.equal(envs[0].ego_rnn_states, envs[1].ego_rnn_states)
        assert torch.equal(envs[0].termination_reasons(), envs[1].termination_reasons())
        assert torch.equal(envs[0].step_count, envs[1].step_count)
    assert envs[0].termination_counts() == envs[1].termination_counts()


@pytest.mark.parametrize('n,mode,waves,block', [(200, 'persistent', 8, 0), (1, 'persistent', 8, 0), (33, 'persistent', 8, 0),
                                                (8_192, 'persistent', 8, 0), (10_037, 'persistent', 8, 0), (10_037, 'queue', 8, 0), (10_037, 'queue', 8, 1),
                                                (20_011, 'queue', 8, 7), (95, 'queue', 8, 1), (95, 'queue', 8, 50), (700, 'queue', 8, 3),
                                                (10_037, 'guests', 8, 0), (9_001, 'guests', 8, 3), (12_288, 'guests', 8, 1), (16_000, 'guests', 8, 0),
                                                (200, 'guests', 8, 0), (10_037, 'auto', 0, 0), (8_192, 'auto', 0, 0),
                                                (200, 'dual', 8, 0), (33, 'dual', 8, 0), (1, 'dual', 8, 0), (95, 'dual', 8, 0), (10_037, 'dual', 8, 0),
                                                (16_384, 'dual', 8, 0), (20_011, 'dual', 8, 0), (16_385, 'auto', 0, 0)])
def test_planning_persistent_kernel_equals_the_launch_by_launch_path(golden_dir, n, mode, waves, block):
    """np_planning_loop.mode = persistent / queue: all 50 iterations in ONE launch of the persistent kernel (np_planning.hip; a
    workgroup per 32-row tile, or resident workgroups pulling (tile, block of iterations) items, or the static guest schedule — every
    resident workgroup owns a tile and hosts one block of a guest tile in between; eight-wave tiles run the pipelined
    schedule: an inner step's Overload evaluation on waves 4..7 during the next controller call) — states, observation, reward, flags, recurrent
    state, step counters and termination statistics equal the launch-by-launch path bit for bit over several macro-steps (rows that
    terminate mid-step and stay frozen included); ragged last tile; the first macro-step starts from an invalid coefficient cache."""
    from neuralplane_amd.actor import FusedActor, pack_ppo_actor
    from neuralplane_amd.envs.planning_env import PlanningEnv
    w = pack_ppo_actor(_sd(np.load(f'{golden_dir}/actor_kat.npz')))
    envs = [PlanningEnv(num_envs=n, config='tracking', model='F16', random_seed=13, device='cuda:0', controller=FusedActor(w, 'cuda:0'))
            for _ in range(2)]
    envs[0].use_inner_loop = False
    envs[1].loop_mode, envs[1].loop_waves, envs[1].loop_block = mode, waves, block
    for e in envs:
        e.termination_reasons()
        e._batch.track_reward_terms()
    g = torch.Generator(device='cpu').manual_seed(n)
    for k in range(4):
        a = (torch.rand((n, 3), generator=g) * 2 - 1).cuda()
        if k == 2:   # the caller edits the state between macro-steps: the cached coefficients are stale for both paths
            for e in envs:
                e.model.s[: max(1, n // 3), 7] += 0.01
        outs = [e.step(a) for e in envs]
        for x, y in zip(outs[0][:5], outs[1][:5]):
            assert torch.equal(x, y), f'macro-step {k}'
        assert torch.equal(envs[0].model.s, envs[1].model.s) and torch.equal(envs[0].model.u, envs[1].model.u)
        assert torch.equal(envs[0].ego_rnn_states, envs[1].ego_rnn_states)
        assert torch.equal(envs[0].termination_reasons(), envs[1].termination_reasons())
        assert torch.equal(envs[0]._batch.reward_task, envs[1]._batch.reward_task)
        assert torch.equal(envs[0].step_count, envs[1].step_count)
        m = (n // 64) * 16 * 64   # whole 64-row tiles of the cache: 14 coefficients + the 2 key rows (the rows beyond n of a last tile are never written)
        assert torch.equal(envs[0]._batch.coef_cache[: m], envs[1]._batch.coef_cache[: m])
    assert envs[0].termination_counts() == envs[1].termination_counts()
    assert any(v > 0 for v in envs[0].termination_counts().values()), 'the comparison should include rows that terminated'


def test_two_guest_schedule_kernels_on_two_streams_do_not_wait_for_each_other_forever(golden_dir):
    """The guest / queue schedules need all their workgroups resident at once; two such kernels side by side (two envs stepped from two
    streams) could starve each other of CUs.  The library chains them per device by an event: both envs finish, with the results of a
    run on one stream."""
    from neuralplane_amd.actor import FusedActor, pack_ppo_actor
    from neuralplane_amd.envs.planning_env import PlanningEnv
    w = pack_ppo_actor(_sd(np.load(f'{golden_dir}/actor_kat.npz')))
    n = 10_037
    ref = PlanningEnv(num_envs=n, config='tracking', model='F16', random_seed=21, device='cuda:0', controller=FusedActor(w, 'cuda:0'))
    ref.loop_mode = 'launches'
    envs = [PlanningEnv(num_envs=n, config='tracking', model='F16', random_seed=21, device='cuda:0', controller=FusedActor(w, 'cuda:0')) for _ in range(2)]
    envs[0].loop_mode, envs[1].loop_mode = 'guests', 'queue'
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    a = (torch.rand((n, 3), generator=torch.Generator().manual_seed(5)) * 2 - 1).cuda()
    torch.cuda.synchronize()
    outs = [None, None]
    for k in range(4):
        r = ref.step(a)
        for j in (0, 1):
            with torch.cuda.stream(streams[j]):
                outs[j] = envs[j].step(a)
        torch.cuda.synchronize()
        for j in (0, 1):
            for x, y in zip(r[:5], outs[j][:5]):
                assert torch.equal(x, y), f'macro-step {k}, env {j}'


def test_two_processes_run_the_guest_and_queue_schedules_on_one_gpu():
    """Two processes cannot chain their launches: their persistent grids (one eight-wave workgroup per CU each, polling progress words)
    share the GPU, neither fully resident.  Forward progress rests on a workgroup waiting only for a lower-indexed one (np_planning.hip,
    header); each process compares its results with the launch-by-launch path."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'microbench', 'planning_two_procs.py'), '9000', '20', '2'], cwd=root,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:]
    assert 'guests: 2 processes, exit codes [0, 0]' in r.stdout and 'queue: 2 processes, exit codes [0, 0]' in r.stdout, r.stdout[-2000:]


def test_planning_inner_loop_rejects_bad_arguments(golden_dir):
    """np_planning_inner_loop fails loudly (no launch) on aliased ping-pong buffers and on an impossible group count."""
    from neuralplane_amd.actor import FusedActor, pack_ppo_actor
    from neuralplane_amd.envs.planning_env import PlanningEnv
    w = pack_ppo_actor(_sd(np.load(f'{golden_dir}/actor_kat.npz')))
    n = 128
    env = PlanningEnv(num_envs=n, config='tracking', model='F16', random_seed=1, device='cuda:0', controller=FusedActor(w, 'cuda:0'))
    b = env._batch
    b.reset(want_obs=False)
    tgt3 = torch.zeros((3, n), device='cuda')
    ll = [b.lowlevel_obs(tgt3), torch.empty((n, 22), device='cuda')]
    rnn = [torch.zeros((n, 128), device='cuda'), torch.zeros((n, 128), device='cuda')]
    masks, act, fl = torch.ones(n, device='cuda'), torch.empty((n, 4), device='cuda'), torch.empty((3, n), dtype=torch.uint8, device='cuda')
    s0 = b.s.clone()
    with pytest.raises(RuntimeError, match='ping-pong'):
        b.planning_inner_loop(env.controller.weights, ll, [rnn[0], rnn[0]], masks, act, tgt3, fl, 50)
    with pytest.raises(RuntimeError, match='groups'):
        b.planning_inner_loop(env.controller.weights, ll, rnn, masks, act, tgt3, fl, 50, groups=9)
    torch.cuda.synchronize()
    assert torch.equal(b.s, s0), 'a rejected call must not have launched anything'


def test_misaligned_recurrent_state(golden_dir):
    """h_in / h_out are read and written 16 bytes at a time: the raw entry point rejects a misaligned pointer loudly, the
    FusedActor wrapper realigns a view at an odd storage offset and returns the same result."""
    import ctypes as C
    from neuralplane_amd import _lib
    from neuralplane_amd.actor import FusedActor, NUM_FLOATS
    g = np.load(f'{golden_dir}/actor_kat.npz')
    fa = FusedActor(_sd(g), 'cuda:0')
    n = 70
    rng = np.random.RandomState(3)
    obs = torch.from_numpy(rng.normal(0, 1, (n, 22)).astype(np.float32)).cuda()
    big = torch.from_numpy(rng.normal(0, 0.5, (n * 128 + 1,)).astype(np.float32)).cuda()
    m = torch.ones(n, 1, device='cuda')
    h_view = big[1:].view(n, 1, 128)                    # storage offset 1 float: 4-byte aligned only
    assert h_view.data_ptr() % 16 != 0
    a0, _, h0 = fa(obs, h_view.clone(), m)
    a1, _, h1 = fa(obs, h_view, m)
    assert torch.equal(a0, a1) and torch.equal(h0, h1)
    act = torch.empty((n, 4), device='cuda')
    h_out = torch.empty((n, 128), device='cuda')
    rc = fa.lib.np_actor_forward(fa.weights.data_ptr(), NUM_FLOATS, n, obs.data_ptr(), h_view.data_ptr(), m.data_ptr(), act.data_ptr(),
                                 h_out.data_ptr(), 0, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc != 0 and b'aligned' in _lib.load().np_last_error()


def test_planning_env_loads_a_checkpoint_into_the_fused_controller(golden_dir, tmp_path):
    """PlanningEnv(controller='fused', controller_checkpoint=path): the reference's actor_latest.pt format (a PPOActor
    state_dict) runs through the fused kernel; same results as handing the state_dict over directly; a missing file fails loudly."""
    from neuralplane_amd.actor import FusedActor
    from neuralplane_amd.envs.planning_env import PlanningEnv
    d = np.load(f'{golden_dir}/actor_kat.npz')
    sd = {k: torch.from_numpy(v) for k, v in _sd(d).items()}
    path = tmp_path / 'actor_latest.pt'
    torch.save(sd, path)
    n = 200
    act = torch.rand(n, 3, device='cuda') * 2 - 1
    for numerics in ('i8', 'fp32'):      # 'i8' (block fixed point) is what controller='fused' takes unless controller_numerics says otherwise
        kw = {} if numerics == 'i8' else {'controller_numerics': 'fp32'}
        a = PlanningEnv(num_envs=n, config='tracking', model='F16', random_seed=2, device='cuda:0', controller='fused', controller_checkpoint=str(path), **kw)
        b = PlanningEnv(num_envs=n, config='tracking', model='F16', random_seed=2, device='cuda:0', controller=FusedActor(sd, 'cuda:0', numerics=numerics))
        assert a.controller.numerics == numerics
        for _ in range(2):
            ra, rb = a.step(act), b.step(act)
            assert all(torch.equal(x, y) for x, y in zip(ra[:5], rb[:5]))
    with pytest.raises(RuntimeError, match='not found'):
        PlanningEnv(num_envs=4, config='tracking', model='F16', random_seed=0, device='cuda:0', controller='fused', controller_checkpoint=str(tmp_path / 'nope.pt'))


# ---------------------------------------------------------------------------------------------------------------------------------
# PlanningEnv CLOSED LOOP against the reference and against the oracle, first-hand (round 5): tests/golden/planning_closed_kat.npz is the
# reference's own PlanningEnv.step x 3 with a stored actor state_dict (tools/gen_golden.py::gen_planning_closed).
# ---------------------------------------------------------------------------------------------------------------------------------
CLOSED_MODES = [('launches', 0, 0), ('persistent', 8, 0), ('guests', 8, 0), ('queue', 8, 7), ('queue', 8, 0), ('dual', 8, 0), ('auto', 0, 0)]
CLOSED_CASES = [(m, w, b, 'fp32') for m, w, b in CLOSED_MODES] + [(m, w, b, 'i8') for m, w, b in CLOSED_MODES if m != 'dual']


def _closed_env(g, mode, waves, block, numerics='fp32'):
    from neuralplane_amd.actor import FusedActor, pack_ppo_actor
    from neuralplane_amd.envs.planning_env import PlanningEnv
    from tests.planning_closed import actor_state_dict
    w = pack_ppo_actor(actor_state_dict(g))
    n = g['hi_actions'].shape[1]
    env = PlanningEnv(num_envs=n, config='tracking', model='F16', random_seed=0, device='cuda:0', controller=FusedActor(w, 'cuda:0', numerics=numerics))
    if mode == 'eager':
        env.use_inner_loop = False
    else:
        env.loop_mode, env.loop_waves, env.loop_block = mode, waves, block
    return env, w


def _closed_result(env, out):
    obs, rew, done, bad, tmo, _ = out
    return {'s': env.model.s.cpu().numpy(), 'u': env.model.u.cpu().numpy(), 'tgt': env._batch.tgt.cpu().numpy().T.copy(),
            'step_count': env.step_count.cpu().numpy(), 'rnn': env.ego_rnn_states.cpu().numpy()[:, 0], 'obs': obs.cpu().numpy(), 'reward': rew.cpu().numpy(),
            'flags': np.stack([done.cpu().numpy(), bad.cpu().numpy(), tmo.cpu().numpy()]).astype(np.uint8)}


@pytest.mark.parametrize('mode,waves,block,numerics', [('eager', 0, 0, 'fp32'), ('eager', 0, 0, 'i8')] + CLOSED_CASES)
def test_planning_closed_loop_vs_the_reference_recording(golden_dir, mode, waves, block, numerics):
    """PlanningEnv(controller=FusedActor).step x 3 — 150 closed-loop inner steps, the recurrent state feeding back — against the REFERENCE's
    own PlanningEnv.step on the same actor state_dict, high-level actions and reset draws (envs/planning_env.py:144-177,
    algorithms/ppo/ppo_actor.py:38-64): every mask and counter equal, states <= 1e-4 (SURVEY §8(d) floors), recurrent state <= 5e-5,
    observation <= 1e-4 — the launch-by-launch path, the 2 x 50-launch call and every schedule of the persistent kernel, with the fp32 controller
    and with the block-fixed-point one (measured: states 3.9e-5 / 4.1e-5, recurrent state 1.5e-5 / 1.8e-5)."""
    from tests.planning_closed import compare_with_reference
    g = np.load(f'{golden_dir}/planning_closed_kat.npz')
    env, _ = _closed_env(g, mode, waves, block, numerics)
    for k in range(g['hi_actions'].shape[0]):
        env._batch.reset(rand_u=g[f'rand_u_{k}'], want_obs=False)     # the reference's draws: PlanningEnv.step begins with self.reset()
        out = env.step(torch.from_numpy(g['hi_actions'][k]).cuda())
        compare_with_reference(_closed_result(env, out), g, k)        # measured (profiles/r05_parity.json): states 3.9e-5, recurrent state 1.5e-5
    assert int((env.step_count == 150).sum()) >= 30


@pytest.mark.parametrize('mode,numerics', [('auto', 'i8'), ('auto', 'fp32'), ('launches', 'i8'), ('launches', 'fp32'), ('eager', 'i8')])
def test_planning_closed_loop_over_1000_inner_steps_vs_the_reference_and_the_oracle(golden_dir, mode, numerics):
    """PlanningEnv(controller=FusedActor).step x 20 = 1 000 closed-loop inner steps against the REFERENCE's own
    PlanningEnv.step x 20 on the same actor state_dict / high-level actions / reset draws (tests/golden/planning_closed_long_kat.npz;
    reference envs/planning_env.py:144-177, algorithms/ppo/ppo_actor.py:38-64; 62 of 64 rows never terminate), for the default
    block-fixed-point controller and the fp32 one: after EVERY macro-step masks and counters equal, states <= 1e-4 (SURVEY floors),
    recurrent state <= 5e-5, observation <= 1e-4 — and bit-identical to the CPU oracle's closed loop all the way (persistent kernel and
    launch-by-launch).  The per-numerics worst errors go to gpurun_out/parity_planning_long.json."""
    import json
    import os
    from tests.planning_closed import OracleClosedLoop, compare_with_reference
    g = np.load(f'{golden_dir}/planning_closed_long_kat.npz')
    env, w = _closed_env(g, mode, 0, 0, numerics)
    cl = OracleClosedLoop(g, w, numerics) if mode != 'eager' else None
    worst = {}
    for k in range(g['hi_actions'].shape[0]):
        env._batch.reset(rand_u=g[f'rand_u_{k}'], want_obs=False)
        got = _closed_result(env, env.step(torch.from_numpy(g['hi_actions'][k]).cuda()))
        for q, v in compare_with_reference(got, g, k).items():
            worst[q] = max(worst.get(q, 0.0), v)
        if cl is not None:
            want = cl.macro_step(k)
            for q in ('flags', 'step_count', 's', 'u', 'tgt', 'rnn', 'obs', 'reward'):
                assert _same(got[q], want[q]), f'macro-step {k}: {q} differs from the oracle'
    assert int(got['step_count'][g['never_flagged']].min()) == 1000 and int(g['never_flagged'].sum()) >= 32
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        os.makedirs(os.path.join(root, 'gpurun_out'), exist_ok=True)
        path = os.path.join(root, 'gpurun_out', 'parity_planning_long.json')
        rep = json.load(open(path)) if os.path.exists(path) else {}
        rep[f'{numerics}/{mode}'] = {'inner_steps': 1000, 'rows': int(g['hi_actions'].shape[1]), 'rows_never_terminated': int(g['never_flagged'].sum()),
                                     'worst_over_20_macro_steps_vs_reference': worst, 'bit_identical_to_oracle_closed_loop': cl is not None}
        json.dump(rep, open(path, 'w'), indent=1)
    except OSError:
        pass


@pytest.mark.parametrize('mode,waves,block,numerics', CLOSED_CASES)
def test_planning_macro_steps_equal_the_oracle_closed_loop_bit_for_bit(golden_dir, mode, waves, block, numerics):
    """DIRECT: whole macro-steps of the persistent kernel (each schedule; and the 2 x 50-launch call) against Oracle.reset / lowlevel_obs /
    step_inner + ActorOracle run closed loop on the CPU — states, controls, targets, counters, recurrent state, observation, reward and
    masks bit for bit after every one of three macro-steps (rows that terminate mid-step and stay frozen included)."""
    from tests.planning_closed import OracleClosedLoop
    g = np.load(f'{golden_dir}/planning_closed_kat.npz')
    env, w = _closed_env(g, mode, waves, block, numerics)
    cl = OracleClosedLoop(g, w, numerics)
    for k in range(g['hi_actions'].shape[0]):
        env._batch.reset(rand_u=g[f'rand_u_{k}'], want_obs=False)
        got = _closed_result(env, env.step(torch.from_numpy(g['hi_actions'][k]).cuda()))
        want = cl.macro_step(k)
        for q in ('flags', 'step_count', 's', 'u', 'tgt', 'rnn', 'obs', 'reward'):
            assert _same(got[q], want[q]), f'macro-step {k}: {q} differs from the oracle (max abs {np.nanmax(np.abs(got[q].astype(np.float64) - want[q]))})'
    assert int(want['flags'][1].sum()) > 0 and int((want['step_count'] == 150).sum()) >= 30


@pytest.mark.parametrize('mode,n', [('guests', 10_037), ('queue', 10_037), ('queue', 700)])
def test_a_stalled_guest_or_queue_schedule_is_an_error_and_a_clean_fallback_not_a_hang(golden_dir, monkeypatch, mode, n):
    """Bounded waits (include/neuralplane_amd.h, ABI 15).  Fault injection: NP_PLANNING_DEBUG_STALL makes workgroup 0 never raise a progress
    word, so its successors' waits expire (NP_PLANNING_WAIT_MS = 40 here, 2 s shipped).  The kernel must END; np_planning_inner_loop returns
    NP_E_PLANNING_STALLED naming workgroup / tile / iteration with every in-place buffer restored; PlanningEnv re-runs the macro-step launch by
    launch (a RuntimeWarning, env.loop_fallbacks) — and the results equal the launch-by-launch path bit for bit, this macro-step and the
    following ones (which run the schedule again, un-stalled, on freshly cleared queue words)."""
    from neuralplane_amd import _lib
    from neuralplane_amd.actor import FusedActor, pack_ppo_actor
    from neuralplane_amd.envs.planning_env import PlanningEnv
    w = pack_ppo_actor(_sd(np.load(f'{golden_dir}/actor_kat.npz')))
    envs = [PlanningEnv(num_envs=n, config='tracking', model='F16', random_seed=17, device='cuda:0', controller=FusedActor(w, 'cuda:0')) for _ in range(2)]
    envs[0].loop_mode = 'launches'
    envs[1].loop_mode = mode
    for e in envs:
        e.termination_reasons()
    g = torch.Generator(device='cpu').manual_seed(n)
    monkeypatch.setenv('NP_PLANNING_WAIT_MS', '40')
    for k in range(4):
        a = (torch.rand((n, 3), generator=g) * 2 - 1).cuda()
        ref = envs[0].step(a)
        if k == 1:
            monkeypatch.setenv('NP_PLANNING_DEBUG_STALL', '1')
            with pytest.warns(RuntimeWarning, match='re-running this macro-step launch by launch'):
                out = envs[1].step(a)
            monkeypatch.delenv('NP_PLANNING_DEBUG_STALL')
            assert envs[1].loop_fallbacks == 1
        else:
            out = envs[1].step(a)
        for x, y in zip(ref[:5], out[:5]):
            assert torch.equal(x, y), f'macro-step {k}'
        assert torch.equal(envs[0].model.s, envs[1].model.s) and torch.equal(envs[0].model.u, envs[1].model.u)
        assert torch.equal(envs[0].ego_rnn_states, envs[1].ego_rnn_states) and torch.equal(envs[0].step_count, envs[1].step_count)
        assert torch.equal(envs[0].termination_reasons(), envs[1].termination_reasons())
    assert envs[0].termination_counts() == envs[1].termination_counts() and envs[1].loop_fallbacks == 1
    # the raw entry point: the error names what waited for what, and check = deferred delivers the verdict with the next call
    env = envs[1]
    monkeypatch.setenv('NP_PLANNING_DEBUG_STALL', '1')
    env.loop_check = 'deferred'
    env.step(a)                                   # returns at once; nothing is kept
    with pytest.raises(_lib.PlanningStalled, match=r'workgroup \d+ waited \d+ ms for tile \d+ to reach iteration \d+') as ei:
        env.step(a)
    assert not ei.value.restored
    monkeypatch.delenv('NP_PLANNING_DEBUG_STALL')
    assert _lib.load().np_planning_check(env._batch._ctx) == 0


def test_guest_and_queue_schedules_refuse_a_capturing_stream(golden_dir):
    """their counter / progress-word bases are host state baked into the launch — a replayed graph would silently compute nothing
    (queue) or import tiles before they were exported (guests).  Explicit requests fail during capture; 'auto' picks a capturable mode."""
    from neuralplane_amd.actor import FusedActor, pack_ppo_actor
    from neuralplane_amd.envs.planning_env import PlanningEnv
    w = pack_ppo_actor(_sd(np.load(f'{golden_dir}/actor_kat.npz')))
    n = 10_037
    env = PlanningEnv(num_envs=n, config='tracking', model='F16', random_seed=3, device='cuda:0', controller=FusedActor(w, 'cuda:0'))
    a = torch.zeros((n, 3), device='cuda')
    env.step(a)                                   # buffers exist, nothing allocates during the capture below
    torch.cuda.synchronize()
    for mode in ('guests', 'queue'):
        env.loop_mode = mode
        side = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        with pytest.raises(RuntimeError, match='cannot be captured'):
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    env.step(a)
        torch.cuda.synchronize()
    env.loop_mode = 'auto'
    env.step(a)


# ---------------------------------------------------------------------------------------------------------------------------------
# The controller's second numerics spec: block fixed point on the i8 matrix pipe (csrc/np_actor_i8.h; restated in f16_actor_i8.inc)
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('n', [1, 31, 32, 33, 64, 65, 97, 1000, 10_037])
def test_i8_actor_bit_exact_vs_its_integer_restatement(golden_dir, n):
    """np_actor_forward with the NP_ACTOR_I8_NUM_FLOATS buffer == ActorOracle(numerics='i8') bit for bit over five consecutive calls: ragged
    last tile, masked rows, |h| > 1, observation scales 0.1 .. 30, a saturating input."""
    from neuralplane_amd.actor import FusedActor, pack_ppo_actor
    d = np.load(f'{golden_dir}/actor_kat.npz')
    w = pack_ppo_actor(_sd(d))
    fa, o = FusedActor(w, 'cuda:0', numerics='i8'), ActorOracle(w, 'i8')
    rng = np.random.RandomState(n)
    h = (rng.normal(0, 0.5, (n, 1, 128))).astype(np.float32)
    h_o = h[:, 0].copy()
    h_t = torch.from_numpy(h).cuda()
    for t in range(5):
        obs = (rng.normal(0, 1, (n, 22)) * rng.uniform(0.1, 30, (1, 22))).astype(np.float32)
        masks = (rng.uniform(0, 1, (n, 1)) > 0.2).astype(np.float32)
        if t == 3:
            obs[0, 5] = np.float32(1e20)
        a_t, _, h_t = fa(torch.from_numpy(obs).cuda(), h_t, torch.from_numpy(masks).cuda())
        a_o, h_o = o.forward(obs, h_o, masks)
        assert _same(h_t.cpu().numpy()[:, 0], h_o), f'rnn state differs at call {t}: max {np.nanmax(np.abs(h_t.cpu().numpy()[:, 0] - h_o))}'
        assert _same(a_t.cpu().numpy(), a_o), f'actions differ at call {t}'


def test_i8_actor_close_to_reference_recording(golden_dir):
    """The same bound the fp32 kernel is held to against the reference PPOActor's own recording (tighter: measured 5.2e-6 / 2.5e-6)."""
    from neuralplane_amd.actor import FusedActor
    d = np.load(f'{golden_dir}/actor_kat.npz')
    fa = FusedActor(_sd(d), 'cuda:0', numerics='i8')
    steps, n = d['obs'].shape[:2]
    h = torch.zeros((n, 1, 128), device='cuda')
    for t in range(steps):
        a, _, h = fa(torch.from_numpy(d['obs'][t]).cuda(), h, torch.from_numpy(d['masks'][t]).cuda())
        assert np.max(np.abs(a.cpu().numpy() - d['actions'][t])) < 2e-5 and np.max(np.abs(h.cpu().numpy()[:, 0] - d['rnn'][t][:, 0])) < 2e-5
