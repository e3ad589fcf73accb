"""GPU parity of the fused HIP step against the CPU oracle and the reference's golden vectors.

All calls go through the C ABI (neuralplane_amd.core.F16Batch -> libneuralplane_hip.so).
Bar: HIP == oracle BIT-EXACT (same numerics spec) for states, targets,
observations, rewards, step counters and the three masks; HIP vs the reference's plain goldens
within 1e-4 relative (BASELINE.json), masks exact in teacher-forced mode.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.f16_oracle import MODE_PWL, Oracle  # noqa: E402  (the checker; test infrastructure)

TASKS = ['heading', 'control', 'tracking']


def _batch(task, n, solver=None, seed=0, row0=0, tables=False, variant='auto'):
    from neuralplane_amd.core import F16Batch
    from neuralplane_amd.envs.utils.utils import parse_config
    b = F16Batch(n, parse_config(task), task, 'cuda:0', seed=seed, solver=solver, row0=row0, aero_1d_tables=tables)
    b.set_kernel_variant(variant)   # 'auto' picks the 4-waves-per-tile latency kernel at these sizes
    return b


VARIANTS = ['latency', 'latency4w', 'latency8', 'latency2', 'throughput', 'pair']   # 'pair' and 'latency8' fall back to the throughput kernel in the 1-D table mode


def _load_state(b, st):
    b.s.copy_(torch.from_numpy(st['s'].T.copy()))
    b.u.copy_(torch.from_numpy(st['u'].T.copy()))
    b.tgt.copy_(torch.from_numpy(st['tgt'].T.copy()))
    b.step_count.copy_(torch.from_numpy(st['step_count']))
    b.flags.copy_(torch.from_numpy(np.stack([st['done'], st['bad'], st['timeout']])))


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def _check_equal(b, obs, rew, flags, st, o_obs, o_rew, what):
    assert _same(b.s.cpu().numpy().T, st['s']), f'{what}: state differs from oracle'
    assert _same(b.u.cpu().numpy().T, st['u']), f'{what}: controls differ'
    assert _same(b.tgt.cpu().numpy().T, st['tgt']), f'{what}: targets differ'
    assert np.array_equal(b.step_count.cpu().numpy(), st['step_count']), f'{what}: step_count differs'
    f = flags.cpu().numpy()
    assert np.array_equal(f[0], st['done']) and np.array_equal(f[1], st['bad']) and np.array_equal(f[2], st['timeout']), \
        f'{what}: masks differ'
    assert _same(obs.cpu().numpy(), o_obs), f'{what}: obs differs'
    if rew is not None:
        assert _same(rew.cpu().numpy(), o_rew), f'{what}: reward differs'


@pytest.mark.parametrize('variant', VARIANTS)
@pytest.mark.parametrize('tables', [False, True], ids=['mlp', 'aero_1d_tables'])
@pytest.mark.parametrize('task,fixture,solver', [('heading', 'step_kat_heading', None), ('control', 'step_kat_control', None),
                                                 ('tracking', 'step_kat_tracking', None),
                                                 ('heading', 'step_kat_heading_rk4', 'rk4')])
def test_step_kat_bit_exact_vs_oracle_and_close_to_reference(task, fixture, solver, tables, variant, golden_dir):
    g = np.load(f'{golden_dir}/{fixture}.npz')
    n = g['action'].shape[0]
    for pre, ru, nz in [('', 'rand_u', 'noise'), ('first_', 'first_rand_u', 'first_noise')]:
        if pre == '':
            st = {k: g['in_' + k].copy() for k in ['s', 'u', 'tgt', 'step_count', 'done', 'bad', 'timeout']}
        else:
            st = Oracle.new_state(n)
        b = _batch(task, n, solver=solver, tables=tables, variant=variant)
        _load_state(b, st)
        obs, rew, flags = b.step(torch.from_numpy(g['action']).cuda(), rand_u=g[ru], noise=g[nz])
        torch.cuda.synchronize()
        o = Oracle(task, solver=solver, mode=MODE_PWL if tables else 0)
        o_obs, o_rew, _, _, _ = o.step(st, g['action'], rand_u=g[ru], noise=g[nz])
        _check_equal(b, obs, rew, flags, st, o_obs, o_rew, f'{fixture}/{pre or "mid"}')
        # reference (plain ATen arithmetic): 1e-4 relative with per-state floors, masks exact
        key = pre + 'out_'
        floors = np.array([100, 100, 100, .1, .1, .1, 10, .1, .1, .1, .1, .1], np.float32)
        ref_s = g[key + 's']
        err = np.abs(b.s.cpu().numpy().T - ref_s) / np.maximum(np.abs(ref_s), floors)
        assert np.nanmax(err) <= 1e-4, f'{fixture}: state vs reference {np.nanmax(err)}'
        f = flags.cpu().numpy()
        for k, name in enumerate(['done', 'bad', 'timeout']):
            assert np.array_equal(f[k], g[key + name]), f'{fixture}: {name} mask differs from the reference'
        ref_obs = g[key + 'obs']
        eo = np.abs(obs.cpu().numpy() - ref_obs) / np.maximum(np.abs(ref_obs), 1e-1)
        assert np.nanmax(eo) <= 1e-4


@pytest.mark.parametrize('variant', VARIANTS)
@pytest.mark.parametrize('tables', [False, True], ids=['mlp', 'aero_1d_tables'])
@pytest.mark.parametrize('task', TASKS)
def test_free_running_production_rng_bit_exact_vs_oracle(task, tables, variant):
    """reset + 60 free-running steps with the in-kernel Philox RNG: HIP == oracle bit for bit,
    including auto-resets (hazard-rich actions) and the ragged tail of the last workgroup."""
    n, steps, seed, row0 = 1000, 60, 1234, 7_000_000_000
    b = _batch(task, n, seed=seed, row0=row0, tables=tables, variant=variant)
    o = Oracle(task, mode=MODE_PWL if tables else 0)
    st = Oracle.new_state(n)
    rng = np.random.RandomState(5)
    obs = b.reset()
    o_obs = o.reset(st, seed=seed, call_idx=0, row0=row0)
    _check_equal(b, obs, None, b.flags, st, o_obs, None, f'{task}: reset')
    n_resets = 0   # rows flagged by a step that a LATER step of this run re-initialises (the auto-reset path, in-kernel reset draws)
    for t in range(steps):
        a = rng.uniform(-1.5, 1.5, (n, 4)).astype(np.float32)
        a[:, 1] *= 3.0 if t % 7 == 0 else 1.0
        obs, rew, flags = b.step(torch.from_numpy(a).cuda())
        o_obs, o_rew, _, _, _ = o.step(st, a, seed=seed, call_idx=t + 1, row0=row0)
        _check_equal(b, obs, rew, flags, st, o_obs, o_rew, f'{task}: step {t}')
        if t < steps - 1:
            n_resets += int((flags.cpu().numpy() != 0).any(axis=0).sum())
    # the hazard-rich actions end episodes all along the run (the oracle alone: 1005 auto-resets per task in these 60 steps)
    assert n_resets >= 500, f'{task}: only {n_resets} auto-resets happened — the free-running run did not exercise the reset path'


def test_derived_getters_bit_exact_vs_oracle(golden_dir):
    g = np.load(f'{golden_dir}/getters_kat.npz')
    n = g['s'].shape[0]
    b = _batch('heading', n)
    b.s.copy_(torch.from_numpy(g['s'].T.copy()))
    b.u.copy_(torch.from_numpy(g['u'].T.copy()))
    d = b.derived().cpu().numpy()
    o = Oracle('heading')
    x17 = np.hstack([g['s'], g['u']]).astype(np.float32)
    assert _same(d[0:12].T, o.nlplant(x17))
    assert _same(d[12:15].T, o.get_acceleration(g['s'], g['u']))
    assert _same(d[15:18].T, o.get_accels(g['s'], g['u']))
    assert _same(d[18], o.get_eas2tas(g['s']))
    assert _same(d[20:23].T, o.get_atmos(g['s']))                  # F16Model.get_atmos: (mach, qbar, ps)
    assert np.abs(d[20:23].T - g['atmos']).max() / 2000 < 1e-6
    # vs the reference itself
    err = np.abs(d[12:15].T - g['accel']) / np.maximum(np.abs(g['accel']), 1.0)
    assert err.max() < 1e-4
    assert np.abs(d[18] - g['eas2tas']).max() < 1e-6 and np.abs(d[19] - g['eas']).max() / 1000 < 1e-6


@pytest.mark.parametrize('task', TASKS)
def test_env_obs_is_the_observation_of_the_current_state_and_changes_nothing(task):
    """BaseEnv.obs() (env_base.py:58-59): what task.get_obs returns for the state as it is — here one launch of the reset kernel
    with all-clear flags.  Equal to the oracle's observation of that state (its noise draw keyed by the same call counter), and
    state, targets, counters and the flags left by the last step are untouched, flagged rows included."""
    from neuralplane_amd.envs.control_env import ControlEnv
    n, seed = 200, 9
    env = ControlEnv(num_envs=n, config=task, model='F16', random_seed=seed, device='cuda:0')
    env.reset()
    rng = np.random.RandomState(2)
    for _ in range(40):
        a = torch.from_numpy(rng.uniform(-1.5, 1.5, (n, 4)).astype(np.float32)).cuda()
        last = env.step(a)
    b = env._batch
    b.s[7, :5] = 1.2                                                  # push a few rows over a limit: flagged by the next step
    last = env.step(a)
    before = {k: getattr(b, k).clone() for k in ('s', 'u', 'tgt', 'step_count', 'flags')}
    assert before['flags'].any()
    call_idx = b.call_idx
    obs = env.obs()
    for k, v in before.items():
        assert torch.equal(getattr(b, k), v), k
    assert b.call_idx == call_idx + 1
    st = dict(s=before['s'].t().cpu().numpy().copy(), u=before['u'].t().cpu().numpy().copy(),
              tgt=before['tgt'].t().cpu().numpy().copy(), step_count=before['step_count'].cpu().numpy().copy(),
              done=np.zeros(n, np.uint8), bad=np.zeros(n, np.uint8), timeout=np.zeros(n, np.uint8))
    o_obs = Oracle(task).reset(st, seed=seed, call_idx=call_idx, row0=0)
    assert _same(obs.cpu().numpy(), o_obs)
    if env.task.noise_scale == 0:                                     # tracking.yaml: the step's own observation, again
        assert torch.equal(obs, last[0])
    assert env.task.get_obs(env).shape == obs.shape
    for fn in (env.reward, env.done, lambda: env.task.get_reward(env), lambda: env.task.reset(env)):
        with pytest.raises(RuntimeError, match='fused'):
            fn()


def test_cross_step_cache_is_invalidated_by_external_state_edits():
    """The 14 cached aero coefficients are valid only while nobody but the kernels wrote `s`.  Editing the
    state through the reference-style views (`model.s[rows] = ...`, planning_env.py:166) between steps must
    fall back to the un-cached kernel for that step; results stay bit-identical to the oracle either way."""
    from neuralplane_amd.envs.control_env import ControlEnv
    n, seed = 700, 3
    env = ControlEnv(num_envs=n, config='heading', model='F16', random_seed=seed, device='cuda:0')
    o = Oracle('heading')
    st = Oracle.new_state(n)
    rng = np.random.RandomState(9)
    env.reset()
    o.reset(st, seed=seed, call_idx=0)
    b = env._batch
    used_cached = []
    for t in range(12):
        if t in (4, 9):  # push some aircraft to another attitude / speed behind the kernels' back
            rows = torch.arange(0, n, 7, device='cuda')
            s_view = env.model.s
            s_view[rows, 7] = 0.21          # alpha
            s_view[rows, 8] = -0.05         # beta
            s_view[rows, 6] = 640.0         # vt
            st['s'][::7, 7], st['s'][::7, 8], st['s'][::7, 6] = 0.21, -0.05, 640.0
        a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
        valid_before = b._cache_valid and b.s._version == b._s_version
        used_cached.append(bool(valid_before))
        obs, rew, done, bad, tmo, _ = env.step(torch.from_numpy(a).cuda())
        o_obs, o_rew, o_done, o_bad, _ = o.step(st, a, seed=seed, call_idx=t + 1)
        assert _same(env.model.s.cpu().numpy(), st['s']), f'state differs at step {t}'
        assert _same(obs.cpu().numpy(), o_obs) and _same(rew.cpu().numpy(), o_rew), f'obs/reward differ at step {t}'
        assert np.array_equal(bad.cpu().numpy(), o_bad.astype(bool)) and np.array_equal(done.cpu().numpy(), o_done.astype(bool))
    assert used_cached[0] is False and used_cached[4] is False and used_cached[9] is False
    assert all(used_cached[i] for i in (1, 2, 3, 5, 6, 7, 8, 10, 11))


@pytest.mark.parametrize('variant', ['latency8', 'latency', 'latency4w', 'latency2', 'pair', 'throughput'])
@pytest.mark.parametrize('solver', ['euler', 'rk4'])
def test_cross_step_cache_validates_itself_against_invisible_state_edits(variant, solver):
    """writes that bump no version counter — `env.model.s.data[...] = `, a write through an alias torch does not
    know to be one (DLPack; what a foreign kernel holding the raw pointer does) — used to leave the next step integrating with the
    coefficients of the OLD (alpha, beta).  The cache now carries the (alpha, beta) its coefficients belong to and the step kernel
    re-evaluates them in the waves that find a difference: bit-identical to the oracle whoever wrote the state, with the cached kernel
    still chosen (the version counter never moved)."""
    from neuralplane_amd.envs.control_env import ControlEnv
    if solver == 'rk4' and variant.startswith('latency'):
        pytest.skip('the latency family serves the Euler solver')
    n, seed = 1300, 5
    env = ControlEnv(num_envs=n, config='heading', model='F16', random_seed=seed, device='cuda:0', solver=solver)
    b = env._batch
    b.set_kernel_variant(variant)
    o = Oracle('heading', solver=solver)
    st = Oracle.new_state(n)
    rng = np.random.RandomState(17)
    env.reset()
    o.reset(st, seed=seed, call_idx=0)
    alias = torch.from_dlpack(torch.utils.dlpack.to_dlpack(b.s))   # same memory, its own version counter
    assert alias.data_ptr() == b.s.data_ptr()
    cached = []
    for t in range(10):
        if t == 3:    # `.data`: no version bump
            v0 = b.s._version
            env.model.s.data[5::11, 7] = 0.17
            env.model.s.data[5::11, 8] = 0.04
            assert b.s._version == v0
            st['s'][5::11, 7], st['s'][5::11, 8] = 0.17, 0.04
        if t == 6:    # the raw pointer: one row in the middle of a tile, one in the ragged last tile; alpha only / beta only
            v0 = b.s._version
            alias[7, 700] = -0.08
            alias[8, n - 1] = 0.11
            assert b.s._version == v0
            st['s'][700, 7], st['s'][n - 1, 8] = -0.08, 0.11
        a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
        cached.append(bool(b._cache_valid and b.s._version == b._s_version))
        obs, rew, done, bad, tmo, _ = env.step(torch.from_numpy(a).cuda())
        o_obs, o_rew, o_done, o_bad, _ = o.step(st, a, seed=seed, call_idx=t + 1)
        assert _same(env.model.s.cpu().numpy(), st['s']), f'state differs at step {t}'
        assert _same(obs.cpu().numpy(), o_obs) and _same(rew.cpu().numpy(), o_rew), f'obs/reward differ at step {t}'
        assert np.array_equal(bad.cpu().numpy(), o_bad.astype(bool)) and np.array_equal(done.cpu().numpy(), o_done.astype(bool))
    assert cached[3] and cached[6], 'the edits must have been invisible to the host-side check (else this test proves nothing)'


def test_env_surface_matches_reference_contract():
    """Shapes, dtypes, attributes and getters of the ControlEnv / GPUVecEnv surface."""
    from neuralplane_amd.envs.control_env import ControlEnv
    from neuralplane_amd.envs.env_wrappers import GPUVecEnv
    n = 130
    venv = GPUVecEnv([lambda: ControlEnv(num_envs=n, config='control', model='F16', random_seed=1, device='cuda:0')])
    env = venv.env
    assert (env.n, env.num_agents, env.num_observation, env.num_actions) == (n, 1, 22, 4)
    assert env.observation_space.shape == (22,) and env.action_space.shape == (4,)
    assert env.is_done.dtype == torch.bool and bool(env.is_done.all()) and env.step_count.dtype == torch.int64
    obs = venv.reset()
    assert obs.shape == (n, 1, 22) and obs.dtype == np.float32 and not env.is_done.any()
    o, r, d, b, t, info = venv.step(np.zeros((n, 1, 4), np.float32))
    assert o.shape == (n, 1, 22) and r.shape == (n, 1, 1) and d.shape == b.shape == t.shape == (n, 1, 1) and info == {}
    assert d.dtype == np.bool_ and r.dtype == np.float32 and int(env.step_count[0]) == 1
    m = env.model
    assert m.s.shape == (n, 12) and m.u.shape == (n, 5) and m.dt == 0.02
    assert all(x.shape == (n,) for x in m.get_position() + m.get_posture() + m.get_angular_velocity() + m.get_acceleration())
    assert m.get_G().shape == (n,) and m.get_EAS().shape == (n,) and m.get_extended_state().shape == (n, 17)
    assert torch.allclose(m.get_TAS(), m.get_vt()) and env.task.target_pitch.shape == (n,) and env.task.noise_scale == 0.01
    # flags returned by step are fresh tensors: the next step does not mutate them
    d_prev = env.is_done.clone()
    keep = env.is_done
    venv.step(np.ones((n, 1, 4), np.float32))
    assert torch.equal(keep, d_prev)


def test_checkpoint_resume_is_bit_exact_and_device_vec_env_keeps_tensors_on_gpu(tmp_path):
    from neuralplane_amd.envs.control_env import ControlEnv
    from neuralplane_amd.envs.env_wrappers import DeviceVecEnv
    n = 900
    mk = lambda: DeviceVecEnv([lambda: ControlEnv(num_envs=n, config='tracking', model='F16', random_seed=4, device='cuda:0')])  # noqa: E731
    g = torch.Generator(device='cuda')
    g.manual_seed(0)
    acts = [torch.rand((n, 1, 4), generator=g, device='cuda') * 2.4 - 1.2 for _ in range(40)]
    a_env = mk()
    obs = a_env.reset()
    assert obs.is_cuda and obs.shape == (n, 1, 22)
    for a in acts[:25]:
        out = a_env.step(a)
    assert all(t.is_cuda for t in out[:5]) and out[1].shape == (n, 1, 1) and out[2].dtype == torch.bool
    torch.save(a_env.env.state_dict(), tmp_path / 'env.pt')
    ref = [a_env.step(a) for a in acts[25:]]
    b_env = mk()
    b_env.env.load_state_dict(torch.load(tmp_path / 'env.pt'))
    for a, r in zip(acts[25:], ref):
        out = b_env.step(a)
        for x, y in zip(out[:5], r[:5]):
            assert torch.equal(x, y)
    assert torch.equal(b_env.env.model.s, a_env.env.model.s) and torch.equal(b_env.env.step_count, a_env.env.step_count)
    assert bool(ref[-1][3].any()) or bool(a_env.env.step_count.min() < 40)   # the resumed stretch really contained resets


@pytest.mark.parametrize('variant', ['auto'] + VARIANTS)
def test_planning_env_bit_exact_vs_oracle_and_close_to_reference(golden_dir, variant):
    """PlanningEnv mirror (reset + 50 x {low-level obs kernel, controller, inner fused step}) with the reference's
    recorded low-level actions replayed as the controller: == oracle bit for bit, masks == reference — whichever kernel
    variant runs the inner steps."""
    from neuralplane_amd.envs.planning_env import PlanningEnv
    g = np.load(f'{golden_dir}/planning_kat.npz')
    hi = g['hi_actions']
    n = hi.shape[1]

    class Replay:
        def __init__(self):
            self.k, self.i, self.obs_seen = 0, 0, []

        def __call__(self, obs, rnn, masks, deterministic=True):
            a = torch.from_numpy(g[f'll_act_{self.k}'][self.i]).cuda()
            self.obs_seen.append(obs)
            self.i += 1
            return a, None, rnn

    ctrl = Replay()
    env = PlanningEnv(num_envs=n, config='tracking', model='F16', random_seed=0, device='cuda:0', controller=ctrl)
    env._batch.set_kernel_variant(variant)
    o = Oracle('tracking')
    st = Oracle.new_state(n)
    for k in range(hi.shape[0]):
        ctrl.k, ctrl.i, ctrl.obs_seen = k, 0, []
        # inject the reference's reset draws: PlanningEnv.step begins with self.reset()
        env._batch.reset(rand_u=g[f'rand_u_{k}'], want_obs=False)
        obs, rew, done, bad, tmo, _ = env.step(torch.from_numpy(hi[k]).cuda())
        o.reset(st, rand_u=g[f'rand_u_{k}'], want_obs=False)
        a = np.clip(hi[k], -1, 1).astype(np.float32)
        tgt3 = np.stack([st['s'][:, 4] + a[:, 0] * np.float32(0.3), st['s'][:, 5] + a[:, 1] * np.float32(0.3),
                         st['s'][:, 6] + a[:, 2] * np.float32(30)], 1).astype(np.float32)
        for i in range(50):
            ll = o.lowlevel_obs(st, tgt3)
            assert _same(ctrl.obs_seen[i].cpu().numpy(), ll), f'low-level obs differs (outer {k}, inner {i})'
            o_obs, o_rew, o_d, o_b, o_t = o.step_inner(st, g[f'll_act_{k}'][i])
        assert _same(env.model.s.cpu().numpy(), st['s']) and _same(env.model.u.cpu().numpy(), st['u'])
        assert _same(obs.cpu().numpy(), o_obs) and _same(rew.cpu().numpy(), o_rew)
        assert np.array_equal(env.step_count.cpu().numpy(), st['step_count'])
        fl = g[f'flags_{k}']
        for got, orc, ref in ((done, o_d, fl[0]), (bad, o_b, fl[1]), (tmo, o_t, fl[2])):
            assert np.array_equal(got.cpu().numpy(), orc.astype(bool)) and np.array_equal(orc, ref)
    with pytest.raises(RuntimeError):
        PlanningEnv(num_envs=4, config='tracking', model='F16', random_seed=0, device='cuda:0')   # no controller, no checkpoint
    with pytest.raises(NotImplementedError):
        PlanningEnv(num_envs=4, config='heading', model='F16', random_seed=0, device='cuda:0', controller=ctrl)


def test_render_writes_tacview_frames(tmp_path):
    """env.step(render=True) appends TacView frames (env_base.py:111-151) without disturbing the trajectory."""
    from neuralplane_amd.envs.control_env import ControlEnv
    from neuralplane_amd.envs.utils.acmi import parse_acmi
    a = torch.zeros(3, 4, device='cuda')
    env = ControlEnv(num_envs=3, config='heading', model='F16', random_seed=1, device='cuda:0')
    ref = ControlEnv(num_envs=3, config='heading', model='F16', random_seed=1, device='cuda:0')
    env.reset()
    ref.reset()
    base = str(tmp_path / 'tracks' / 'rec-')
    for k in range(3):
        env._batch.step(a)
        env.render(count=k, filename=base)
        ref.step(a)
    assert torch.equal(env.model.s, ref.model.s)
    header, frames = parse_acmi(open(base + '0.txt.acmi').read())
    assert header[0] == 'FileType=text/acmi/tacview' and len(frames) == 3 and [len(f[1]) for f in frames] == [3, 3, 3]
    assert frames[2][0] == pytest.approx(0.06) and frames[2][1][1][0] == 101
    alt_m = env.model.s[0, 2].item() * 0.3048
    assert abs(frames[2][1][0][1][2] - alt_m) < 1.0


def test_contexts_with_different_numerics_options_coexist():
    """Two live contexts on one device (MLP numerics and aero_1d_tables), stepped alternately with resets in flight:
    each stays bit-identical to its own oracle (the reset coefficients are per context)."""
    n, seed = 400, 3
    bs = [_batch('heading', n, seed=seed, tables=False), _batch('heading', n, seed=seed, tables=True)]
    os_ = [Oracle('heading'), Oracle('heading', mode=MODE_PWL)]
    sts = [Oracle.new_state(n), Oracle.new_state(n)]
    for b, o, st in zip(bs, os_, sts):
        obs = b.reset()
        o_obs = o.reset(st, seed=seed, call_idx=0)
        _check_equal(b, obs, None, b.flags, st, o_obs, None, 'reset')
    rng = np.random.RandomState(1)
    resets = 0
    for t in range(90):
        a = rng.uniform(-2.0, 2.0, (n, 4)).astype(np.float32)
        a[:, 1] = 1.0 if t < 45 else -1.0      # full elevator: drives alpha out of the envelope -> resets in flight
        for b, o, st in zip(bs, os_, sts):
            obs, rew, flags = b.step(torch.from_numpy(a).cuda())
            o_obs, o_rew, _, _, _ = o.step(st, a, seed=seed, call_idx=t + 1)
            _check_equal(b, obs, rew, flags, st, o_obs, o_rew, f'step {t}')
            resets += int(st['bad'].sum())
    assert resets > 0


class _TinyActor(torch.nn.Module):
    """Capturable stand-in with the interface of the reference's PPOActor (obs, rnn_states, masks) -> (actions, _, rnn_states)."""

    def __init__(self):
        super().__init__()
        self.base = torch.nn.Sequential(torch.nn.LayerNorm(22), torch.nn.Linear(22, 64), torch.nn.ReLU())
        self.gru = torch.nn.GRUCell(64, 128)
        self.head = torch.nn.Linear(128, 4)

    def forward(self, obs, rnn, masks, deterministic=True):
        h = self.gru(self.base(obs), rnn[:, 0] * masks)
        return torch.tanh(self.head(h)) * 1.5, None, h.unsqueeze(1)


def test_planning_env_hip_graph_replay_equals_eager():
    """enable_graph(): the 1 + 50 x 2 launches of a PlanningEnv.step (plus the controller) replayed from one HIP graph give
    the eager path's results bit for bit — obs noise included (device-side RNG counter), across eager/graph interleaving
    and a checkpoint restore."""
    from neuralplane_amd.envs.planning_env import PlanningEnv
    n = 300
    torch.manual_seed(0)
    ctrl = _TinyActor().cuda().eval()
    envs = [PlanningEnv(num_envs=n, config='tracking', model='F16', random_seed=4, device='cuda:0', controller=ctrl) for _ in range(2)]
    envs[1].enable_graph()
    g = torch.Generator(device='cpu').manual_seed(1)
    acts = [(torch.rand((n, 3), generator=g) * 2.4 - 1.2).cuda() for _ in range(6)]
    terminated = 0
    for k, a in enumerate(acts):
        if k == 3:       # one eager step in the middle of the graph-mode env, then back
            envs[1].enable_graph(False)
        if k == 4:
            envs[1].enable_graph(True)
        outs = [e.step(a) for e in envs]
        for x, y in zip(outs[0][:5], outs[1][:5]):
            assert torch.equal(x, y), f'macro-step {k}'
        assert torch.equal(envs[0].model.s, envs[1].model.s) and torch.equal(envs[0].model.u, envs[1].model.u)
        assert torch.equal(envs[0].step_count, envs[1].step_count) and torch.equal(envs[0].ego_rnn_states, envs[1].ego_rnn_states)
        terminated += int(outs[0][3].sum())
    assert envs[0]._batch.call_idx == envs[1]._batch.call_idx == 6 * 51
    # the capture's warm-up macro-step must not leak into the per-condition termination statistics
    assert envs[0].termination_counts() == envs[1].termination_counts() and sum(envs[0].termination_counts().values()) > 0
    # a checkpoint carries the controller's recurrent state: restored into the graph-mode env AND into a fresh eager env,
    # the run continues bit for bit (two macro-steps, so that the restored GRU state has been consumed and re-written)
    sd = envs[0].state_dict()
    assert 'ego_rnn_states' in sd
    ref = [envs[0].step(a) for a in acts[:2]]
    fresh = PlanningEnv(num_envs=n, config='tracking', model='F16', random_seed=4, device='cuda:0', controller=ctrl)
    for e in (envs[1], fresh):
        e.load_state_dict(sd)
        got = [e.step(a) for a in acts[:2]]
        for r, g_ in zip(ref, got):
            for x, y in zip(r[:5], g_[:5]):
                assert torch.equal(x, y)
        assert torch.equal(e.ego_rnn_states, envs[0].ego_rnn_states) and torch.equal(e.model.s, envs[0].model.s)
    with pytest.raises(KeyError):
        fresh.load_state_dict(envs[0]._batch.state_dict())      # a bare batch checkpoint lacks the controller state


def test_planning_env_graph_follows_the_optional_output_pointers():
    """termination_reasons() / reward_terms() switched on AFTER enable_graph() has captured: the captured launches baked NULL
    pointers, so the graph is re-captured before its next replay and the per-aircraft outputs equal the eager env's; switched off
    again, the graph no longer writes the freed buffers.  The reason bits accumulate over the 50 inner iterations: every row the
    step flags bad_done shows at least one of the bad conditions."""
    from neuralplane_amd.envs.planning_env import PlanningEnv
    n = 300
    torch.manual_seed(0)
    ctrl = _TinyActor().cuda().eval()
    envs = [PlanningEnv(num_envs=n, config='tracking', model='F16', random_seed=9, device='cuda:0', controller=ctrl) for _ in range(2)]
    envs[1].enable_graph()
    g = torch.Generator(device='cpu').manual_seed(3)
    acts = [(torch.rand((n, 3), generator=g) * 2.4 - 1.2).cuda() for _ in range(5)]
    for e in envs:
        e.step(acts[0])                                      # the graph is captured here, without the optional outputs
    first_graph = envs[1]._graph
    reasons = [e.termination_reasons() for e in envs]        # tracking on: a new pointer
    terms = [e.reward_terms()[0] for e in envs]
    n_bad = 0
    for a in acts[1:4]:
        outs = [e.step(a) for e in envs]
        for x, y in zip(outs[0][:5], outs[1][:5]):
            assert torch.equal(x, y)
        r0, r1 = envs[0].termination_reasons(), envs[1].termination_reasons()
        assert torch.equal(r0, r1), 'graph replay did not write the per-aircraft condition bits'
        assert torch.equal(envs[0].reward_terms()[0], envs[1].reward_terms()[0])
        bad = outs[0][3]
        n_bad += int(bad.sum())
        assert bool(((r0[bad] & 0x3F) != 0).all()), 'a row flagged bad_done shows no bad condition: the bits did not accumulate over the inner steps'
        assert bool((r0[~bad & ~outs[0][2]] & 0x3F).eq(0).all())
    assert envs[1]._graph is not first_graph, 'the graph was not re-captured after the output pointers changed'
    assert n_bad > 0
    second_graph = envs[1]._graph
    envs[1]._batch.track_termination_reasons(False)          # pointer dropped: the next replay must come from a fresh capture
    envs[0]._batch.track_termination_reasons(False)
    outs = [e.step(acts[4]) for e in envs]
    for x, y in zip(outs[0][:5], outs[1][:5]):
        assert torch.equal(x, y)
    assert envs[1]._graph is not second_graph
    del reasons, terms


def test_hip_parity_report_vs_reference_recordings():
    """The headline parity claim, directly, and the artefact SURVEY.md §8(d) asks for: the HIP env against what the REFERENCE
    recorded (tests/golden/traj_*.npz, plain ATen arithmetic; the authors' CUDA episode) — tools/parity_report.py with the HIP
    engine.  Heading N = 256 x 1000 steps open loop (7254 resets: the episode schedule) and closed loop (no reset: 1000
    uninterrupted steps), Control / Tracking 300 steps.  Bounds are the measured ones: MAX over aircraft < 1e-4 at every
    reported step, p99 < 5e-5.  The report is written to gpurun_out/ (copied to profiles/ by the builder)."""
    import json
    import os
    from test_oracle_golden import check_parity_rows
    from tools.parity_report import build
    rep = build('hip')
    for tr in rep['trajectories']:
        check_parity_rows(tr, tr['n'])
    from test_oracle_golden import check_done_chain
    for tr in rep['trajectories_with_done_events']:      # flown by the reference's PID stack: `done` fires, rows re-initialise mid-trajectory
        check_parity_rows(tr, tr['n'], p99=1e-4, median=5e-5)      # episodes of up to 2 500 uninterrupted steps (test_oracle_golden.py)
        check_done_chain(tr, tr['n'])
    cl = rep['closed_loop']
    assert cl['resets_in_reference'] == 0
    check_parity_rows(cl, cl['n'])
    assert cl['at'][-1]['t'] == 1000 and cl['at'][-1]['max'] < 2e-5
    ep = {r['t']: r for r in rep['recorded_episode']['at']}
    # the recording ends in a departure: measured envelope 5.1e-6 @200, 8.2e-5 @400, 1.76e-4 @426 — asserted as measured; the reference's
    # own CPU replay ends at 2.4e-5 plain and at 1.80e-4 with its MLPs in fp64 (tests/golden/recorded_episode0_ref_cpu.npz, in the report)
    assert ep[200]['max_so_far'] < 1e-5 and ep[400]['max_so_far'] < 1e-4 and ep[426]['max_so_far'] < 2e-4
    att = rep['recorded_episode']['attribution']
    assert att['oracle_pin_mode_equals_reference_cpu_pin_mode_bit_for_bit'] and att['reference_cpu_vs_cuda_recording'] < 3e-5
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'parity_hip.json'), 'w') as f:
            json.dump(rep, f, indent=1)
    except OSError:
        pass


def test_pinned_vec_env_equals_gpu_vec_env():
    """PinnedVecEnv (page-locked staging, one sync per step, ring of host buffers) returns what GPUVecEnv returns."""
    from neuralplane_amd.envs.control_env import ControlEnv
    from neuralplane_amd.envs.env_wrappers import GPUVecEnv, PinnedVecEnv
    mk = lambda: ControlEnv(num_envs=500, config='heading', model='F16', random_seed=2, device='cuda:0')  # noqa: E731
    a, b = GPUVecEnv([mk]), PinnedVecEnv([mk], ring=2)
    oa, ob = a.reset(), b.reset()
    assert isinstance(ob, np.ndarray) and ob.shape == (500, 1, 22) and np.array_equal(oa, ob)
    rng = np.random.RandomState(3)
    prev = None
    for k in range(6):
        act = rng.uniform(-1.5, 1.5, (500, 1, 4)).astype(np.float32)
        ra, rb = a.step(act), b.step(act)
        for x, y in zip(ra[:5], rb[:5]):
            assert x.shape == y.shape and x.dtype == y.dtype and np.array_equal(x, y)
        if prev is not None:      # ring = 2: the arrays of the previous step are still intact
            assert np.array_equal(prev[0], prev[1])
        prev = (ra[0].copy(), rb[0])


def test_drop_in_example_runs_against_the_reference_import_paths():
    """examples/drop_in_rollout.py: host code written against `from envs.control_env import ControlEnv` /
    `from envs.env_wrappers import GPUVecEnv` runs unchanged once `envs` is aliased."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'examples', 'drop_in_rollout.py'), '512', '40'], cwd=root,
                       capture_output=True, text=True, timeout=300, env=dict(os.environ, PYTHONPATH=root))
    assert r.returncode == 0, r.stderr[-2000:]
    assert 'numpy VecEnv loop' in r.stdout and 'device-resident loop' in r.stdout and 'termination statistics' in r.stdout


def _second_aircraft_blob(path):
    """A second aero set with the F-16 topology (what the kernel template is parameterised by): every coefficient's
    un-normalisation (out_mean, out_std of the 128-byte blob records) is changed, weights and PWL tables stay valid."""
    import struct
    from neuralplane_amd.core import ASSET_BLOB
    blob = bytearray(open(ASSET_BLOB, 'rb').read())
    for k in range(43):
        off = 16 + 128 * k + 104
        mean, std = struct.unpack_from('<dd', blob, off)
        struct.pack_into('<dd', blob, off, mean + 0.013 * std * ((k % 5) - 2), std * (0.85 + 0.01 * (k % 7)))
    open(path, 'wb').write(bytes(blob))
    return path


@pytest.mark.parametrize('tables', [False, True], ids=['mlp', 'aero_1d_tables'])
def test_two_aircraft_types_coexist_on_one_device(tmp_path, tables):
    """Per-context weights: an F-16 context and a context built from another blob with the same net topology run side by
    side (alternating launches); each is bit-identical to the oracle loaded with ITS blob, and they differ from each other."""
    from neuralplane_amd.core import F16Batch
    from neuralplane_amd.envs.utils.utils import parse_config
    other = _second_aircraft_blob(str(tmp_path / 'second_aero_set.bin'))
    n, seed = 500, 8
    cfg = parse_config('control')
    bs = [F16Batch(n, cfg, 'control', 'cuda:0', seed=seed, aero_1d_tables=tables),
          F16Batch(n, cfg, 'control', 'cuda:0', seed=seed, aero_1d_tables=tables, blob_path=other)]
    os_ = [Oracle('control', mode=MODE_PWL if tables else 0), Oracle('control', mode=MODE_PWL if tables else 0, blob_path=other)]
    sts = [Oracle.new_state(n), Oracle.new_state(n)]
    rng = np.random.RandomState(2)
    for t in range(30):
        a = rng.uniform(-1.2, 1.2, (n, 4)).astype(np.float32)
        for b, o, st in zip(bs, os_, sts):
            obs, rew, flags = b.step(torch.from_numpy(a).cuda())
            o_obs, o_rew, _, _, _ = o.step(st, a, seed=seed, call_idx=t)
            _check_equal(b, obs, rew, flags, st, o_obs, o_rew, f'step {t}')
    assert not np.array_equal(sts[0]['s'], sts[1]['s'])


# another aircraft: every field of np_f16_airframe moved off the F-16's value (mass + 25 %, other inertias incl. a non-zero engine angular
# momentum, smaller wing, c.g. further aft, other control / command scales and atmosphere constants) — physically plausible enough to fly
SECOND_AIRFRAME = dict(g=32.174, mass=800.0, B=34.5, S=345.0, cbar=10.1, xcgr=0.33, xcg=0.27, Heng=160.0, Jy=61000.0, Jxz=1100.0, Jz=70500.0,
                       Jx=11000.0, ail_ref=20.0, rud_ref=28.0, atm_lapse=0.69e-5, atm_exp=4.2, rho0=2.4e-3, lag_keep=0.88, lag_new=0.12,
                       thrust_frac=0.2, thrust_max=90000.0, thrust_unit=0.3, surf_max=(40.0, 42.0, 47.0))


@pytest.mark.parametrize('task,solver,variant,n', [('heading', None, 'latency8', 300), ('heading', None, 'latency', 700), ('control', None, 'latency2', 900),
                                                   ('tracking', None, 'throughput', 500), ('heading', None, 'pair', 150_000), ('control', 'rk4', 'pair', 600),
                                                   ('heading', None, 'latency4w', 1000)])
def test_a_second_airframe_is_data_and_runs_hip_equal_to_the_oracle(tmp_path, task, solver, variant, n):
    """the airframe is DATA (np_f16_cfg.airframe, ABI 16) — mass, inertias, S, B, cbar, c.g., H_eng, g, control and command
    scales, atmosphere constants — not literals in the kernel.  (a) A context whose block spells out the F-16 values is bit-identical to the
    all-zero block (= what the kernels computed when these were compile-time literals: every pinned fixture and parity test of this suite runs
    through the same loads).  (b) A context built from ANOTHER blob and ANOTHER airframe block runs HIP == oracle bit for bit — state, controls,
    targets, masks, observation, reward, the derived getters — on every kernel variant, and differs from the F-16.  Parity of a non-F-16
    aircraft against the reference is unpinned by construction: the reference has none."""
    from neuralplane_amd.core import F16Batch
    from neuralplane_amd.envs.utils.utils import parse_config
    other = _second_aircraft_blob(str(tmp_path / 'second_aero_set.bin'))
    seed = 8
    f16_spelled = dict(g=32.17, mass=636.94, B=30.0, S=300.0, cbar=11.32, xcgr=0.35, xcg=0.30, Heng=0.0, Jy=55814.0, Jxz=982.0, Jz=63100.0, Jx=9496.0)

    def mk(airframe, blob=None):
        cfg = parse_config(task)
        if airframe:
            cfg.airframe = airframe
        b = F16Batch(n, cfg, task, 'cuda:0', seed=seed, solver=solver, **({'blob_path': blob} if blob else {}))
        b.set_kernel_variant(variant)
        return b
    bs = [mk(None), mk(f16_spelled), mk(SECOND_AIRFRAME, other)]
    o2 = Oracle(task, solver=solver, overrides={'airframe': SECOND_AIRFRAME}, blob_path=other)
    o1 = Oracle(task, solver=solver)
    st1, st2 = Oracle.new_state(n), Oracle.new_state(n)
    rng = np.random.RandomState(2)
    steps = 25 if n < 10_000 else 6
    for t in range(steps):
        a = rng.uniform(-1.2, 1.2, (n, 4)).astype(np.float32)
        outs = [b.step(torch.from_numpy(a).cuda()) for b in bs]
        o_obs, o_rew, _, _, _ = o1.step(st1, a, seed=seed, call_idx=t)
        _check_equal(bs[0], *outs[0], st1, o_obs, o_rew, f'F-16 step {t}')
        for x, y in zip(outs[0], outs[1]):
            assert torch.equal(x, y), f'step {t}: the spelled-out F-16 block differs from the all-zero block'
        assert torch.equal(bs[0].s, bs[1].s) and torch.equal(bs[0].u, bs[1].u)
        o_obs, o_rew, _, _, _ = o2.step(st2, a, seed=seed, call_idx=t)
        _check_equal(bs[2], *outs[2], st2, o_obs, o_rew, f'second aircraft step {t}')
    assert not np.array_equal(st1['s'], st2['s'])
    with pytest.raises(ValueError, match='airframe'):        # a checkpoint continues on the aircraft it was written with
        bs[0].load_state_dict(bs[2].state_dict())
    bs[0].load_state_dict(bs[1].state_dict())                # (the all-zero block and the spelled-out F-16 values are the same aircraft)
    d = bs[2].derived().cpu().numpy()                       # the getters that need the dynamics / atmosphere read the same block
    x = np.hstack([st2['s'], st2['u']]).astype(np.float32)
    assert _same(d[:12].T, o2.nlplant(x)) and _same(d[18], o2.get_eas2tas(st2['s'])) and _same(d[20:23].T, o2.get_atmos(st2['s']))


@pytest.mark.parametrize('seed', range(6))
def test_random_airframe_blocks_hip_equals_the_oracle(seed):
    """"HIP == oracle for ANY airframe block": six random blocks (every field drawn around the F-16's value: masses and inertias x 0.5 .. 2, areas
    and lengths x 0.7 .. 1.4, c.g. positions, engine angular momentum, control / command scales, atmosphere constants), random task, kernel variant
    and solver each — state, controls, targets, masks, observation and reward bit for bit over 20 steps with auto-resets."""
    from neuralplane_amd.core import F16Batch
    from neuralplane_amd.envs.utils.utils import parse_config
    rng = np.random.RandomState(100 + seed)
    f = lambda lo, hi: float(rng.uniform(lo, hi))   # noqa: E731
    Jx, Jz = 9496.0 * f(0.5, 2), 63100.0 * f(0.5, 2)
    af = dict(g=32.17 * f(0.9, 1.1), mass=636.94 * f(0.5, 2), B=30.0 * f(0.7, 1.4), S=300.0 * f(0.7, 1.4), cbar=11.32 * f(0.7, 1.4), xcgr=f(0.25, 0.4), xcg=f(0.2, 0.4),
              Heng=f(0, 300), Jy=55814.0 * f(0.5, 2), Jxz=f(-0.3, 0.3) * (Jx * Jz) ** 0.5, Jz=Jz, Jx=Jx, ail_ref=21.5 * f(0.8, 1.2), rud_ref=30.0 * f(0.8, 1.2),
              atm_lapse=0.703e-5 * f(0.9, 1.1), atm_exp=4.14 * f(0.9, 1.1), rho0=2.377e-3 * f(0.9, 1.1), lag_keep=f(0.8, 0.95), lag_new=f(0.05, 0.2),
              thrust_frac=0.225 * f(0.8, 1.2), thrust_max=76300.0 * f(0.6, 1.5), thrust_unit=0.3048 * f(0.9, 1.1), surf_max=(45.0 * f(0.8, 1.1), 45.0 * f(0.8, 1.1), 45.0 * f(0.8, 1.1)))
    task = ('heading', 'control', 'tracking')[seed % 3]
    solver = 'rk4' if seed == 4 else None
    variant, n = [('latency8', 200), ('latency', 900), ('pair', 140_000), ('latency2', 1500), ('pair', 700), ('throughput', 400)][seed]
    cfg = parse_config(task)
    cfg.airframe = af
    b = F16Batch(n, cfg, task, 'cuda:0', seed=seed, solver=solver)
    b.set_kernel_variant(variant)
    o, st = Oracle(task, solver=solver, overrides={'airframe': af}), Oracle.new_state(n)
    for t in range(20 if n < 10_000 else 5):
        a = rng.uniform(-1.2, 1.2, (n, 4)).astype(np.float32)
        obs, rew, flags = b.step(torch.from_numpy(a).cuda())
        o_obs, o_rew, _, _, _ = o.step(st, a, seed=seed, call_idx=t)
        _check_equal(b, obs, rew, flags, st, o_obs, o_rew, f'airframe seed {seed} step {t}')


def test_a_second_airframe_in_single_combat_and_planning_env_hip_equals_the_oracle(tmp_path):
    """The same block through the other two kernels that integrate the FDM: SingleCombatEnv (np_f16_combat_cfg.airframe) and PlanningEnv's
    persistent kernel (the env record's cfg) — HIP == oracle bit for bit with the second airframe."""
    from neuralplane_amd.actor import NUM_FLOATS, FusedActor
    from neuralplane_amd.envs.planning_env import PlanningEnv
    from neuralplane_amd.envs.singlecombat_env import SingleCombatEnv
    from oracle.f16_oracle import ActorOracle, CombatOracle
    seed, E = 4, 300
    cenv = SingleCombatEnv(num_envs=E, config='selfplay', random_seed=seed, device='cuda:0', airframe=SECOND_AIRFRAME)
    co = CombatOracle(overrides={'airframe': SECOND_AIRFRAME})
    cst = co.new_state(E)
    assert _same(cenv.reset().cpu().numpy(), co.combat_reset(cst, seed=seed, call_idx=0))
    rng = np.random.RandomState(1)
    for t in range(6):
        a = rng.uniform(-1, 1, (2 * E, 4)).astype(np.float32)
        obs, rew, done, bad, tmo, _ = cenv.step(torch.from_numpy(a).cuda())
        o_obs, o_rew, _, o_bad, _ = co.combat_step(cst, a, pid_first=(t == 0), seed=seed, call_idx=t + 1)
        assert _same(cenv.s.cpu().numpy(), cst['s']) and _same(obs.cpu().numpy(), o_obs) and _same(rew.cpu().numpy(), o_rew), f'combat step {t}'
    ref = CombatOracle()
    rst = ref.new_state(E)
    ref.combat_reset(rst, seed=seed, call_idx=0)
    ref.combat_step(rst, np.zeros((2 * E, 4), np.float32), pid_first=True, seed=seed, call_idx=1)
    P = 70
    w = np.random.RandomState(3).normal(0, 0.08, NUM_FLOATS).astype(np.float32)
    for numerics in ('i8', 'fp32'):
        penv = PlanningEnv(num_envs=P, config='tracking', model='F16', random_seed=seed, device='cuda:0', controller=FusedActor(w, 'cuda:0', numerics=numerics),
                           airframe=SECOND_AIRFRAME)
        po, pa, pst, ph = Oracle('tracking', overrides={'airframe': SECOND_AIRFRAME}), ActorOracle(w, numerics), Oracle.new_state(P), np.zeros((P, 128), np.float32)
        for k in range(2):
            ru = rng.uniform(0, 1, (P, 5)).astype(np.float32)
            hi = rng.uniform(-1, 1, (P, 3)).astype(np.float32)
            penv._batch.reset(rand_u=ru, want_obs=False)
            obs, rew, done, bad, tmo, _ = penv.step(torch.from_numpy(hi).cuda())
            po.reset(pst, rand_u=ru, want_obs=False)
            ac = np.clip(hi, -1, 1).astype(np.float32)
            s = pst['s']
            tgt3 = np.stack([s[:, 4] + ac[:, 0] * np.float32(0.3), s[:, 5] + ac[:, 1] * np.float32(0.3), s[:, 6] + ac[:, 2] * np.float32(30)], 1).astype(np.float32)
            ones = np.ones(P, np.float32)
            for i in range(50):
                act, ph = pa.forward(po.lowlevel_obs(pst, tgt3), ph, ones)
                o_obs, o_rew, d, b_, t_ = po.step_inner(pst, act)
            assert _same(penv.model.s.cpu().numpy(), pst['s']) and _same(obs.cpu().numpy(), o_obs) and _same(rew.cpu().numpy(), o_rew), f'planning {numerics} macro-step {k}'
            assert np.array_equal(bad.cpu().numpy(), b_.astype(bool))


def test_c_abi_from_a_plain_cpp_host_matches_the_python_surface(tmp_path):
    """examples/c_abi_demo.cpp: a C++ program with no Python and no PyTorch links libneuralplane_hip.so, allocates with
    hipMalloc, and drives np_f16_ctx_create / np_f16_reset / np_f16_step through the public header.  Its final state must be
    bit-identical to the same run through the Python env surface."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / 'c_abi_demo')
    csrc = os.path.join(root, 'neuralplane_amd', 'csrc')
    cmd = ['g++', '-O2', '-std=c++17', '-D__HIP_PLATFORM_AMD__', '-I/opt/rocm/include', '-I', os.path.join(root, 'include'),
           os.path.join(root, 'examples', 'c_abi_demo.cpp'), '-o', exe, '-L' + csrc, '-lneuralplane_hip', '-L/opt/rocm/lib', '-lamdhip64',
           '-Wl,-rpath,' + csrc, '-Wl,-rpath,/opt/rocm/lib']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    n, steps = 1000, 30
    out = str(tmp_path / 'state.bin')
    r = subprocess.run([exe, os.path.join(root, 'neuralplane_amd', 'assets', 'f16_aero_mlp.bin'), str(n), str(steps), out],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'C_ABI_DEMO' in r.stdout, (r.stdout, r.stderr[-2000:])
    s_c = np.fromfile(out, dtype=np.float32).reshape(12, n)
    b = _batch('heading', n, seed=42)
    b.reset()
    i = np.arange(n)
    for t in range(steps):
        a = np.stack([0.5 + 0.25 * ((i + t) % 3), 0.125 * ((i + 2 * t) % 5) - 0.25, 0.0625 * ((i * 3 + t) % 7) - 0.1875,
                      0.03125 * ((i + 5 * t) % 9) - 0.125], axis=1).astype(np.float32)
        b.step(torch.from_numpy(a).cuda())
    assert np.array_equal(b.s.cpu().numpy(), s_c)
