"""GPU edge cases through the C ABI, HIP == oracle bit for bit on both kernel variants:

* ragged and degenerate batch sizes (1, 2, 63, 64, 65, 127, 129, 257) and the empty batch;
* hostile inputs: NaN / +-inf / huge actions, states outside every envelope (vt ~ 0, negative altitude, alpha far out of
  range, NaN and inf state components) — the numerics spec's "non-finite inputs poison every coefficient" rule and the
  comparison semantics of the termination conditions (NaN compares false) must agree lane by lane;
* scenario constants far from the shipped YAMLs (dt, airspeed offset, limits, check intervals, reset ranges, noise scale).
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.f16_oracle import Oracle  # noqa: E402  (the checker; test infrastructure)

VARIANTS = ['latency', 'latency4w', 'latency8', 'latency2', 'throughput', 'pair']


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return bool(np.all((a == b) | (np.isnan(a.astype(np.float64)) & np.isnan(b.astype(np.float64)))))


def _mk(task, n, variant, overrides=None, seed=0, solver=None):
    from neuralplane_amd.core import F16Batch
    from neuralplane_amd.envs.utils.utils import parse_config
    cfg = parse_config(task)
    for k, v in (overrides or {}).items():
        setattr(cfg, k, v)
    b = F16Batch(n, cfg, task, 'cuda:0', seed=seed, solver=solver)
    b.set_kernel_variant(variant)
    return b, Oracle(task, solver=solver, overrides=overrides)


def _load(b, st):
    b.s.copy_(torch.from_numpy(st['s'].T.copy()))
    b.u.copy_(torch.from_numpy(st['u'].T.copy()))
    b.tgt.copy_(torch.from_numpy(st['tgt'].T.copy()))
    b.step_count.copy_(torch.from_numpy(st['step_count']))
    b.flags.copy_(torch.from_numpy(np.stack([st['done'], st['bad'], st['timeout']])))


def _check(b, obs, rew, flags, st, o_obs, o_rew, what):
    assert _same(b.s.cpu().numpy().T, st['s']), f'{what}: state'
    assert _same(b.u.cpu().numpy().T, st['u']), f'{what}: controls'
    assert _same(b.tgt.cpu().numpy().T, st['tgt']), f'{what}: targets'
    assert np.array_equal(b.step_count.cpu().numpy(), st['step_count']), f'{what}: step_count'
    f = flags.cpu().numpy()
    assert np.array_equal(f[0], st['done']) and np.array_equal(f[1], st['bad']) and np.array_equal(f[2], st['timeout']), f'{what}: masks'
    assert _same(obs.cpu().numpy(), o_obs), f'{what}: obs'
    assert _same(rew.cpu().numpy(), o_rew), f'{what}: reward'


@pytest.mark.parametrize('variant', VARIANTS)
@pytest.mark.parametrize('n', [1, 2, 63, 64, 65, 127, 129, 257])
def test_ragged_batch_sizes(n, variant):
    b, o = _mk('heading', n, variant, seed=5)
    st = Oracle.new_state(n)
    rng = np.random.RandomState(n)
    for t in range(12):
        a = rng.uniform(-1.2, 1.2, (n, 4)).astype(np.float32)
        obs, rew, flags = b.step(torch.from_numpy(a).cuda())
        o_obs, o_rew, _, _, _ = o.step(st, a, seed=5, call_idx=t)
        _check(b, obs, rew, flags, st, o_obs, o_rew, f'n={n} step {t}')


def test_empty_batch_is_a_no_op():
    from neuralplane_amd import _lib
    b, _ = _mk('heading', 4, 'auto')
    io = b._io(torch.empty((3, 4), dtype=torch.uint8, device='cuda'), torch.zeros((4, 4), device='cuda'),
               torch.empty((4, 22), device='cuda'), torch.empty(4, device='cuda'), None, None)
    before = b.s.clone()
    assert b.lib.np_f16_step(b._ctx, 0, C.byref(io), None) == 0      # n = 0: success, nothing launched
    assert b.lib.np_f16_reset(b._ctx, 0, C.byref(io), None) == 0
    torch.cuda.synchronize()
    assert torch.equal(before, b.s)
    assert b.lib.np_f16_step(b._ctx, 8, C.byref(io), None) != 0      # ld (4) < n (8): rejected, not launched
    assert b'ld' in _lib.load().np_last_error()


@pytest.mark.parametrize('variant', VARIANTS)
@pytest.mark.parametrize('task', ['heading', 'control', 'tracking'])
def test_hostile_actions_and_states(task, variant):
    n = 192
    b, o = _mk(task, n, variant, seed=9)
    st = Oracle.new_state(n)
    rng = np.random.RandomState(17)
    o.reset(st, seed=9, call_idx=0)
    b.reset()
    s = st['s']
    # rows 0..95: poisoned / out-of-envelope states (the same edit on both sides)
    s[0, 6] = 0.0                     # vt = 0 -> the 0.01 clamp of nlplant, division by vt outside it
    s[1, 6] = 0.005
    s[2, 2] = -500.0                  # below ground
    s[3, 7] = 2.5                     # alpha = 143 deg: MLP inputs far outside the fitted range
    s[4, 8] = -1.9
    s[5, 4] = np.float32(np.pi / 2)   # pitch = 90 deg: tan / 1/cos singular
    s[6, 2] = np.nan
    s[7, 6] = np.inf
    s[8, 9] = -np.inf
    s[9, 7] = np.nan
    s[10, 3] = 1.0e9                  # huge roll angle: the >= 2^30 branch of the trig reduction
    s[11, 5] = -3.0e10
    s[12, 2] = 160000.0               # tfac < 0: pow of a negative base -> NaN density
    s[13, 6] = 1.0e6
    s[16:96, 3:12] = rng.normal(0, 1.5, (80, 9)).astype(np.float32)
    s[16:96, 6] = rng.uniform(1, 3000, 80).astype(np.float32)
    st['u'][20:40, 1:4] = rng.uniform(-200, 200, (20, 3)).astype(np.float32)
    st['step_count'][40:60] = rng.randint(250, 2600, 20)
    _load(b, st)
    for t in range(6):
        a = rng.uniform(-1.5, 1.5, (n, 4)).astype(np.float32)
        a[100, 0] = np.nan
        a[101, 1] = np.inf
        a[102, 2] = -np.inf
        a[103] = [1e30, -1e30, 1e-40, -0.0]
        a[104:110] = 0.0
        obs, rew, flags = b.step(torch.from_numpy(a).cuda())
        o_obs, o_rew, _, _, _ = o.step(st, a, seed=9, call_idx=t + 1)
        _check(b, obs, rew, flags, st, o_obs, o_rew, f'{task} step {t}')
    assert np.isnan(st['s']).any() or st['bad'].any()


OVERRIDES = [
    {'dt': 0.005, 'airspeed': 35.0, 'noise_scale': 0.2, 'altitude_limit': 18000.0, 'max_velocity': 1.05, 'min_velocity': 0.9},
    {'dt': 0.05, 'acceleration_limit': 40.0, 'min_alpha': -2, 'max_alpha': 6, 'min_beta': -1, 'max_beta': 1, 'noise_scale': 0.0},
    {'max_check_interval': 7, 'min_check_interval': 3, 'max_altitude': 40000, 'min_altitude': 3000, 'max_vt': 2500, 'min_vt': 200,
     'max_heading_increment': 3.0, 'max_pitch_increment': 1.0, 'max_velocities_u_increment': 500, 'max_distance': 9000, 'min_distance': 10},
]


@pytest.mark.parametrize('variant', VARIANTS)
@pytest.mark.parametrize('task', ['heading', 'control', 'tracking'])
@pytest.mark.parametrize('ov', range(len(OVERRIDES)))
def test_scenario_constants_far_from_the_shipped_yaml(task, ov, variant):
    overrides = dict(OVERRIDES[ov], init_state={'init_T': 3333.0})
    n = 150
    b, o = _mk(task, n, variant, overrides=overrides, seed=21)
    st = Oracle.new_state(n)
    rng = np.random.RandomState(ov)
    for t in range(15):
        a = rng.uniform(-1.3, 1.3, (n, 4)).astype(np.float32)
        obs, rew, flags = b.step(torch.from_numpy(a).cuda())
        o_obs, o_rew, _, _, _ = o.step(st, a, seed=21, call_idx=t)
        _check(b, obs, rew, flags, st, o_obs, o_rew, f'{task} overrides {ov} step {t}')


@pytest.mark.parametrize('variant', VARIANTS)
@pytest.mark.parametrize('task', ['heading', 'control', 'tracking'])
def test_termination_counters_match_per_condition_sums(task, variant):
    """np_f16_io.term_counters (wave ballot + popcount + one atomic per wave) == the per-condition sums the reference
    prints, computed by the oracle on the same states — over a run with plenty of terminations of every kind."""
    n, seed = 700, 6
    b, o = _mk(task, n, variant, seed=seed, overrides={'max_check_interval': 30, 'min_check_interval': 5})
    st = Oracle.new_state(n)
    rng = np.random.RandomState(1)
    expect = np.zeros(7, np.int64)
    seen = np.zeros(n, np.uint8)
    for t in range(70):
        a = rng.uniform(-1.0, 1.0, (n, 4)).astype(np.float32)
        a[:, 1] = 1.0 if (t // 20) % 2 == 0 else -1.0
        if t == 10:
            reasons = b.track_termination_reasons(True)      # np_f16_io.term_reasons: the same conditions per aircraft, from here on
        obs, rew, flags = b.step(torch.from_numpy(a).cuda())
        o_obs, o_rew, _, _, _ = o.step(st, a, seed=seed, call_idx=t)
        r = o.termination_reasons(st)
        expect += np.array([int(((r >> k) & 1).sum()) for k in range(7)])
        if t >= 10:
            assert np.array_equal(reasons.cpu().numpy(), r), f'per-aircraft condition bits differ from the oracle at step {t}'
            seen |= r
    assert sum(1 for k in range(7) if ((seen >> k) & 1).any()) >= 3
    _check(b, obs, rew, flags, st, o_obs, o_rew, 'final step')
    got = b.termination_counts()
    assert list(got.values()) == expect.tolist(), (got, expect)
    assert sum(1 for v in got.values() if v > 0) >= 3
    assert b.termination_counts(reset=True) == got and sum(b.termination_counts().values()) == 0


def _random_overrides(rng):
    """Scenario constants drawn far around the shipped YAML values (every key the kernel's constant block is derived from)."""
    lo_alt = float(rng.uniform(500, 6000))
    lo_vt = float(rng.uniform(150, 600))
    return {
        'dt': float(rng.choice([0.005, 0.01, 0.02, 0.04])), 'airspeed': float(rng.uniform(0, 60)), 'noise_scale': float(rng.choice([0.0, 0.01, 0.3])),
        'altitude_limit': float(rng.uniform(1000, 9000)), 'acceleration_limit': float(rng.uniform(20, 400)),
        'max_velocity': float(rng.uniform(1.0, 4.0)), 'min_velocity': float(rng.uniform(0.005, 0.6)),
        'min_alpha': float(rng.uniform(-30, -3)), 'max_alpha': float(rng.uniform(5, 60)), 'min_beta': float(rng.uniform(-40, -2)),
        'max_beta': float(rng.uniform(2, 40)), 'max_check_interval': int(rng.randint(8, 60)), 'min_check_interval': int(rng.randint(2, 8)),
        'min_altitude': lo_alt, 'max_altitude': lo_alt + float(rng.uniform(100, 30000)), 'min_vt': lo_vt, 'max_vt': lo_vt + float(rng.uniform(10, 2000)),
        'max_heading_increment': float(rng.uniform(0.1, 3.1)), 'max_altitude_increment': float(rng.uniform(10, 5000)),
        'max_velocities_u_increment': float(rng.uniform(1, 300)), 'max_pitch_increment': float(rng.uniform(0.05, 1.5)),
        'max_distance': float(rng.uniform(1000, 20000)), 'min_distance': float(rng.uniform(1, 900)),
        'init_state': {'init_T': float(rng.uniform(0, 20000))},
    }


@pytest.mark.parametrize('seed', range(8))
@pytest.mark.parametrize('task', ['heading', 'control', 'tracking'])
def test_randomised_scenario_constants(task, seed):
    """A seeded sweep over the whole configuration surface: whatever constants the YAML carries, the kernel's constant block and
    the oracle's must lead to the same bits (both kernel variants alternate by seed)."""
    rng = np.random.RandomState(1000 + seed)
    overrides = _random_overrides(rng)
    n = 130
    b, o = _mk(task, n, VARIANTS[seed % 4], overrides=overrides, seed=seed)
    st = Oracle.new_state(n)
    for t in range(14):
        a = rng.uniform(-1.4, 1.4, (n, 4)).astype(np.float32)
        obs, rew, flags = b.step(torch.from_numpy(a).cuda())
        o_obs, o_rew, _, _, _ = o.step(st, a, seed=seed, call_idx=t)
        _check(b, obs, rew, flags, st, o_obs, o_rew, f'{task} random overrides {seed} step {t}')


@pytest.mark.parametrize('task', ['heading', 'control'])
def test_heading_wrap_is_the_exact_remainder_for_every_magnitude(task):
    """np_wrap_pi (torch.remainder semantics, envs/utils/utils.py:144-154) on the device against the oracle's: yaw and pitch errors
    that are exact multiples of fp32(2 pi), one ulp around them, around +-pi, signed zeros, denormals, huge, infinite and NaN —
    observation, reward and masks bit for bit.  (A hand-written remainder replacing the library fmodf was built and measured:
    no gain, dropped.)"""
    n = 4096
    b, o = _mk(task, n, 'pair', seed=3, overrides={'noise_scale': 0.0})
    st = Oracle.new_state(n)
    o.reset(st, seed=3, call_idx=0)
    b.reset()
    c = np.float32(6.28318530717958647692)
    rng = np.random.RandomState(2)
    k = rng.randint(-160000, 160000, n).astype(np.float64)
    x = (k * np.float64(c)).astype(np.float32)                                # near-multiples of 2 pi up to ~1e6
    x[0::4] = np.nextafter(x[0::4], np.float32(np.inf))
    x[1::4] = np.nextafter(x[1::4], np.float32(-np.inf))
    x[2::8] += np.float32(np.pi)
    special = np.array([0.0, -0.0, c, -c, 2 * c, np.pi, -np.pi, np.nextafter(np.float32(np.pi), np.float32(4)), 1e-42, -1e-42, 1048575.94,
                        1048576.0, 1048576.1, -1048576.0, 3e7, -3e7, 1e30, -1e30, np.inf, -np.inf, np.nan, 1.17549435e-38], np.float32)
    x[:special.size] = special
    x[special.size:special.size + 1000] = rng.uniform(-20, 20, 1000).astype(np.float32)
    st['s'][:, 5] = x                      # yaw
    st['tgt'][:, 1] = 0.0                  # target heading 0: the wrapped error is the wrapped yaw (after one Euler step of yaw rate)
    if task == 'control':
        st['s'][:, 4] = np.roll(x, 7) * np.float32(1e-3)   # pitch error through the same wrap, small enough to keep cos(theta) sane
        st['tgt'][:, 0] = 0.0
    _load(b, st)
    for t in range(2):
        a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
        obs, rew, flags = b.step(torch.from_numpy(a).cuda())
        o_obs, o_rew, _, _, _ = o.step(st, a, seed=3, call_idx=t + 1)
        _check(b, obs, rew, flags, st, o_obs, o_rew, f'{task} wrap step {t}')


def test_constant_division_sequence_equals_ieee_division_on_this_device_for_all_floats():
    """np_selfcheck_divc: for every constant the path divides by (the same list the CPU proof uses), all 2^32 bit patterns of x —
    the device's np_divc equals the IEEE quotient wherever the quotient is a normal number and |x| >= 2^-100; what differs
    (denormal quotients, tiny x) is counted separately and stays a vanishing fraction."""
    from neuralplane_amd import _lib
    from test_oracle_golden import _divisor_constants
    lib = _lib.load()
    total_soft, rows = 0, []
    for c in _divisor_constants():
        cnt = (C.c_uint64 * 3)()
        _lib.check(lib.np_selfcheck_divc(C.c_float(c), cnt, 0))
        assert cnt[2] == 2 ** 32, c
        assert cnt[0] == 0, (c, cnt[0])
        total_soft += cnt[1]
        rows.append({'c': c, 'mismatch_normal_range': int(cnt[0]), 'mismatch_tiny_or_denormal': int(cnt[1]), 'inputs': int(cnt[2])})
        assert cnt[1] < 2 ** 32 // 50, (c, cnt[1])     # only sub-2^-100 inputs / denormal quotients (< 2 % of all bit patterns)
    import json
    import os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        json.dump(rows, open(os.path.join(out, 'divc_selfcheck.json'), 'w'), indent=1)
    except OSError:
        pass


def test_termination_condition_objects_read_the_step_kernels_verdict():
    """envs/termination_conditions/*.py as objects: same constructors and get_termination(task, env, info) -> (bad_done, done,
    exceed_time_limit, info) as the reference; each returns its bit of what the fused step found, and OR-ing them the way
    task_base.py:75-96 does reproduces the three masks env.step returned."""
    from neuralplane_amd.envs.control_env import ControlEnv
    from neuralplane_amd.envs.termination_conditions.extreme_state import ExtremeState
    from neuralplane_amd.envs.termination_conditions.high_speed import HighSpeed
    from neuralplane_amd.envs.termination_conditions.low_altitude import LowAltitude
    from neuralplane_amd.envs.termination_conditions.low_speed import LowSpeed
    from neuralplane_amd.envs.termination_conditions.overload import Overload
    from neuralplane_amd.envs.termination_conditions.timeout import Timeout
    from neuralplane_amd.envs.termination_conditions.unreach_heading import UnreachHeading
    n = 600
    env = ControlEnv(num_envs=n, config='heading', model='F16', random_seed=3, device='cuda:0')
    conds = [Overload(env.config), LowAltitude(env.config), HighSpeed(env.config), LowSpeed(env.config), ExtremeState(env.config),
             UnreachHeading(env.config, env.device), Timeout(env.config)]
    env.termination_reasons()                                   # switches the per-aircraft bits on
    env.reset()
    g = torch.Generator(device='cuda').manual_seed(5)
    fired = 0
    for t in range(330):
        a = torch.rand((n, 4), device='cuda', generator=g) * 2 - 1
        a[:, 1] = 1.0
        obs, rew, done, bad, tmo, info = env.step(a)
        bd = torch.zeros_like(bad)
        dn = torch.zeros_like(done)
        tl = torch.zeros_like(tmo)
        for c in conds:
            b_, d_, t_, info_ = c.get_termination(env.task, env, {})
            assert b_.dtype == torch.bool and b_.shape == (n,) and info_ == {}
            bd, dn, tl = bd | b_, dn | d_, tl | t_
        assert torch.equal(bd, bad) and torch.equal(dn, done) and torch.equal(tl, tmo), t
        # the same through the lists the task carries (heading_task.py:34-48), and the reward as the sum of its reward functions
        bd2 = torch.zeros_like(bad)
        for c in env.task.termination_conditions:
            bd2 |= c.get_termination(env.task, env, {})[0]
        assert torch.equal(bd2, bad)
        if t > 0:      # the term tracking was switched on by the first call below, it reports from the following step on
            terms = [f.get_reward(env.task, env) for f in env.task.reward_functions]
            assert len(terms) == 2 and torch.equal((0.0 + terms[0]) + terms[1], rew)
            s_ = env.model.s
            da = (s_[:, 2] - env.task.target_altitude) * 0.3048 / 1000
            dh = torch.remainder(s_[:, 5] - env.task.target_heading + torch.pi, 2 * torch.pi) - torch.pi
            dv = (s_[:, 6] - env.task.target_vt) * 0.3048 / 340
            want = -(da ** 2) - (dh / torch.pi) ** 2 - dv ** 2
            live = ~(bad | done)
            assert torch.allclose(terms[0][live], want[live], rtol=1e-4, atol=1e-6)
        else:
            env.reward_terms()
        fired += int(bad.sum()) + int(done.sum())
    assert fired > 50


def test_obs_after_step_leaves_the_reason_bits_of_the_last_step_alone():
    """env.obs() (one np_f16_reset launch on an all-clear flag input) is observe-only: the per-aircraft condition bits "of the LAST
    step" and the task reward term survive it; only a real reset() clears the bits."""
    from neuralplane_amd.envs.control_env import ControlEnv
    n = 400
    env = ControlEnv(num_envs=n, config='heading', model='F16', random_seed=9, device='cuda:0')
    env.termination_reasons()
    env.reward_terms()
    env.reset()
    a = torch.zeros((n, 4), device='cuda')
    a[:, 1] = 1.0                                         # full elevator: overload / extreme-state trips within a few dozen steps
    seen = False
    for t in range(120):
        env.step(a)
        bits = env.termination_reasons().clone()
        if int((bits != 0).sum()) > 0:
            terms = env._batch.reward_task.clone()
            env.obs()
            assert torch.equal(env.termination_reasons(), bits), 'obs() cleared the reason bits of the last step'
            assert torch.equal(env._batch.reward_task, terms)
            seen = True
            break
    assert seen, 'no condition fired: the test did not exercise anything'
    env.reset()
    assert int(env.termination_reasons().sum()) == 0, 'a real reset() clears the bits'


@pytest.mark.parametrize('task,solver', [('heading', None), ('tracking', None), ('control', 'rk4')])
@pytest.mark.parametrize('n', [300, 20_000, 150_000])      # latency family, latency2, pair variant
def test_model_update_and_model_reset_as_callables_equal_the_oracle_and_mix_with_env_step(task, solver, n):
    """`env.model.update(action)` / `env.model.reset(env)` (reference envs/models/F16_model.py:51-67, :33-45; called
    directly by envs/planning_env.py:160 and example/quick_start.ipynb) are ONE launch each instead of a RuntimeError.  20 x update ==
    the oracle's update bit for bit (state and controls advance for every row; step_count, flags, targets untouched; recent_s / recent_u
    hold the previous state as the reference's do); then update and env.step MIXED — the cross-step coefficient cache is refreshed by
    update as by any step, so the steps that follow read valid keys — and model.reset(env) on the flags a step left behind
    (flagged rows get the re-initialised s / u, nothing else changes), all bit for bit against the oracle on the production RNG."""
    from neuralplane_amd.envs.control_env import ControlEnv
    seed = 11
    env = ControlEnv(num_envs=n, config=task, model='F16', random_seed=seed, device='cuda:0', solver=solver)
    o, st = Oracle(task, solver=solver), Oracle.new_state(n)
    obs = env.reset()
    assert _same(obs.cpu().numpy(), o.reset(st, seed=seed, call_idx=0))
    call = 1
    rng = np.random.RandomState(n % 1000)
    nsteps = 20 if n <= 20_000 else 6

    def state_equal(what):
        b = env._batch
        assert _same(b.s.cpu().numpy().T, st['s']), f'{what}: state'
        assert _same(b.u.cpu().numpy().T, st['u']), f'{what}: controls'
        assert _same(b.tgt.cpu().numpy().T, st['tgt']), f'{what}: targets'
        assert np.array_equal(b.step_count.cpu().numpy(), st['step_count']), f'{what}: step_count'
        f = b.flags.cpu().numpy()
        assert np.array_equal(f[0], st['done']) and np.array_equal(f[1], st['bad']) and np.array_equal(f[2], st['timeout']), f'{what}: masks'

    for t in range(nsteps):                                   # 20 x update
        a = rng.uniform(-1.3, 1.3, (n, 4)).astype(np.float32)
        prev_s, prev_u = env.model.s.clone(), env.model.u.clone()
        env.model.update(torch.from_numpy(a).cuda())
        o.update(st, a)
        call += 1
        state_equal(f'update {t}')
        assert torch.equal(env.model.recent_s, prev_s) and torch.equal(env.model.recent_u, prev_u)
    assert int(env.step_count.max()) == 0 and not bool(env._batch.flags.any())
    flagged_total = 0
    for t in range(nsteps):                                   # update and env.step mixed; model.reset on the flags a step left
        a = rng.uniform(-1.3, 1.3, (n, 4)).astype(np.float32)
        if t % 3 == 1:
            env.model.update(torch.from_numpy(a).cuda())
            o.update(st, a)
            call += 1
            state_equal(f'mixed update {t}')
            continue
        obs, rew, done, bad, tmo, _ = env.step(torch.from_numpy(a).cuda())
        o_obs, o_rew, _, _, _ = o.step(st, a, seed=seed, call_idx=call)
        call += 1
        state_equal(f'mixed step {t}')
        assert _same(obs.cpu().numpy(), o_obs) and _same(rew.cpu().numpy(), o_rew), f'mixed step {t}: obs / reward'
        if t % 3 == 2:
            flagged = (st['done'] | st['bad'] | st['timeout']).astype(bool)
            flagged_total += int(flagged.sum())
            before = {k: v.copy() for k, v in st.items()}
            env.model.reset(env)
            o.model_reset(st, seed=seed, call_idx=call)
            call += 1
            state_equal(f'model.reset {t}')
            assert np.array_equal(st['s'][~flagged], before['s'][~flagged]) and np.array_equal(st['tgt'], before['tgt'])
            if flagged.any():
                assert np.all(st['s'][flagged][:, [0, 1, 3, 4, 5, 7, 8, 9, 10, 11]] == 0) and np.all(st['u'][flagged, 1:] == 0)
    assert flagged_total > 0 or n < 1000, 'no row was ever flagged: model.reset was not exercised'
