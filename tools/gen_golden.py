#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

    python tools/gen_golden.py            # writes every fixture under tests/golden/

The reference (`/root/reference`, pure Python/PyTorch) is imported unmodified on device='cpu'
through the two import shims in tools/oracle_shims/ (torchdiffeq, gym).  Fixtures are DATA only:
inputs and the reference's outputs.  Nothing of the reference travels to the GPU box.

Two flavours are recorded for the floating-point fixtures:
  * plain      — the reference exactly as it runs (ATen sgemm for nn.Linear, SLEEF sin/cos/pow);
                 the oracle / HIP path must agree within tolerance (tests state it).
  * `_pin`     — "pin mode": the only implementation-defined pieces are made implementation-
                 independent by monkey-patching the *libraries the reference calls* (never its
                 files): nn.Linear stacks are evaluated in fp64 and rounded once, sin/cos/tan/pow/sqrt
                 are evaluated in fp64 and rounded once (ATen's fp32 sqrt goes through MKL VML and is
                 not correctly rounded).  What remains is the reference's own
                 fp32 operation order, which the oracle must then reproduce BIT-EXACTLY.

Randomness: the reference draws from torch's global generator; the draws are recorded here by
wrapping torch.rand / rand_like / randn_like and stored in the fixtures (`rand_u`, `noise`) so the
consumers can inject them.
"""
import contextlib
import io
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, 'oracle_shims'))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, '/root/reference/envs')

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402

OUT = os.path.join(REPO, 'tests', 'golden')
torch.set_num_threads(1)

TGT_ATTRS = {'heading': ('target_altitude', 'target_heading', 'target_vt'),
             'control': ('target_pitch', 'target_heading', 'target_vt'),
             'tracking': ('target_npos', 'target_epos', 'target_altitude')}


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()):
        yield


# ------------------------------------------------------------------------------------------------
# library patches
# ------------------------------------------------------------------------------------------------
class Recorder:
    """Wraps torch.rand / rand_like / randn_like to log every draw the reference makes."""

    def __init__(self):
        self.log = []
        self._orig = (torch.rand, torch.rand_like, torch.randn_like)

    def __enter__(self):
        o_rand, o_rand_like, o_randn_like = self._orig

        def rand(*a, **k):
            r = o_rand(*a, **k)
            self.log.append(('rand', r.clone()))
            return r

        def rand_like(*a, **k):
            r = o_rand_like(*a, **k)
            self.log.append(('rand', r.clone()))
            return r

        def randn_like(*a, **k):
            r = o_randn_like(*a, **k)
            self.log.append(('randn', r.clone()))
            return r

        torch.rand, torch.rand_like, torch.randn_like = rand, rand_like, randn_like
        return self

    def __exit__(self, *exc):
        torch.rand, torch.rand_like, torch.randn_like = self._orig

    def take(self):
        out, self.log = self.log, []
        return out


@contextlib.contextmanager
def pin_mode(mlp_cls):
    """fp64-and-round-once versions of the library calls whose fp32 result is implementation-defined."""
    o_sin, o_cos, o_tan, o_pow, o_tpow, o_fwd = torch.sin, torch.cos, torch.tan, torch.pow, torch.Tensor.__pow__, mlp_cls.forward
    o_sqrt = torch.sqrt

    def pw(x, e):
        if isinstance(e, (int, float)) and e == 2:
            return x * x  # ATen's own special case (pow_tensor_scalar): exact fp32 product
        assert isinstance(e, float)
        return o_pow(x.double(), float(np.float32(e))).float()  # ATen rounds the exponent to fp32

    def fwd(self, x):
        h = x.to(torch.float32).double()
        for layer in self.layers:
            h = F.linear(h, layer.weight.double(), layer.bias.double()) if isinstance(layer, nn.Linear) else torch.relu(h)
        return h.float().reshape(-1)

    torch.sin = lambda x: o_sin(x.double()).float()
    torch.cos = lambda x: o_cos(x.double()).float()
    torch.tan = lambda x: o_tan(x.double()).float()
    # ATen CPU routes sqrt through MKL VML (<1 ulp, not correctly rounded): make it the IEEE sqrt
    torch.sqrt = lambda x: o_sqrt(x.double()).float()
    torch.pow = pw
    torch.Tensor.__pow__ = pw
    mlp_cls.forward = fwd
    try:
        yield
    finally:
        torch.sin, torch.cos, torch.tan, torch.pow = o_sin, o_cos, o_tan, o_pow
        torch.sqrt = o_sqrt
        torch.Tensor.__pow__ = o_tpow
        mlp_cls.forward = o_fwd


@contextlib.contextmanager
def maybe_pin(pin, mlp_cls):
    if pin:
        with pin_mode(mlp_cls):
            yield
    else:
        yield


def make_env(task, n, seed=0, solver=None):
    from envs.control_env import ControlEnv
    with quiet():
        env = ControlEnv(num_envs=n, config=task, model='F16', random_seed=seed, device='cpu')
    if solver:
        env.model.solver = solver
    return env


def mlp_class(env):
    return type(env.model.dynamics.hifi_F16.Cx_model)


# ------------------------------------------------------------------------------------------------
# fixture builders
# ------------------------------------------------------------------------------------------------
def aero_inputs(rng, n):
    a = rng.uniform(-20, 90, n)
    b = rng.uniform(-30, 30, n)
    e = rng.uniform(-25, 25, n)
    # edge rows: exact zeros (every reset aircraft), far extrapolation, signed values
    a[:6] = [0, 0, 45, -20, 170, -90]
    b[:6] = [0, 30, -30, 0, 80, -60]
    e[:6] = [0, 0, 25, -25, 45, -45]
    return [np.asarray(v, np.float32) for v in (a, b, e)]


def aero_eval(hifi, a, b, e):
    a, b, e = (torch.from_numpy(v) for v in (a, b, e))
    groups = [hifi.hifi_C(a, b, e), hifi.hifi_damping(a), hifi.hifi_C_lef(a, b), hifi.hifi_damping_lef(a),
              hifi.hifi_rudder(a, b), hifi.hifi_ailerons(a, b), hifi.hifi_other_coeffs(a, e)[:4]]
    return torch.stack([t for g in groups for t in g], dim=1).numpy()


def gen_aero(env):
    rng = np.random.RandomState(11)
    a, b, e = aero_inputs(rng, 384)
    hifi = env.model.dynamics.hifi_F16
    out = aero_eval(hifi, a, b, e)
    with pin_mode(mlp_class(env)):
        out_pin = aero_eval(hifi, a, b, e)
    assert out.shape == (384, 43)
    np.savez_compressed(os.path.join(OUT, 'aero_kat.npz'), alpha_deg=a, beta_deg=b, el=e, coef=out, coef_pin=out_pin)


# ------------------------------------------------------------------------------------------------
# The reference's own validation data for the aero surrogates: envs/models/F16/model/coefs.csv is the 630-point grid
# (rows 0-2: alpha, beta, el in degrees) with the table-interpolated coefficient of every net (rows 3-45, the row map of
# model/test_model.py:71-335), which the authors' script compares their MLPs against (r2 / mean absolute error).  The
# leading-edge-flap and aileron tables only span the first 400 grid points (alpha <= 45 deg, test_model.py:163-301).
# ------------------------------------------------------------------------------------------------
# column of aero_eval()'s 43-vector -> row of coefs.csv
MODEL_GRID_ROWS = (list(range(3, 9)) + list(range(9, 18)) + list(range(18, 24)) + list(range(24, 33)) + list(range(33, 36))
                   + [36, 39, 37, 40, 38, 41] + list(range(42, 46)))
MODEL_GRID_400 = set(range(15, 30)) | set(range(33, 39))   # C_lef, damping_lef, ailerons: first 400 points


def gen_model_grid(env):
    import pandas as pd
    d = np.array(pd.read_csv('/root/reference/envs/models/F16/model/coefs.csv', header=None))
    assert d.shape == (47, 630) and len(MODEL_GRID_ROWS) == 43
    a, b, e = (d[k].astype(np.float32) for k in range(3))   # test_model.py feeds float64 tensors, MLP.forward casts to float32
    hifi = env.model.dynamics.hifi_F16
    out = aero_eval(hifi, a, b, e)
    with pin_mode(mlp_class(env)):
        out_pin = aero_eval(hifi, a, b, e)
    table = d[MODEL_GRID_ROWS].T.copy()                     # [630, 43] float64, the reference's data file as it is
    npts = np.array([400 if k in MODEL_GRID_400 else 630 for k in range(43)], np.int32)
    r2 = np.zeros(43)
    mae = np.zeros(43)
    for k in range(43):
        y, f = table[:npts[k], k], out[:npts[k], k].astype(np.float64)
        r2[k] = 1.0 - np.sum((y - f) ** 2) / np.sum((y - y.mean()) ** 2)   # sklearn.metrics.r2_score
        mae[k] = np.mean(np.abs(y - f))
    np.savez_compressed(os.path.join(OUT, 'model_grid_kat.npz'), alpha_deg=a, beta_deg=b, el=e, table=table, npts=npts,
                        coef=out, coef_pin=out_pin, ref_r2=r2, ref_mae=mae)
    print('model grid: r2 of the reference MLPs against its tables: min %.4f median %.5f' % (r2.min(), np.median(r2)))


def random_flight_states(rng, n):
    s = np.zeros((n, 12), np.float32)
    s[:, 0] = rng.uniform(-5e4, 5e4, n)
    s[:, 1] = rng.uniform(-5e4, 5e4, n)
    s[:, 2] = rng.uniform(1000, 45000, n)
    s[:, 3] = rng.uniform(-3.2, 3.2, n)
    s[:, 4] = rng.uniform(-1.4, 1.4, n)
    s[:, 5] = rng.uniform(-7, 7, n)
    s[:, 6] = rng.uniform(150, 1500, n)
    s[:, 7] = rng.uniform(-0.4, 0.9, n)
    s[:, 8] = rng.uniform(-0.5, 0.5, n)
    s[:, 9:12] = rng.uniform(-2, 2, (n, 3))
    u = np.zeros((n, 5), np.float32)
    u[:, 0] = rng.uniform(-2000, 60000, n)
    u[:, 1:4] = rng.uniform(-45, 45, (n, 3))
    return s, u


def gen_nlplant(env):
    rng = np.random.RandomState(12)
    s, u = random_flight_states(rng, 512)
    # edge rows
    s[0] = 0; s[0, 2] = 19500; s[0, 6] = 1100; u[0] = [2000, 0, 0, 0, 0]  # a freshly reset aircraft
    s[1, 6] = 0.005          # vt below the 0.01 clamp
    s[2, 6] = -3.0           # negative vt
    s[3, 2] = 36000.0        # above the tropopause switch
    s[4, 4] = 1.5707         # pitch close to pi/2
    s[5, 5] = 50.0; s[5, 3] = -40.0  # many turns of yaw / roll
    s[6, 7] = 0.0; s[6, 8] = 0.0
    x = np.hstack([s, u]).astype(np.float32)
    dyn = env.model.dynamics
    xd = dyn.nlplant(torch.from_numpy(x)).numpy()[:, :12]
    with pin_mode(mlp_class(env)):
        xd_pin = dyn.nlplant(torch.from_numpy(x)).numpy()[:, :12]
    np.savez_compressed(os.path.join(OUT, 'nlplant_kat.npz'), x17=x, xdot=xd, xdot_pin=xd_pin)


def gen_getters(env_unused):
    rng = np.random.RandomState(13)
    n = 256
    s, u = random_flight_states(rng, n)
    env = make_env('heading', n)
    out = {}
    for pin in (False, True):
        with maybe_pin(pin, mlp_class(env)):
            env.model.s = torch.from_numpy(s.copy())
            env.model.u = torch.from_numpy(u.copy())
            m = env.model
            sfx = '_pin' if pin else ''
            out['accel' + sfx] = torch.stack(m.get_acceleration(), 1).numpy()
            out['accels' + sfx] = torch.stack(m.get_accels(), 1).numpy()
            out['G' + sfx] = m.get_G().numpy()
            out['eas2tas' + sfx] = m.get_EAS2TAS().numpy()
            out['eas' + sfx] = m.get_EAS().numpy()
            out['xdot' + sfx] = m.get_extended_state().numpy()[:, :12]
            out['atmos' + sfx] = torch.stack(m.get_atmos(), 1).numpy()      # (mach, qbar, ps), F16_model.py:183-198
    np.savez_compressed(os.path.join(OUT, 'getters_kat.npz'), s=s, u=u, **out)


def get_tgt(env, task):
    return torch.stack([getattr(env.task, a) for a in TGT_ATTRS[task]], 1).numpy().copy()


def set_tgt(env, task, tgt):
    for k, a in enumerate(TGT_ATTRS[task]):
        setattr(env.task, a, torch.from_numpy(tgt[:, k].copy()))


def force_state(env, task, st):
    env.model.s = torch.from_numpy(st['s'].copy())
    env.model.u = torch.from_numpy(st['u'].copy())
    env.model.recent_s = env.model.s.clone()
    env.model.recent_u = env.model.u.clone()
    set_tgt(env, task, st['tgt'])
    env.step_count = torch.from_numpy(st['step_count'].copy())
    env.is_done = torch.from_numpy(st['done'].astype(bool))
    env.bad_done = torch.from_numpy(st['bad'].astype(bool))
    env.exceed_time_limit = torch.from_numpy(st['timeout'].astype(bool))


def read_state(env, task):
    return dict(s=env.model.s.numpy().copy(), u=env.model.u.numpy().copy(), tgt=get_tgt(env, task),
                step_count=env.step_count.numpy().copy())


def draws_to_arrays(log, reset_mask, n, n_task_draws, which_randn):
    """Map the recorded draws of one reset()/step() call onto per-row arrays."""
    rand = [t.numpy() for k, t in log if k == 'rand']
    randn = [t.numpy() for k, t in log if k == 'randn']
    rand_u = np.zeros((n, 5), np.float32)
    assert len(rand) == 2 + n_task_draws, (len(rand), n_task_draws)
    for c, r in enumerate(rand):
        assert r.shape == (int(reset_mask.sum()),)
        rand_u[reset_mask, c] = r
    noise = randn[which_randn].astype(np.float32)
    assert noise.shape == (n, 22)
    return rand_u, noise


def ref_step(env, task, st, action, pin):
    """One teacher-forced env.step from state `st`; returns outputs + the randomness consumed."""
    n = st['s'].shape[0]
    force_state(env, task, st)
    reset_mask = (st['done'] | st['bad'] | st['timeout']).astype(bool)
    with maybe_pin(pin, mlp_class(env)), Recorder() as rec, quiet():
        obs, rew, done, bad, tmo, _ = env.step(torch.from_numpy(action.copy()))
        log = rec.take()
    rand_u, noise = draws_to_arrays(log, reset_mask, n, 0 if task == 'heading' else 3, which_randn=1)
    out = read_state(env, task)
    out.update(obs=obs.numpy().copy(), reward=rew.numpy().copy(), done=done.numpy().astype(np.uint8),
               bad=bad.numpy().astype(np.uint8), timeout=tmo.numpy().astype(np.uint8))
    return out, rand_u, noise


def build_step_inputs(task, n, rng):
    """Input states for the teacher-forced step KAT: a mix of fresh rows (all flags set, as after
    construction), mid-flight rows, rows at the step_count gates and rows sitting on their target."""
    s, u = random_flight_states(rng, n)
    # keep most rows inside the envelope so that not everything terminates
    s[:, 2] = rng.uniform(3000, 30000, n)
    s[:, 3] = rng.uniform(-1.0, 1.0, n)
    s[:, 4] = rng.uniform(-0.5, 0.5, n)
    s[:, 6] = rng.uniform(400, 1300, n)
    s[:, 7] = rng.uniform(-0.1, 0.4, n)
    s[:, 8] = rng.uniform(-0.1, 0.1, n)
    s[:, 9:12] = rng.uniform(-0.5, 0.5, (n, 3))
    u[:, 0] = rng.uniform(0, 30000, n)
    u[:, 1:4] = rng.uniform(-10, 10, (n, 3))
    tgt = np.zeros((n, 3), np.float32)
    if task == 'heading':
        tgt[:, 0] = s[:, 2] + rng.uniform(-1500, 1500, n)
        tgt[:, 1] = rng.uniform(-3.1, 3.1, n)
        tgt[:, 2] = s[:, 6] + rng.uniform(-100, 100, n)
    elif task == 'control':
        tgt[:, 0] = rng.uniform(-1.0, 1.0, n)
        tgt[:, 1] = rng.uniform(-3.1, 3.1, n)
        tgt[:, 2] = s[:, 6] + rng.uniform(-100, 100, n)
    else:
        tgt[:, 0] = s[:, 0] + rng.uniform(-3000, 3000, n)
        tgt[:, 1] = s[:, 1] + rng.uniform(-3000, 3000, n)
        tgt[:, 2] = s[:, 2] + rng.uniform(-1500, 1500, n)
    step_count = rng.randint(0, 2600, n).astype(np.int64)
    gates = [298, 299, 300, 301, 2498, 2499, 2500, 2501]
    step_count[:32] = np.tile(gates, 4)
    # rows 0..63: sitting (almost) on the target so that `done` can fire / just miss
    k = 64
    if task == 'heading':
        tgt[:k, 0] = s[:k, 2] + rng.uniform(-120, 120, k)
        tgt[:k, 1] = s[:k, 5] + rng.uniform(-0.1, 0.1, k)
        tgt[:k, 2] = s[:k, 6] + rng.uniform(-25, 25, k)
    elif task == 'control':
        tgt[:k, 0] = s[:k, 4] + rng.uniform(-0.1, 0.1, k)
        tgt[:k, 1] = s[:k, 5] + rng.uniform(-0.1, 0.1, k)
        tgt[:k, 2] = s[:k, 6] + rng.uniform(-25, 25, k)
    else:
        tgt[:k] = s[:k, :3] + rng.uniform(-120, 120, (k, 3))
    # hazard rows: low altitude, slow, fast, extreme alpha/beta, violent rates (overload)
    s[64, 2] = 2500.5; s[65, 2] = 2499.0; s[66, 6] = 3400.0; s[67, 6] = 9.0
    s[68, 7] = 0.79; s[69, 7] = -0.36; s[70, 8] = 0.53; s[71, 8] = -0.53
    s[72, 9:12] = [6.0, -5.0, 4.0]; s[73, 10] = 9.0; u[74, 1] = 45.0; s[75, 6] = 0.004
    done = np.zeros(n, np.uint8)
    bad = np.zeros(n, np.uint8)
    tmo = np.zeros(n, np.uint8)
    # rows 96..159 arrive flagged (different flag combinations) and get re-initialised first
    done[96:128] = 1
    bad[112:144] = 1
    tmo[140:160] = 1
    action = rng.uniform(-1.3, 1.3, (n, 4)).astype(np.float32)  # beyond [-1,1] on purpose: clamp
    action[0] = [1, 0, 0, 0]
    return dict(s=s.astype(np.float32), u=u.astype(np.float32), tgt=tgt.astype(np.float32),
                step_count=step_count, done=done, bad=bad, timeout=tmo), action


def gen_step_kat(task, solver=None, n=256, tag=None):
    rng = np.random.RandomState({'heading': 21, 'control': 22, 'tracking': 23}[task] + (100 if solver else 0))
    env = make_env(task, n, solver=solver)
    st, action = build_step_inputs(task, n, rng)
    data = {('in_' + k): v for k, v in st.items()}
    data['action'] = action
    for pin in (False, True):
        torch.manual_seed(1234)
        out, rand_u, noise = ref_step(env, task, st, action, pin)
        sfx = '_pin' if pin else ''
        for k, v in out.items():
            data['out_' + k + sfx] = v
        data['rand_u'] = rand_u  # same seed -> same draws in both flavours
        data['noise'] = noise
    # the very first step of a fresh env: every row flagged (env_base.py:31-33)
    env2 = make_env(task, n, solver=solver)
    st0 = dict(s=np.zeros((n, 12), np.float32), u=np.zeros((n, 5), np.float32), tgt=np.zeros((n, 3), np.float32),
               step_count=np.zeros(n, np.int64), done=np.ones(n, np.uint8), bad=np.ones(n, np.uint8),
               timeout=np.ones(n, np.uint8))
    for pin in (False, True):
        torch.manual_seed(4321)
        out, rand_u, noise = ref_step(env2, task, st0, action, pin)
        sfx = '_pin' if pin else ''
        for k, v in out.items():
            data['first_out_' + k + sfx] = v
        data['first_rand_u'] = rand_u
        data['first_noise'] = noise
    name = tag or f'step_kat_{task}'
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **data)


def traj_actions(T, n, seed=123):
    """Deterministic action sequence shared with tests/ (numpy RandomState is version-stable)."""
    rng = np.random.RandomState(seed)
    t = np.arange(T, dtype=np.float64)[:, None, None]
    phase = rng.uniform(0, 2 * np.pi, (1, n, 4))
    freq = rng.uniform(0.002, 0.02, (1, n, 4))
    a = 0.3 * np.sin(2 * np.pi * freq * t + phase) + rng.uniform(-1, 1, (T, n, 4)) * np.array([1.0, 0.3, 0.3, 0.3])
    a[..., 0] = 0.5 + 0.5 * a[..., 0]
    return np.clip(a, -1, 1).astype(np.float32)


def gen_traj(task='heading', n=256, T=1000):
    env = make_env(task, n, seed=0)
    env.task.noise_scale = 0  # keeps the fixture small; noise parity is covered by step_kat
    acts = traj_actions(T, n)
    states, obs_l, rew_l = [], [], []
    flags = np.zeros((T, n, 3), np.uint8)
    rand_u = np.zeros((T, n, 5), np.float32)
    tgts = []
    n_task = 0 if task == 'heading' else 3
    for t in range(T):
        prev = (env.is_done | env.bad_done | env.exceed_time_limit).numpy().astype(bool)
        with Recorder() as rec, quiet():
            obs, rew, done, bad, tmo, _ = env.step(torch.from_numpy(acts[t]))
            log = rec.take()
        ru, _ = draws_to_arrays(log, prev, n, n_task, which_randn=1)
        rand_u[t] = ru
        flags[t, :, 0] = done.numpy()
        flags[t, :, 1] = bad.numpy()
        flags[t, :, 2] = tmo.numpy()
        if (t + 1) % 10 == 0 or t < 3 or t + 1 == 426:   # 426 = the length of the authors' recorded episode
            states.append(np.hstack([env.model.s.numpy(), env.model.u.numpy()[:, :4], get_tgt(env, task)]))
            obs_l.append(obs.numpy().copy())
            rew_l.append(rew.numpy().copy())
    rec_steps = np.array([t for t in range(T) if (t + 1) % 10 == 0 or t < 3 or t + 1 == 426], np.int64)
    np.savez_compressed(os.path.join(OUT, f'traj_{task}_N{n}_T{T}.npz'), action_seed=np.int64(123),
                        rec_steps=rec_steps, state=np.stack(states).astype(np.float32),
                        obs=np.stack(obs_l).astype(np.float32), reward=np.stack(rew_l).astype(np.float32),
                        flags=flags, rand_u=rand_u, step_count_final=env.step_count.numpy())
    print(f'traj {task}: done={int(flags[:, :, 0].sum())} bad={int(flags[:, :, 1].sum())}')


def gen_traj_closed(n=256, T=1000):
    """Heading with a policy in the loop (tools/parity_report.py::closed_loop_action on the reference's own state): every
    aircraft flies the whole 1000 steps, so round-off has 1000 uninterrupted steps to grow."""
    sys.path.insert(0, REPO)
    from tools.parity_report import closed_loop_action, closed_loop_setup
    env = make_env('heading', n, seed=0)
    env.task.noise_scale = 0
    th_cmd, phi_cmd, rng = closed_loop_setup(n)
    states = []
    flags = np.zeros((T, n, 3), np.uint8)
    rand_u = np.zeros((T, n, 5), np.float32)
    s = np.zeros((n, 12), np.float32)
    for t in range(T):
        dither = rng.uniform(-0.05, 0.05, (n, 3)).astype(np.float32)
        a = closed_loop_action(s, th_cmd, phi_cmd, dither)
        prev = (env.is_done | env.bad_done | env.exceed_time_limit).numpy().astype(bool)
        with Recorder() as rec, quiet():
            obs, rew, done, bad, tmo, _ = env.step(torch.from_numpy(a))
            log = rec.take()
        rand_u[t], _ = draws_to_arrays(log, prev, n, 0, which_randn=1)
        flags[t, :, 0], flags[t, :, 1], flags[t, :, 2] = done.numpy(), bad.numpy(), tmo.numpy()
        s = env.model.s.numpy().astype(np.float32).copy()
        if (t + 1) % 10 == 0 or t < 3 or t + 1 == 426:
            states.append(np.hstack([s, env.model.u.numpy()[:, :4], get_tgt(env, 'heading')]))
    rec_steps = np.array([t for t in range(T) if (t + 1) % 10 == 0 or t < 3 or t + 1 == 426], np.int64)
    np.savez_compressed(os.path.join(OUT, f'traj_heading_closed_N{n}_T{T}.npz'), rec_steps=rec_steps, state=np.stack(states).astype(np.float32),
                        flags=flags, rand_u=rand_u, step_count_final=env.step_count.numpy())
    print(f'traj closed: done={int(flags[:, :, 0].sum())} bad={int(flags[:, :, 1].sum())} longest episode {int(env.step_count.max())}')


ACTION_QUANTUM = 4096.0   # PID-driven fixtures store their actions as int16 multiples of 1 / 4096 (what the reference env was driven with)


def _pid_loop(env, task, T, n, drive, rec_every=25, overrides=None, name=None):
    """Free-running reference env with the reference's OWN PID stack in the loop (algorithms/pid/controller.py:69,140 — SURVEY §2 #21):
    `drive(controller, env, t)` issues the outer-loop calls of renders/render_control.py:82-88, the controller's action is rounded to a
    multiple of 1 / 4096 (so that the fixture can hold it exactly as int16) and THAT is what the reference env steps on.  Recorded: every
    action, every mask, the reset draws, and (s, u, targets) every `rec_every` steps and at the step after every `done`."""
    from algorithms.pid.controller import Controller
    n_task = 0 if task == 'heading' else 3
    with Recorder() as rec, quiet():      # the controller reads the env's state before the first step: reset first, its draws recorded too
        env.reset()
        log0 = rec.take()
    rand_u_reset, _ = draws_to_arrays(log0, np.ones(n, bool), n, n_task, which_randn=0)
    with quiet():
        c = Controller(dt=env.model.dt, n=env.n, device='cpu')
    acts = np.zeros((T, n, 4), np.int16)
    flags = np.zeros((T, n, 3), np.uint8)
    rand_u = np.zeros((T, n, 5), np.float32)
    states, rec_steps, after_done = [], [], False
    edit_t, edit_s = [], []
    for t in range(T):
        prev = (env.is_done | env.bad_done | env.exceed_time_limit).numpy().astype(bool)
        s_before = env.model.s.clone()
        with quiet():
            drive(c, env, t)
            c.stabilize(env)
            a = torch.round(torch.clamp(c.get_action(), -2, 2) * ACTION_QUANTUM)
        # The reference's TECS edits the env's state IN PLACE on its first call (the altitude it reads is a view of model.s; +83.33 ft): a quirk
        # of the consumer, outside env.step — recorded as an external state write that the replay applies at the same point
        if not torch.equal(s_before, env.model.s):
            assert torch.equal(s_before[:, [0, 1, 3, 4, 5, 6, 7, 8, 9, 10, 11]], env.model.s[:, [0, 1, 3, 4, 5, 6, 7, 8, 9, 10, 11]])
            edit_t.append(t)
            edit_s.append(env.model.s.numpy().copy())
        with Recorder() as rec, quiet():
            acts[t] = a.numpy().astype(np.int16)
            obs, rew, done, bad, tmo, _ = env.step((a / ACTION_QUANTUM).to(torch.float32))
            log = rec.take()
        rand_u[t], _ = draws_to_arrays(log, prev, n, n_task, which_randn=1)
        flags[t, :, 0], flags[t, :, 1], flags[t, :, 2] = done.numpy(), bad.numpy(), tmo.numpy()
        if (t + 1) % rec_every == 0 or t < 3 or after_done or t + 1 == T:   # (after_done: this step re-initialised the rows that were done)
            states.append(np.hstack([env.model.s.numpy(), env.model.u.numpy()[:, :4], get_tgt(env, task)]))
            rec_steps.append(t)
        after_done = bool(done.any())
    return dict(actions_q=acts, action_quantum=np.float32(ACTION_QUANTUM), flags=flags, rand_u=rand_u, rand_u_reset=rand_u_reset,
                state_edit_steps=np.array(edit_t, np.int64), state_edits=np.stack(edit_s).astype(np.float32) if edit_s else np.zeros((0, n, 12), np.float32), rec_steps=np.array(rec_steps, np.int64),
                state=np.stack(states).astype(np.float32), step_count_final=env.step_count.numpy())


def gen_traj_pid_heading(n=64, T=2600):
    """a Heading trajectory in which `done` FIRES.  heading_task.py:56-67 always asks for +120 deg of heading and
    +1000 ft; the reference's PID stack (heading hold from the first step) flies that turn in 1 700-2 600 steps depending on the bank limit —
    set per aircraft here (Controller.roll_limit 0.70 ... 1.05 rad; the shipped 45 deg would need ~2 700 steps) so that the fast rows reach
    the target (UnreachHeading.done: 300 <= step < 2 500, |d heading| < 5 deg, |d alt| < 100 ft, |d vt| < 20 ft/s, unreach_heading.py:38-53)
    and the slow ones run into max_check_interval = 2 500 (bad).  After a `done` the whole chain is in the recording: BaseEnv.step's
    self.reset() -> F16Model.reset (state re-drawn, F16_model.py:37-45) -> HeadingTask.reset (new targets) -> step_count = 0."""
    env = make_env('heading', n, seed=0)
    env.task.noise_scale = 0
    limits = torch.linspace(0.70, 1.05, n).reshape(-1, 1)

    def drive(c, env, t):
        c.roll_limit = limits
        if t % 5 == 0:
            c.cal_pitch_throttle(env.task.target_altitude.reshape(-1, 1), env.task.target_vt.reshape(-1, 1), env)
            c.update_heading_hold(env.task.target_heading.reshape(-1, 1), env)
    d = _pid_loop(env, 'heading', T, n, drive)
    fl = d['flags']
    n_done, n_bad = int(fl[:, :, 0].sum()), int(fl[:, :, 1].sum())
    t_done = np.nonzero(fl[:, :, 0].any(1))[0]
    assert n_done >= 20, n_done
    assert n_bad >= 1 and t_done.min() + 1 >= 300 and t_done.max() + 1 < 2500
    # the re-initialisations that follow: the step after a done row's flag, its counter restarts at 1 (reset to 0, then += 1)
    print(f'traj pid heading: done={n_done} (steps {t_done.min() + 1}..{t_done.max() + 1}) bad={n_bad} recorded states {len(d["rec_steps"])}')
    np.savez_compressed(os.path.join(OUT, f'traj_pid_heading_N{n}_T{T}.npz'), **d)


PID_CONTROL_OVERRIDES = {'max_pitch_increment': 0.15, 'max_heading_increment': 0.25, 'max_velocities_u_increment': 40.0}


def gen_traj_pid_control(n=64, T=400):
    """VERDICT r5 item 3, Control: target increments small enough (scenario keys max_pitch_increment / max_heading_increment /
    max_velocities_u_increment = 0.15 rad / 0.25 rad / 40 ft/s instead of control.yaml's 3 / 3 / 300; stored in the fixture, both sides
    read them) that UnreachPosture.done (|d pitch| < 5 deg — no wrap —, |d heading| < 5 deg, |d vt| < 20 ft/s, any step count below
    max_check_interval: unreach_posture.py:40-55) fires within 300 steps with the reference's PID stack in the loop: pitch demand = the
    task's target pitch, heading hold on its target heading, TECS throttle on its target vt."""
    import json
    env = make_env('control', n, seed=0)
    env.task.noise_scale = 0
    for k, v in PID_CONTROL_OVERRIDES.items():
        assert hasattr(env.task, k), k
        setattr(env.task, k, v)

    def drive(c, env, t):
        if t % 5 == 0:
            alt = env.model.get_position()[2].reshape(-1, 1)
            c.cal_pitch_throttle(alt, env.task.target_vt.reshape(-1, 1), env)
            c.update_heading_hold(env.task.target_heading.reshape(-1, 1), env)
        c.pitch_dem = env.task.target_pitch.reshape(-1, 1).clone()
    d = _pid_loop(env, 'control', T, n, drive, rec_every=20)
    fl = d['flags']
    n_done = int(fl[:, :, 0].sum())
    late = int(fl[20:, :, 0].sum())          # reached by flying there, not by a lucky draw at reset
    assert n_done >= 20 and late >= 10, (n_done, late)
    print(f'traj pid control: done={n_done} ({late} after step 20) bad={int(fl[:, :, 1].sum())} recorded states {len(d["rec_steps"])}')
    np.savez_compressed(os.path.join(OUT, f'traj_pid_control_N{n}_T{T}.npz'), overrides=np.array(json.dumps(PID_CONTROL_OVERRIDES)), **d)


def gen_recorded_episode():
    """Rows 0..426 of the authors' CUDA recording renders/result/*.npy (render_ppo.py:157-175)."""
    d = '/root/reference/renders/result'
    cols = ['npos', 'epos', 'altitude', 'roll', 'pitch', 'yaw', 'vt', 'alpha', 'beta', 'G', 'T', 'el', 'ail', 'rud']
    arr = np.stack([np.load(os.path.join(d, c + '.npy')).reshape(-1)[:427] for c in cols], 1).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, 'recorded_episode0.npz'), columns=np.array(cols), rows=arr)
    # the same replay with the REFERENCE's own CPU dynamics: x' = x + 0.02 * nlplant(x, recorded controls),
    # as the reference runs (plain ATen) and in pin mode (MLPs / sin / cos / pow in fp64, rounded once).  Both trajectories are stored
    # (recorded_episode0_ref_cpu.npz) so that the attribution of the end-of-episode residual against the CUDA recording is evidence:
    # reference-CPU vs the CUDA recording, this build vs the CUDA recording and this build vs reference-CPU on the SAME inputs.
    from envs.models.F16.F16_dynamics import F16Dynamics
    with quiet():
        dyn = F16Dynamics('cpu')
    floors = np.array([100, 100, 100, .1, .1, .1, 10, .1, .1], np.float32)

    def replay():
        s = torch.zeros(1, 12)
        s[0, 2] = float(arr[0, 2]); s[0, 6] = float(arr[0, 6])
        traj, worst = [s[0].numpy().copy()], 0.0
        for t in range(426):
            u = torch.tensor([[arr[t + 1, 10], arr[t + 1, 11], arr[t + 1, 12], arr[t + 1, 13], 0.0]])
            x = torch.hstack((s, u))
            s = (x + torch.tensor(0.02) * dyn.nlplant(x))[:, :12]
            ref = arr[t + 1, :9]
            worst = max(worst, float(np.max(np.abs(s[0, :9].numpy() - ref) / np.maximum(np.abs(ref), floors))))
            traj.append(s[0].numpy().copy())
        return np.stack(traj).astype(np.float32), worst

    plain, w_plain = replay()
    with pin_mode(type(dyn.hifi_F16.Cx_model)):
        pin, w_pin = replay()
    np.savez_compressed(os.path.join(OUT, 'recorded_episode0_ref_cpu.npz'), states=plain, states_pin=pin,
                        worst_vs_cuda_recording=np.array([w_plain, w_pin]))
    print('recorded episode replay with the reference CPU dynamics: worst rel err vs the CUDA recording (SURVEY floors): plain', w_plain, 'pin mode', w_pin)


def gen_planning(n=48, outer=3):
    """PlanningEnv.step (envs/planning_env.py:144-177) with a seeded RANDOM-INIT low-level PPOActor (the trained
    checkpoint is not part of the reference snapshot).  Records the 50 low-level actions of every outer step so
    that consumers can replay them instead of re-running the GRU (whose GEMMs are implementation-defined)."""
    if not hasattr(np, 'product'):
        np.product = np.prod  # the reference targets numpy 1.x (algorithms/utils/flatten.py:83)
    import envs.planning_env as pe
    o_load = torch.load
    sentinel = object()
    torch.load = lambda f, *a, **k: sentinel if str(f).endswith('actor_latest.pt') else o_load(f, *a, **k)
    actor_cls = pe.PPOActor
    o_lsd = actor_cls.load_state_dict
    # the ACTOR keeps its seeded random initialisation (its checkpoint is not shipped); the aero MLPs load normally
    actor_cls.load_state_dict = lambda self, sd, *a, **k: None if sd is sentinel else o_lsd(self, sd, *a, **k)
    try:
        with quiet():
            env = pe.PlanningEnv(num_envs=n, config='tracking', model='F16', random_seed=0, device='cpu')
    finally:
        torch.load = o_load
        actor_cls.load_state_dict = o_lsd
    # amplify the (gain=0.01) random actor so that the low-level actions actually move the aircraft
    scale = 25.0
    real = env.controller

    class Rec:
        def __init__(self):
            self.log = []

        def __call__(self, obs, rnn, masks, deterministic=True):
            a, lp, rnn = real(obs, rnn, masks, deterministic=deterministic)
            a = a * scale
            self.log.append((obs.numpy().copy(), a.numpy().copy()))
            return a, lp, rnn

    rec = Rec()
    env.controller = rec
    rng = np.random.RandomState(31)
    data = {}
    hi_actions = rng.uniform(-1.2, 1.2, (outer, n, 3)).astype(np.float32)
    for k in range(outer):
        rec.log = []
        prev = (env.is_done | env.bad_done | env.exceed_time_limit).numpy().astype(bool)
        with Recorder() as r, quiet():
            obs, rew, done, bad, tmo, _ = env.step(torch.from_numpy(hi_actions[k]))
            log = r.take()
        rand = [t.numpy() for kind, t in log if kind == 'rand']
        rand_u = np.zeros((n, 5), np.float32)
        assert len(rand) == 5
        for c, x in enumerate(rand):
            rand_u[prev, c] = x
        data[f'rand_u_{k}'] = rand_u
        data[f'll_obs_{k}'] = np.stack([o for o, _ in rec.log])
        data[f'll_act_{k}'] = np.stack([a for _, a in rec.log])
        data[f's_{k}'] = env.model.s.numpy().copy()
        data[f'u_{k}'] = env.model.u.numpy().copy()
        data[f'tgt_{k}'] = get_tgt(env, 'tracking')
        data[f'step_count_{k}'] = env.step_count.numpy().copy()
        data[f'obs_{k}'] = obs.numpy().copy()
        data[f'reward_{k}'] = rew.numpy().copy()
        data[f'flags_{k}'] = np.stack([done.numpy(), bad.numpy(), tmo.numpy()]).astype(np.uint8)
        print('planning outer', k, 'done', int(done.sum()), 'bad', int(bad.sum()))
    np.savez_compressed(os.path.join(OUT, 'planning_kat.npz'), hi_actions=hi_actions, **data)


def teacher_lowlevel_action(obs):
    """An attitude-hold law on PlanningEnv.low_level_obs (planning_env.py:60-142; the gains of tools/parity_report.py::closed_loop_action, which
    keep 256 aircraft flying for 1000 steps): elevator on the pitch error + Q, aileron on roll towards a bank command proportional to the heading
    error + P, yaw damper, throttle on the speed error.  Only the labels of cloned_actor() come from it."""
    d_pitch, d_head, d_vt, sin_roll = obs[:, 0], obs[:, 1], obs[:, 2], obs[:, 4]
    P, Q, R = obs[:, 13], obs[:, 14], obs[:, 15]
    phi_cmd = torch.clamp(-1.0 * d_head, -0.4, 0.4)
    a = torch.stack([0.3 - 3.0 * d_vt, 2.0 * d_pitch + 1.0 * Q, 1.0 * (sin_roll - phi_cmd) + 0.5 * P, 0.2 * R], 1)
    a[:, 0] = torch.clamp(a[:, 0], 0.0, 1.0)
    return torch.clamp(a, -0.97, 0.97)


def cloned_actor(n=64, rounds=5, macro=6, epochs=300, hi_amp=0.5):
    """A reference PPOActor that FLIES: seeded_actor()'s random initialisation fitted (plain regression, Adam, fixed seeds) to
    teacher_lowlevel_action on the (observation, recurrent state) pairs its own closed loop visits in the reference's PlanningEnv
    (DAgger: round 0 the teacher drives, afterwards the actor itself; the labels are always the teacher's).  Test-fixture tooling: nothing
    of an RL algorithm — the result is a state_dict of the reference's own module whose closed loop keeps the aircraft in the air for
    1 000 inner steps, which a random-init head does not (24 of 64 rows end in Overload per macro-step)."""
    import envs.planning_env as pe
    actor = seeded_actor(4.0, 0.0)
    o_load = torch.load
    torch.load = lambda f, *a, **k: {k_: v.clone() for k_, v in actor.state_dict().items()} if str(f).endswith('actor_latest.pt') else o_load(f, *a, **k)
    try:
        with quiet():
            env = pe.PlanningEnv(num_envs=n, config='tracking', model='F16', random_seed=1, device='cpu')
    finally:
        torch.load = o_load
    data_obs, data_h, data_y = [], [], []
    state = {'student_drives': False}

    def controller(obs, rnn, masks, deterministic=True):
        with torch.no_grad():
            a_s, _, h = actor(obs, rnn, masks, deterministic=True)
        y = teacher_lowlevel_action(obs)
        data_obs.append(obs.clone()); data_h.append(rnn.clone()); data_y.append(y)
        return (a_s if state['student_drives'] else y), None, h
    env.controller = controller
    rng = np.random.RandomState(5)
    opt = torch.optim.Adam(actor.parameters(), lr=2e-3)
    for r in range(rounds):
        state['student_drives'] = r > 0
        with quiet():
            env.is_done[:] = 1        # every round starts from freshly drawn states
            env.ego_rnn_states = torch.zeros_like(env.ego_rnn_states)
            bad_total = 0
            for k in range(macro):
                _, _, done, bad, tmo, _ = env.step(torch.from_numpy(rng.uniform(-hi_amp, hi_amp, (n, 3)).astype(np.float32)))
                bad_total += int(bad.sum())
        X, H, Y = torch.cat(data_obs), torch.cat(data_h), torch.cat(data_y)
        ok = torch.isfinite(X).all(1) & torch.isfinite(Y).all(1)
        X, H, Y = X[ok], H[ok], Y[ok]
        actor.train()
        g = torch.Generator().manual_seed(100 + r)
        for e in range(epochs):
            idx = torch.randint(0, X.shape[0], (4096,), generator=g)
            a, _, _ = actor(X[idx], H[idx], torch.ones((idx.numel(), 1)), deterministic=True)
            loss = ((a - Y[idx]) ** 2).mean()
            opt.zero_grad(); loss.backward(); opt.step()
        actor.eval()
        print(f'cloned actor, round {r}: {X.shape[0]} samples, loss {float(loss):.5f}, rows that ended badly while {"the actor" if r > 0 else "the teacher"} flew: {bad_total}', flush=True)
    return actor


def gen_planning_closed(n=80, outer=3, mu_scale=15.0, name='planning_closed_kat.npz', long=False, hi_amp=1.2):
    """PlanningEnv.step CLOSED LOOP (envs/planning_env.py:144-177): the reference env constructs its own PPOActor
    (planning_env.py:41-43) and really loads a state_dict — the one of seeded_actor(), handed over where the reference reads its
    (unshipped) checkpoint file — then runs `outer` macro-steps = 50 x {low_level_obs -> controller -> model.update -> done / reward}
    each, with the recurrent state feeding back.  Recorded per macro-step: the reset draws, everything `step` returns, the env state
    (s, u, targets, step_count, flags, ego_rnn_states); per inner iteration (forward hook on the controller, outputs untouched): the
    controller's input observation, actions and recurrent state.  The actor's state_dict is stored (`sd::*`): consumers run their own
    controller in the loop, nothing is replayed.

    long=True (`planning_closed_long_kat.npz`: n = 64, outer = 20): 1 000 closed-loop inner steps, the horizon north_star
    names.  The per-iteration recordings are dropped (state / recurrent state / masks at the end of every macro-step = every 50th inner step
    stay) and the generator asserts that >= 32 rows fly all 1 000 steps without a termination."""
    if not hasattr(np, 'product'):
        np.product = np.prod  # the reference targets numpy 1.x (algorithms/utils/flatten.py:83)
    import envs.planning_env as pe
    actor = cloned_actor() if long else seeded_actor(mu_scale, 0.0)
    if long:
        hi_amp = 0.5

    def build(sd):
        o_load = torch.load
        torch.load = lambda f, *a, **k: sd if str(f).endswith('actor_latest.pt') else o_load(f, *a, **k)
        try:
            with quiet():
                return pe.PlanningEnv(num_envs=n, config='tracking', model='F16', random_seed=0, device='cpu')
        finally:
            torch.load = o_load

    # centre the head: an un-trimmed F-16 at 1000-1200 ft/s leaves the 300 ft/s^2 Overload limit with two degrees of constant
    # elevator, and a random-init head has a constant offset of that size; subtract the actor's mean action at the env's own
    # first controller input (zero recurrent state) from mu_net's bias, so that what moves the aircraft is the part of the
    # actions that depends on the observation and the recurrent state
    probe = build({k: v.clone() for k, v in actor.state_dict().items()})
    with quiet():
        probe.reset()
    z = torch.zeros(n)
    _, pitch, yaw = probe.model.get_posture()
    if not long:     # (the cloned actor needs no centring: it was fitted to fly)
        with torch.no_grad():
            a0, _, _ = actor(probe.low_level_obs(pitch + z, yaw + z, probe.model.get_vt() + z), torch.zeros((n, 1, 128)), torch.ones((n, 1)), deterministic=True)
            actor.act.action_out.mu_net.fc[0].bias.sub_(a0.mean(0))
    sd = {k: v.detach().clone() for k, v in actor.state_dict().items()}
    env = build(sd)
    for k, v in env.controller.state_dict().items():
        assert torch.equal(v, sd[k]), k            # the reference's own load_state_dict took every tensor
    log = []
    env.controller.register_forward_hook(lambda mod, inp, out: log.append((inp[0].numpy().copy(), out[0].numpy().copy(), out[2].numpy().copy())))
    LL_OBS_AT = (0, 1, 9, 19, 29, 39, 49)
    rng = np.random.RandomState(77)
    hi_actions = rng.uniform(-hi_amp, hi_amp, (outer, n, 3)).astype(np.float32)
    data = {}
    ever_flagged = np.zeros(n, bool)
    for k in range(outer):
        del log[:]
        prev = (env.is_done | env.bad_done | env.exceed_time_limit).numpy().astype(bool)
        with Recorder() as r, quiet():
            obs, rew, done, bad, tmo, _ = env.step(torch.from_numpy(hi_actions[k]))
            draws = r.take()
        rand = [t.numpy() for kind, t in draws if kind == 'rand']
        rand_u = np.zeros((n, 5), np.float32)
        assert len(rand) == 5 and len(log) == 50
        for c, x in enumerate(rand):
            rand_u[prev, c] = x
        data[f'rand_u_{k}'] = rand_u
        if not long:
            data[f'll_obs_{k}'] = np.stack([log[i][0] for i in LL_OBS_AT])                # the controller's input at these inner iterations
            data[f'll_act_{k}'] = np.stack([x[1] for x in log])
            data[f'll_rnn_{k}'] = np.stack([x[2][:, 0] for x in log[9::10]])     # after inner iterations 10, 20, 30, 40, 50
        else:
            data[f'll_act_last_{k}'] = log[-1][1]                                        # the 50th controller call of the macro-step
        ever_flagged |= (done | bad | tmo).numpy().astype(bool)
        data[f's_{k}'] = env.model.s.numpy().copy()
        data[f'u_{k}'] = env.model.u.numpy().copy()
        data[f'tgt_{k}'] = get_tgt(env, 'tracking')
        data[f'step_count_{k}'] = env.step_count.numpy().copy()
        data[f'rnn_{k}'] = env.ego_rnn_states.numpy()[:, 0].copy()
        data[f'obs_{k}'] = obs.numpy().copy()
        data[f'reward_{k}'] = rew.numpy().copy()
        data[f'flags_{k}'] = np.stack([done.numpy(), bad.numpy(), tmo.numpy()]).astype(np.uint8)
        print('planning closed loop, outer', k, 'done', int(done.sum()), 'bad', int(bad.sum()), '|ll action| max', float(np.abs(log[-1][1]).max()),
              'rnn std', float(data[f'rnn_{k}'].std()), 'rows never flagged so far', int((~ever_flagged).sum()), flush=True)
    data['never_flagged'] = ~ever_flagged
    if long:
        assert int((~ever_flagged).sum()) >= 32, int((~ever_flagged).sum())
        assert int(env.step_count[~ever_flagged].min()) == 50 * outer
    np.savez_compressed(os.path.join(OUT, name), hi_actions=hi_actions, ll_obs_at=np.array(LL_OBS_AT), **data,
                        **{'sd::' + k: v.numpy() for k, v in sd.items()})


# ------------------------------------------------------------------------------------------------
# SingleCombat (1v1): the reference's env file is stale (it targets an older BaseEnv API and cannot be
# constructed), but every COMPONENT it calls is importable.  The fixtures below run those components —
# F16Model dynamics + torchdiffeq step, algorithms/pid Controller.stabilize, the termination-condition
# classes, SingleCombatEnv.obs/.reward as unbound methods, utils.{get_AO_TA_R,orientation_fn,...} — in the
# order envs/singlecombat_env.py:207-274 prescribes.  The glue that stands in for the stale lines is listed
# in DESIGN.md §10 (held policy action, direct control write, per-inner-step terminations).
# ------------------------------------------------------------------------------------------------
@contextlib.contextmanager
def pin_mode_combat(mlp_cls):
    """pin_mode + the extra implementation-defined calls of the pairwise geometry/reward functions."""
    names = ('arccos', 'arctanh', 'exp', 'sum')
    orig = {k: getattr(torch, k) for k in names}
    o_norm = torch.linalg.norm
    torch.arccos = lambda x: orig['arccos'](x.double()).float()
    torch.arctanh = lambda x: orig['arctanh'](x.double()).float()
    torch.exp = lambda x: orig['exp'](x.double()).float()

    def tsum(x, *a, **k):
        if x.dtype == torch.float32:
            return orig['sum'](x.double(), *a, **k).float()
        return orig['sum'](x, *a, **k)

    torch.sum = tsum
    torch.linalg.norm = lambda x, *a, **k: o_norm(x.double(), *a, **k).float()
    try:
        with pin_mode(mlp_cls):
            yield
    finally:
        for k in names:
            setattr(torch, k, orig[k])
        torch.linalg.norm = o_norm


def gen_pairwise():
    """Known-answer vectors of the pure pairwise functions (envs/utils/utils.py:156-249)."""
    from utils.utils import get_AO_TA_R, get2d_AO_TA_R, orientation_reward, range_reward, orientation_fn, distance_fn
    rng = np.random.RandomState(41)
    n = 512
    ego_pos = rng.uniform(-2e4, 2e4, (n, 3)).astype(np.float32)
    enm_pos = (ego_pos + rng.normal(0, 1, (n, 3)) * rng.choice([50, 500, 5000, 30000], (n, 1))).astype(np.float32)
    ego_vel = (rng.normal(0, 1, (n, 3)) * 600).astype(np.float32)
    enm_vel = (rng.normal(0, 1, (n, 3)) * 600).astype(np.float32)
    # edge rows: tail chase (AO = 0), head-on, co-located, stationary
    enm_pos[0] = ego_pos[0] + ego_vel[0] * 2
    enm_vel[0] = ego_vel[0]
    enm_pos[1] = ego_pos[1] + ego_vel[1] * 3
    enm_vel[1] = -ego_vel[1]
    enm_pos[2] = ego_pos[2]
    ego_vel[3] = 0
    out = {}
    for tag, ctx in (('', contextlib.nullcontext()), ('_pin', pin_mode_combat(type(make_env('heading', 1).model.dynamics.hifi_F16.Cx_model)))):
        with ctx, quiet():
            T = [torch.from_numpy(x) for x in (ego_pos, enm_pos, ego_vel, enm_vel)]
            AO, TA, R = get_AO_TA_R(*T)
            AO2, TA2, R2, side = get2d_AO_TA_R(*T, return_side=True)
            Rkm = R * 0.3048 / 1000
            out.update({f'AO{tag}': AO, f'TA{tag}': TA, f'R{tag}': R, f'AO2{tag}': AO2, f'TA2{tag}': TA2, f'R2{tag}': R2,
                        f'side{tag}': side, f'orient{tag}': orientation_reward(AO, TA), f'range{tag}': range_reward(3, Rkm),
                        f'ofn{tag}': orientation_fn(AO), f'dfn{tag}': distance_fn(Rkm)})
    np.savez_compressed(os.path.join(OUT, 'pairwise_kat.npz'), ego_pos=ego_pos, enm_pos=enm_pos, ego_vel=ego_vel, enm_vel=enm_vel,
                        **{k: v.numpy() for k, v in out.items()})


def gen_geodesy():
    """Known-answer vectors of the WGS-84 helpers (envs/utils/utils.py:35-142), scalar float64."""
    from utils.utils import geodetic_to_ecef, ecef_to_enu, enu_to_ecef, ecef_to_geodetic, geodetic_to_enu, enu_to_geodetic
    rng = np.random.RandomState(77)
    n = 256
    lat, lon, h = rng.uniform(-89, 89, n), rng.uniform(-180, 180, n), rng.uniform(-500, 30000, n)
    lat0, lon0, h0 = rng.uniform(-80, 80, n), rng.uniform(-180, 180, n), rng.uniform(0, 3000, n)
    enu_in = rng.uniform(-2e5, 2e5, (n, 3))
    lat0[:4], lon0[:4], h0[:4] = [0, 60, 60, -33.9], [0, 120, 120, 151.2], [0, 0, 0, 12.0]     # incl. the recorder's origin
    out = dict(lat=lat, lon=lon, h=h, lat0=lat0, lon0=lon0, h0=h0, enu_in=enu_in)
    out['ecef'] = np.array([geodetic_to_ecef(a, b, c) for a, b, c in zip(lat, lon, h)])
    out['enu'] = np.array([ecef_to_enu(*e, a, b, c) for e, a, b, c in zip(out['ecef'], lat0, lon0, h0)])
    out['enu2'] = np.array([geodetic_to_enu(a, b, c, d, e, f) for a, b, c, d, e, f in zip(lat, lon, h, lat0, lon0, h0)])
    out['ecef_from_enu'] = np.array([enu_to_ecef(*v, a, b, c) for v, a, b, c in zip(enu_in, lat0, lon0, h0)])
    out['geo_from_ecef'] = np.array([ecef_to_geodetic(*e) for e in out['ecef']])
    out['geo_from_enu'] = np.array([enu_to_geodetic(*v, a, b, c) for v, a, b, c in zip(enu_in, lat0, lon0, h0)])
    np.savez_compressed(os.path.join(OUT, 'geodesy_kat.npz'), **out)


PID_STATE = ('roll_dem', 'pitch_dem', 'roll_err', 'roll_int', 'roll_last', 'pitch_err', 'pitch_int', 'pitch_last',
             'yaw_err', 'yaw_int', 'yaw_last')


def _pid_state(c):
    cols = [c.roll_dem, c.pitch_dem]
    for rc in (c.roll_controller, c.pitch_controller, c.yaw_controller):
        cols += [rc.rate_pid.error, rc.rate_pid.integrator, rc.last_out]
    return torch.hstack([x.reshape(-1, 1) for x in cols]).numpy().copy()


def gen_combat(num_envs=24, outer=48, pin=True):
    from types import SimpleNamespace
    from utils.utils import parse_config, get_AO_TA_R, orientation_fn, distance_fn
    from models.F16_model import F16Model
    from torchdiffeq import odeint_adjoint as odeint
    import envs.singlecombat_env as sce
    from algorithms.pid.controller import Controller
    cfg = parse_config('selfplay')
    cfg.init_state = {'init_T': cfg.init_T}   # F16Model reads config.init_state (heading.yaml layout)
    n = 2 * num_envs
    dev = torch.device('cpu')
    with quiet():
        model = F16Model(cfg, n, 'cpu', 0)
        ctrl = Controller(dt=cfg.dt, n=n, device='cpu')
        conds = [sce.Overload(cfg), sce.LowAltitude(cfg), sce.HighSpeed(cfg), sce.LowSpeed(cfg), sce.ExtremeState(cfg),
                 sce.Crash(cfg, 'cpu'), sce.Timeout(cfg), sce.Shutdown(cfg, 'cpu')]
    mlp_cls = type(model.dynamics.hifi_F16.Cx_model)
    E = SimpleNamespace(model=model, n=n, num_envs=num_envs, num_agents=2, device=dev, target_dist=cfg.target_dist,
                        blood=100 * torch.ones(n), step_count=torch.zeros(n, dtype=torch.int64),
                        is_done=torch.ones(n, dtype=torch.bool), bad_done=torch.ones(n, dtype=torch.bool),
                        exceed_time_limit=torch.ones(n, dtype=torch.bool), s=model.s)
    rng = np.random.RandomState(43)
    actions = rng.uniform(-1.3, 1.3, (outer, n, 4)).astype(np.float32)
    actions[:, :, 0] = rng.uniform(0.0, 1.2, (outer, n))
    rand_u = rng.uniform(0, 1, (outer, n, 5)).astype(np.float32)
    data = {}

    def reset_done_envs(ru):  # singlecombat_env.py:207-238 (pairwise: both aircraft of a flagged env)
        flagged = (E.is_done | E.bad_done) | E.exceed_time_limit
        env_reset = torch.any(flagged.reshape(num_envs, 2), dim=-1)
        ra = torch.nonzero(env_reset.repeat_interleave(2)).squeeze(-1)
        U = torch.from_numpy(ru)
        model.s[ra, :] = 0
        model.u[ra, :] = 0
        model.s[ra, 0] = U[ra, 0] * (cfg.max_npos - cfg.min_npos) + cfg.min_npos
        model.s[ra, 1] = U[ra, 1] * (cfg.max_epos - cfg.min_epos) + cfg.min_epos
        model.s[ra, 2] = U[ra, 2] * (cfg.max_altitude - cfg.min_altitude) + cfg.min_altitude
        model.s[ra, 5] = U[ra, 3] * (cfg.max_heading - cfg.min_heading) + cfg.min_heading
        model.s[ra, 6] = U[ra, 4] * (cfg.max_vt - cfg.min_vt) + cfg.min_vt
        model.u[ra, 0] = cfg.init_T
        E.blood[ra] = 100
        E.step_count[ra] = 0
        E.is_done[:] = 0
        E.bad_done[:] = 0
        E.exceed_time_limit[:] = 0

    def outer_step(a_np, ru):
        reset_done_envs(ru)
        if outer_step.first:   # fixture-only state injection so that Crash / Timeout / Shutdown fire within the fixture
            outer_step.first = False
            model.s[1, :3] = model.s[0, :3] + torch.tensor([60.0, -80.0, 40.0])   # pair 0: 108 ft apart -> Crash
            E.step_count[2:4] = 1993                                              # pair 1: Timeout at 2000
            E.blood[4] = -0.5                                                     # pair 2: ego already shot down -> Shutdown bad_done
            E.blood[7] = 0.25                                                     # pair 3: enemy nearly shot down
            model.s[5, :3] = model.s[4, :3] + torch.tensor([900.0, 300.0, 0.0])   # pairs 2,3: close, nose-on geometry
            model.s[5, 5] = 3.0
            model.s[4, 5] = 0.32
            model.s[7, :3] = model.s[6, :3] + torch.tensor([1500.0, 0.0, 50.0])
            model.s[6, 5] = 0.0
            model.s[7, 5] = 0.0
            data['s_init'] = model.s.numpy().copy()
            data['u_init'] = model.u.numpy().copy()
            data['blood_init'] = E.blood.numpy().copy()
            data['step_count_init'] = E.step_count.numpy().copy()
        action = torch.from_numpy(a_np)
        for _ in range(5):                                                        # :243-262
            act = torch.clamp(action, -1, 1)
            ctrl.roll_dem = 0.9 * ctrl.roll_dem + 0.1 * act[:, 1].reshape(-1, 1) * 4 * torch.pi / 9
            ctrl.pitch_dem = 0.9 * ctrl.pitch_dem + 0.1 * act[:, 2].reshape(-1, 1) * torch.pi / 12
            E.s = model.s
            ctrl.stabilize(E)
            T = 0.9 * model.u[:, 0].reshape(-1, 1) + 0.1 * act[:, 0].reshape(-1, 1) * 0.225 * 76300 / 0.3048
            lef = torch.zeros((n, 1))
            model.u = torch.hstack((T, -ctrl.el, -ctrl.ail, -ctrl.rud, lef))
            model.s = odeint(model.dynamics, torch.hstack((model.s, model.u)), torch.tensor([0., model.dt]),
                             method=model.solver)[1, :, :model.num_states]        # F16_model.py:64-67
            E.s = model.s
            E.step_count += 1
            for c in conds:                                                       # task_base.py:75-96, env_base.py:70-75
                bad, done, tmo, _ = c.get_termination(None, E)
                E.is_done = E.is_done + done
                E.bad_done = E.bad_done + bad
                E.exceed_time_limit = E.exceed_time_limit + tmo
        E.es = model.get_extended_state()
        E.velocity = torch.stack(model.get_velocity(), dim=1)
        obs = sce.SingleCombatEnv.obs(E)
        reward = sce.SingleCombatEnv.reward(E)
        ego = torch.arange(num_envs) * 2                                         # :264-271
        enm = ego + 1
        AO, TA, R = get_AO_TA_R(model.s[ego, :3], model.s[enm, :3], E.es[ego, :3], E.es[enm, :3])
        E.blood[enm] -= orientation_fn(AO) * distance_fn(R * 0.3048 / 1000)
        E.blood[ego] -= orientation_fn(torch.pi - TA) * distance_fn(R * 0.3048 / 1000)
        return obs, reward

    outer_step.first = True
    ctx = pin_mode_combat(mlp_cls) if pin else contextlib.nullcontext()
    with ctx, quiet():
        for k in range(outer):
            obs, rew = outer_step(actions[k], rand_u[k])
            data[f's_{k}'] = model.s.numpy().copy()
            data[f'u_{k}'] = model.u.numpy().copy()
            data[f'pid_{k}'] = _pid_state(ctrl)
            data[f'blood_{k}'] = E.blood.numpy().copy()
            data[f'step_count_{k}'] = E.step_count.numpy().copy()
            data[f'obs_{k}'] = obs.numpy().copy()
            data[f'reward_{k}'] = rew.numpy().copy()
            data[f'flags_{k}'] = np.stack([E.is_done.numpy(), E.bad_done.numpy(), E.exceed_time_limit.numpy()]).astype(np.uint8)
    tot = np.sum([data[f'flags_{k}'] for k in range(outer)], axis=(0, 2))
    print('combat', 'pin' if pin else 'plain', 'flag totals done/bad/timeout', tot, 'min blood', min(data[f'blood_{k}'].min() for k in range(outer)))
    np.savez_compressed(os.path.join(OUT, 'combat_kat_pin.npz' if pin else 'combat_kat.npz'), actions=actions, rand_u=rand_u,
                        pid_state_names=np.array(PID_STATE), **data)


def gen_acmi():
    """TacView recording (envs/env_base.py:111-151) of a 1-aircraft ControlEnv for a few steps + enu_to_geodetic
    known answers (envs/utils/utils.py:74-142).  The fixture holds the text the reference wrote and the states it drew."""
    import tempfile
    from utils.utils import enu_to_geodetic
    rng = np.random.RandomState(51)
    pts = np.concatenate([rng.uniform(-3e5, 3e5, (200, 2)), rng.uniform(0, 2e4, (200, 1))], axis=1)
    pts[0] = [0.0, 1e-3, 0.0]
    geo = np.array([enu_to_geodetic(float(e), float(n), float(u), 0, 0, 0) for e, n, u in pts])
    env = make_env('heading', 1, seed=3)
    cwd = os.getcwd()
    states = []
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        os.makedirs('tracks')
        try:
            with quiet():
                env.reset()
                for k in range(4):
                    env.step(torch.tensor([[0.6, 0.2, -0.1, 0.05]]), render=True, count=k)
                    states.append(np.concatenate([env.model.s.numpy()[0], [float(env.step_count[0])]]))
            text = open(os.path.join('tracks', 'F16SimRecording-0.txt.acmi')).read()
        finally:
            os.chdir(cwd)
    np.savez_compressed(os.path.join(OUT, 'acmi_kat.npz'), enu=pts, geodetic=geo, states=np.array(states), text=np.array(text))
    print('acmi: frames', text.count('#'), 'bytes', len(text))


def seeded_actor(mu_scale=60.0, mu_bias=0.1):
    """The reference's PPOActor (algorithms/ppo/ppo_actor.py) with the argument bag of envs/planning_env.py:18-29 and a seeded random
    initialisation whose LayerNorm affine terms / biases are perturbed and whose output layer is re-scaled INSIDE mu_net (the shipped
    gain 0.01 would make every action ~0): a plain PPOActor, no wrapper — its state_dict is the whole controller."""
    if not hasattr(np, 'product'):
        np.product = np.prod
    import gym
    from algorithms.ppo.ppo_actor import PPOActor
    import envs.planning_env as pe
    args = pe.Args() if hasattr(pe, 'Args') else None
    if args is None:
        class A:
            pass
        args = A()
        args.gain, args.hidden_size, args.act_hidden_size, args.activation_id = 0.01, '128 128', '128 128', 1
        args.use_feature_normalization, args.use_recurrent_policy = True, True
        args.recurrent_hidden_size, args.recurrent_hidden_layers, args.use_prior = 128, 1, False
    args.use_prior = False
    torch.manual_seed(123)
    actor = PPOActor(args, gym.spaces.Box(low=-10, high=10, shape=(22,)), gym.spaces.Box(low=-10, high=10, shape=(4,)),
                     device=torch.device('cpu'))
    actor.eval()
    with torch.no_grad():
        for k, v in actor.state_dict().items():       # make LayerNorm affine terms and the head non-trivial
            if k.endswith('norm.weight') or '.fc.2.weight' in k or '.fc.5.weight' in k:
                v.mul_(1.0 + 0.3 * torch.randn_like(v))
            if k.endswith('norm.bias') or '.fc.2.bias' in k or '.fc.5.bias' in k or k.endswith('bias_ih_l0') or k.endswith('bias_hh_l0'):
                v.add_(0.2 * torch.randn_like(v))
        actor.act.action_out.mu_net.fc[0].weight.mul_(mu_scale)
        actor.act.action_out.mu_net.fc[0].bias.add_(mu_bias * torch.randn(4))
    return actor


def gen_actor(n=96, steps=4):
    """PlanningEnv's low-level controller (seeded_actor above): state_dict, inputs, and the actions / rnn states of `steps`
    consecutive deterministic calls."""
    actor = seeded_actor()
    rng = np.random.RandomState(61)
    obs = (rng.normal(0, 1, (steps, n, 22)) * rng.uniform(0.1, 3, (1, 1, 22))).astype(np.float32)
    masks = np.ones((steps, n, 1), np.float32)
    masks[2, ::7] = 0.0                                # a few rows reset their recurrent state
    h = torch.zeros((n, 1, 128))
    acts, hs = [], []
    with torch.no_grad():
        for t in range(steps):
            a, _, h = actor(torch.from_numpy(obs[t]), h, torch.from_numpy(masks[t]), deterministic=True)
            acts.append(a.numpy().copy())
            hs.append(h.numpy().copy())
    sd = {k: v.numpy() for k, v in actor.state_dict().items()}
    np.savez_compressed(os.path.join(OUT, 'actor_kat.npz'), obs=obs, masks=masks, actions=np.stack(acts), rnn=np.stack(hs),
                        **{'sd::' + k: v for k, v in sd.items()})
    print('actor: |action| max', float(np.abs(np.stack(acts)).max()), 'rnn std', float(np.stack(hs).std()))


def seeded_policy(act_dim, obs_dim=22):
    """The reference's PPOPolicy (algorithms/ppo/ppo_policy.py: PPOActor + PPOCritic) in the configuration the training scripts build
    (scripts/train_heading.sh:17 / train_tracking.sh:17: hidden "128 128", act-hidden "128 128", GRU 128 x 1; config.py defaults for the
    rest) with a seeded initialisation whose LayerNorm terms, biases, output layers and log_std are moved off their trivial start values."""
    if not hasattr(np, 'product'):
        np.product = np.prod
    import gym
    from algorithms.ppo.ppo_policy import PPOPolicy

    class A:
        pass
    args = A()
    args.gain, args.hidden_size, args.act_hidden_size, args.activation_id = 0.01, '128 128', '128 128', 1
    args.use_feature_normalization, args.use_recurrent_policy = True, True
    args.recurrent_hidden_size, args.recurrent_hidden_layers, args.use_prior, args.lr = 128, 1, False, 3e-4
    torch.manual_seed(900 + act_dim + (0 if obs_dim == 22 else 100 * obs_dim))
    pol = PPOPolicy(args, gym.spaces.Box(low=-10, high=10, shape=(obs_dim,)), gym.spaces.Box(low=-10, high=10, shape=(act_dim,)), device=torch.device('cpu'))
    pol.prep_rollout()
    with torch.no_grad():
        for net in (pol.actor, pol.critic):
            for k, v in net.state_dict().items():
                if k.endswith('norm.weight') or '.fc.2.weight' in k or '.fc.5.weight' in k:
                    v.mul_(1.0 + 0.3 * torch.randn_like(v))
                if k.endswith('norm.bias') or '.fc.2.bias' in k or '.fc.5.bias' in k or k.endswith('bias_ih_l0') or k.endswith('bias_hh_l0'):
                    v.add_(0.2 * torch.randn_like(v))
        pol.actor.act.action_out.mu_net.fc[0].weight.mul_(40.0)
        pol.actor.act.action_out.mu_net.fc[0].bias.add_(0.1 * torch.randn(act_dim))
        pol.actor.act.action_out.log_std.add_(-0.7 + 0.4 * torch.randn(act_dim))
        pol.critic.value_out.weight.mul_(6.0)
        pol.critic.value_out.bias.add_(0.3)
    return pol


def gen_policy(n=96, steps=5):
    """The rollout policy's inference step, PPOPolicy.get_actions (ppo_policy.py:26-32), for the heading (4 actions) and tracking (3 actions)
    policies: `steps` chained calls (recurrent states fed back, a few masks zeroed) with sampled actions.  The standard normal draws behind
    every sample are stored too — drawn here from the same generator state with normal_(), and checked to reproduce the reference's actions
    exactly as fl(fl(eps * std) + mean) — together with the means (act(..., deterministic=True)) and get_values."""
    out = {}
    # (actions, observations): heading / control 4 x 22, tracking 3 x 22, the 1v1 combat env's policies 4 x 15 (envs/configs/selfplay.yaml)
    for act_dim, obs_dim in ((4, 22), (3, 22), (4, 15)):
        pol = seeded_policy(act_dim, obs_dim)
        rng = np.random.RandomState(70 + act_dim + (0 if obs_dim == 22 else obs_dim))
        obs = (rng.normal(0, 1, (steps, n, obs_dim)) * rng.uniform(0.1, 3, (1, 1, obs_dim))).astype(np.float32)
        masks = np.ones((steps, n, 1), np.float32)
        masks[2, ::5] = 0.0
        masks[4, 1::9] = 0.0
        ha, hc = torch.zeros((n, 1, 128)), torch.zeros((n, 1, 128))
        std = pol.actor.act.action_out.log_std.detach().exp()
        rec = {k: [] for k in ('eps', 'values', 'actions', 'logp', 'ha', 'hc', 'means', 'values_only')}
        with torch.no_grad():
            for t in range(steps):
                o, m = torch.from_numpy(obs[t]), torch.from_numpy(masks[t])
                mean, _ = pol.act(o, ha, m, deterministic=True)
                vonly = pol.get_values(o, hc, m)
                torch.manual_seed(5000 + 10 * act_dim + t)
                eps = torch.empty(n, act_dim).normal_()
                torch.manual_seed(5000 + 10 * act_dim + t)
                values, actions, logp, ha, hc = pol.get_actions(o, ha, hc, m)
                assert torch.equal(actions, eps * std + mean), 'the sample is not fl(fl(eps * std) + mean)'
                assert torch.equal(values, vonly)
                for k, v in (('eps', eps), ('values', values), ('actions', actions), ('logp', logp), ('ha', ha), ('hc', hc), ('means', mean), ('values_only', vonly)):
                    rec[k].append(v.numpy().copy())
        pre = f'a{act_dim}::' if obs_dim == 22 else f'a{act_dim}o{obs_dim}::'
        out.update({pre + 'obs': obs, pre + 'masks': masks, pre + 'std': std.numpy(), pre + 'log_std': pol.actor.act.action_out.log_std.detach().numpy().copy()})
        out.update({pre + k: np.stack(v) for k, v in rec.items()})
        out.update({pre + 'actor::' + k: v.numpy() for k, v in pol.actor.state_dict().items()})
        out.update({pre + 'critic::' + k: v.numpy() for k, v in pol.critic.state_dict().items()})
        print(f'policy {pre} |mean| max', float(np.abs(np.stack(rec['means'])).max()), 'values', float(np.stack(rec['values']).min()),
              float(np.stack(rec['values']).max()), 'logp', float(np.stack(rec['logp']).min()), float(np.stack(rec['logp']).max()), 'std', std.numpy())
    np.savez_compressed(os.path.join(OUT, 'policy_kat.npz'), **out)


def gen_policy_long():
    """PPOPolicy.get_actions (algorithms/ppo/ppo_policy.py:26-32) x 200 CHAINED — both recurrent states fed back, episode
    ends through the masks — for the heading policy (4 actions, 22 observations).  Inputs come from tests/policy_kat.py::long_inputs (a
    seeded formula, not stored); stored: the normal draws behind every sample, the reference's values / actions / log-probabilities /
    recurrent states at every 10th step, and the two state_dicts."""
    sys.path.insert(0, REPO)
    from tests.policy_kat import LONG_EVERY, long_inputs
    pol = seeded_policy(4, 22)
    obs, masks = long_inputs()
    steps, n = obs.shape[:2]
    ha, hc = torch.zeros((n, 1, 128)), torch.zeros((n, 1, 128))
    std = pol.actor.act.action_out.log_std.detach().exp()
    rec = {k: [] for k in ('values', 'actions', 'logp', 'ha', 'hc')}
    eps_all = np.zeros((steps, n, 4), np.float32)
    with torch.no_grad():
        for t in range(steps):
            o, m = torch.from_numpy(obs[t]), torch.from_numpy(masks[t])
            mean, _ = pol.act(o, ha, m, deterministic=True)
            torch.manual_seed(7000 + t)
            eps = torch.empty(n, 4).normal_()
            torch.manual_seed(7000 + t)
            values, actions, logp, ha, hc = pol.get_actions(o, ha, hc, m)
            assert torch.equal(actions, eps * std + mean), 'the sample is not fl(fl(eps * std) + mean)'
            eps_all[t] = eps.numpy()
            if (t + 1) % LONG_EVERY == 0:
                for k, v in (('values', values), ('actions', actions), ('logp', logp), ('ha', ha[:, 0]), ('hc', hc[:, 0])):
                    rec[k].append(v.numpy().copy())
    out = {k: np.stack(v) for k, v in rec.items()}
    out.update(eps=eps_all, std=std.numpy(), log_std=pol.actor.act.action_out.log_std.detach().numpy().copy())
    out.update({'actor::' + k: v.numpy() for k, v in pol.actor.state_dict().items()})
    out.update({'critic::' + k: v.numpy() for k, v in pol.critic.state_dict().items()})
    print('policy long: steps', steps, 'rows', n, 'episode ends', int((masks == 0).sum()), '|ha| max', float(np.abs(out['ha']).max()), 'values', float(out['values'].min()), float(out['values'].max()))
    np.savez_compressed(os.path.join(OUT, 'policy_long_kat.npz'), **out)


def gen_combat_all():
    gen_geodesy()
    gen_pairwise()
    gen_combat(pin=True)
    gen_combat(pin=False)


def gen_buffer():
    """The rollout storage on the policy side of the path: the reference's ReplayBuffer (algorithms/utils/buffer.py:27-256)
    filled through insert() with seeded random data; compute_returns() in its four modes (GAE x proper time limits),
    the normalised advantages, and the mini-batches of recurrent_generator() under a fixed torch seed."""
    if not hasattr(np, 'product'):
        np.product = np.prod
    import gym
    from algorithms.utils.buffer import ReplayBuffer
    T, NT, NA, H = 12, 5, 2, 8
    out = {'T': T, 'n_rollout_threads': NT, 'num_agents': NA, 'hidden': H, 'gamma': 0.99, 'gae_lambda': 0.95,
           'num_mini_batch': 2, 'data_chunk_length': 4, 'torch_seed': 3}
    obs_space, act_space = gym.spaces.Box(low=-10, high=10, shape=(22,)), gym.spaces.Box(low=-10, high=10, shape=(4,))
    rng = np.random.RandomState(2718)
    steps = []
    for t in range(T):
        steps.append(dict(obs=rng.normal(0, 1, (NT, NA, 22)).astype(np.float32), actions=rng.uniform(-1, 1, (NT, NA, 4)).astype(np.float32),
                          rewards=rng.normal(0, 3, (NT, NA, 1)).astype(np.float32),
                          masks=(rng.uniform(0, 1, (NT, NA, 1)) > 0.2).astype(np.float32),
                          action_log_probs=rng.normal(-2, 1, (NT, NA, 1)).astype(np.float32),
                          value_preds=rng.normal(0, 5, (NT, NA, 1)).astype(np.float32),
                          rnn_states_actor=rng.normal(0, 1, (NT, NA, 1, H)).astype(np.float32),
                          rnn_states_critic=rng.normal(0, 1, (NT, NA, 1, H)).astype(np.float32),
                          bad_masks=(rng.uniform(0, 1, (NT, NA, 1)) > 0.15).astype(np.float32)))
    obs0 = rng.normal(0, 1, (NT, NA, 22)).astype(np.float32)
    next_value = rng.normal(0, 5, (NT, NA, 1)).astype(np.float32)
    for k in steps[0]:
        out['in::' + k] = np.stack([s[k] for s in steps])
    out['in::obs0'], out['in::next_value'] = obs0, next_value
    for proper in (False, True):
        for gae in (False, True):
            class A:
                pass
            a = A()
            a.buffer_size, a.n_rollout_threads, a.gamma, a.gae_lambda = T, NT, 0.99, 0.95
            a.use_proper_time_limits, a.use_gae = proper, gae
            a.recurrent_hidden_size, a.recurrent_hidden_layers = H, 1
            buf = ReplayBuffer(a, NA, obs_space, act_space)
            buf.obs[0] = obs0.copy()
            for s in steps:
                buf.insert(**s)
            assert buf.step == 0
            buf.compute_returns(next_value)
            tag = f'proper{int(proper)}_gae{int(gae)}'
            out[f'{tag}::returns'], out[f'{tag}::value_preds'] = buf.returns.copy(), buf.value_preds.copy()
            out[f'{tag}::advantages'] = buf.advantages.copy()
            if gae and not proper:          # the shipped configuration: also the stored fields and the generator's batches
                for f in ('obs', 'actions', 'rewards', 'masks', 'bad_masks', 'action_log_probs', 'rnn_states_actor', 'rnn_states_critic'):
                    out['stored::' + f] = getattr(buf, f).copy()
                torch.manual_seed(3)
                names = ('obs', 'actions', 'masks', 'old_action_log_probs', 'advantages', 'returns', 'value_preds', 'rnn_states_actor',
                         'rnn_states_critic')
                for b, batch in enumerate(ReplayBuffer.recurrent_generator(buf, 2, 4)):
                    for nm, x in zip(names, batch):
                        out[f'batch{b}::{nm}'] = np.asarray(x)
                buf.after_update()
                out['after_update::obs0'], out['after_update::masks0'] = buf.obs[0].copy(), buf.masks[0].copy()
                out['after_update::rnn_states_actor0'] = buf.rnn_states_actor[0].copy()
    np.savez_compressed(os.path.join(OUT, 'buffer_kat.npz'), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == 'buffer':
        gen_buffer()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'planning':
        gen_planning()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'planning_closed':
        gen_planning_closed()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'combat':
        gen_combat_all()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'acmi':
        gen_acmi()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'policy':
        gen_policy()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'actor':
        gen_actor()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'geodesy':
        gen_geodesy()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'getters':
        gen_getters(None)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'model_grid':
        gen_model_grid(make_env('heading', 4))
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'traj_pid_heading':
        gen_traj_pid_heading()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'traj_pid_control':
        gen_traj_pid_control()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'planning_closed_long':
        gen_planning_closed(n=64, outer=20, name='planning_closed_long_kat.npz', long=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'policy_long':
        gen_policy_long()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'traj':
        gen_traj('heading', 256, 1000)     # BASELINE.json configs[0] / SURVEY.md App. D.3 #5: N = 256
        gen_traj('control', 64, 300)
        gen_traj('tracking', 64, 300)
        gen_traj_closed(256, 1000)
        return
    env = make_env('heading', 4)
    gen_aero(env)
    gen_model_grid(env)
    gen_nlplant(env)
    gen_getters(env)
    for task in ('heading', 'control', 'tracking'):
        gen_step_kat(task)
    gen_step_kat('heading', solver='rk4', tag='step_kat_heading_rk4')
    gen_traj('heading', 256, 1000)
    gen_traj('control', 64, 300)
    gen_traj('tracking', 64, 300)
    gen_traj_closed(256, 1000)
    gen_recorded_episode()
    gen_planning()
    gen_combat_all()
    gen_acmi()
    gen_actor()
    gen_planning_closed()
    gen_buffer()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == '__main__':
    main()
