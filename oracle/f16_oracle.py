"""ctypes/numpy front-end of the CPU parity oracle (oracle/f16_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, `__graft_entry__.smoke()` and bench.py's
`cpu_baseline` leg — never by `neuralplane_amd`.  Arrays use the REFERENCE's layout
(`s[n,12]`, `u[n,5]`, `tgt[n,3]`, `obs[n,22]`, row-major), see f16_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import yaml

_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(_HERE)
_SO = os.path.join(_HERE, '_build', 'libf16oracle.so')
DEFAULT_BLOB = os.path.join(_REPO, 'neuralplane_amd', 'assets', 'f16_aero_mlp.bin')
CONFIG_DIR = os.path.join(_REPO, 'neuralplane_amd', 'envs', 'configs')

TASKS = {'heading': 0, 'control': 1, 'tracking': 2}
SOLVERS = {'euler': 0, 'rk4': 1}
MODE_MLP_F64 = 1
MODE_LIBM = 2
MODE_PWL = 4
MODE_BIAS_LAST = 16   # EXPERIMENT, never the spec: the aero MLPs with the bias added last (f16_oracle.h)
MODE_DIV_IEEE = 8   # constant divisions as plain IEEE x / c instead of the Markstein sequence (identical results; tests run both)


class Cfg(C.Structure):
    _fields_ = [('task', C.c_int32), ('solver', C.c_int32),
                ('dt', C.c_double), ('airspeed', C.c_double), ('noise_scale', C.c_double),
                ('altitude_limit', C.c_double), ('acceleration_limit', C.c_double),
                ('max_velocity', C.c_double), ('min_velocity', C.c_double),
                ('min_alpha', C.c_double), ('max_alpha', C.c_double),
                ('min_beta', C.c_double), ('max_beta', C.c_double),
                ('max_check_interval', C.c_int64), ('min_check_interval', C.c_int64),
                ('init_T', C.c_double), ('max_altitude', C.c_double), ('min_altitude', C.c_double),
                ('max_vt', C.c_double), ('min_vt', C.c_double),
                ('max_heading_increment', C.c_double), ('max_pitch_increment', C.c_double),
                ('max_velocities_u_increment', C.c_double),
                ('max_distance', C.c_double), ('min_distance', C.c_double)]


class Airframe(C.Structure):
    """f16o_airframe (f16_oracle.h); all zero = the F-16."""
    _fields_ = [(k, C.c_double) for k in ('g', 'mass', 'B', 'S', 'cbar', 'xcgr', 'xcg', 'Heng', 'Jy', 'Jxz', 'Jz', 'Jx', 'ail_ref', 'rud_ref',
                                          'atm_lapse', 'atm_exp', 'rho0', 'lag_keep', 'lag_new', 'thrust_frac', 'thrust_max', 'thrust_unit')] + [('surf_max', C.c_double * 3)]


def make_airframe(lib, overrides):
    """The F-16 defaults with {field: value} on top (None / {} -> None = keep the model's defaults)."""
    if not overrides:
        return None
    a = Airframe()
    lib.f16o_airframe_default(C.byref(a))
    for k, v in dict(overrides).items():
        if k == 'surf_max':
            for j, x in enumerate(v):
                a.surf_max[j] = float(x)
        else:
            assert hasattr(a, k), k
            setattr(a, k, float(v))
    return a


def build(force=False):
    """Compile oracle/_build/libf16oracle.so with gcc (Makefile)."""
    src_m = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ('f16_oracle.c', 'f16_combat.inc', 'f16_actor.inc', 'f16_actor_i8.inc', 'f16_rollout.inc', 'f16_oracle.h', 'Makefile'))
    if os.environ.get('F16O_SO'):      # e.g. the sanitizer build (`make -C oracle asan-test`)
        return os.environ['F16O_SO']
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < src_m:
        subprocess.run(['make', '-C', _HERE], check=True, stdout=subprocess.DEVNULL)
    return _SO


def load_cfg(task, solver=None, overrides=None):
    """Scenario YAML -> Cfg, using the same defaults as the reference's getattr() calls."""
    with open(os.path.join(CONFIG_DIR, f'{task}.yaml')) as f:
        y = yaml.safe_load(f)
    y.update(overrides or {})
    g = y.get
    c = Cfg()
    c.task = TASKS[task]
    c.solver = SOLVERS[solver or g('solver', 'euler')]
    c.dt = g('dt', 0.02)
    c.airspeed = g('airspeed', 0)
    c.noise_scale = g('noise_scale', 0.01)
    c.altitude_limit = g('altitude_limit', 2500.0)
    c.acceleration_limit = g('acceleration_limit', 300.0)
    c.max_velocity = g('max_velocity', 3)
    c.min_velocity = g('min_velocity', 0.01)
    c.min_alpha, c.max_alpha = g('min_alpha', -20), g('max_alpha', 45)
    c.min_beta, c.max_beta = g('min_beta', -30), g('max_beta', 30)
    c.max_check_interval = g('max_check_interval', 1500)
    c.min_check_interval = g('min_check_interval', 300)
    c.init_T = y['init_state']['init_T']
    c.max_altitude, c.min_altitude = g('max_altitude', 20000), g('min_altitude', 19000)
    c.max_vt, c.min_vt = g('max_vt', 1200), g('min_vt', 1000)
    c.max_heading_increment = g('max_heading_increment', 0.3)
    c.max_pitch_increment = g('max_pitch_increment', 0.3)
    c.max_velocities_u_increment = g('max_velocities_u_increment', 100)
    c.max_distance, c.min_distance = g('max_distance', 2000), g('min_distance', 2000)
    return c


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=C.c_float):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


class Oracle:
    """One loaded model + scenario config.  State lives in the caller's numpy arrays."""

    def __init__(self, task='heading', solver=None, overrides=None, blob_path=DEFAULT_BLOB, mode=0, threads=None):
        self.lib = C.CDLL(build())
        L = self.lib
        L.f16o_model_load.restype = C.c_void_p
        L.f16o_model_load.argtypes = [C.c_char_p, C.c_size_t]
        L.f16o_model_free.argtypes = [C.c_void_p]
        L.f16o_tan.restype = C.c_float
        L.f16o_tan.argtypes = [C.c_float]
        L.f16o_pow.restype = C.c_float
        L.f16o_pow.argtypes = [C.c_float, C.c_float]
        L.f16o_wrap_pi.restype = C.c_float
        L.f16o_wrap_pi.argtypes = [C.c_float]
        L.f16o_sincos.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.f16o_aero.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float)]
        with open(blob_path, 'rb') as f:
            blob = f.read()
        self.model = L.f16o_model_load(blob, len(blob))
        if not self.model:
            raise RuntimeError('f16o_model_load failed')
        self.task = task
        overrides = dict(overrides or {})
        af = make_airframe(L, overrides.pop('airframe', None))    # scenario key `airframe`: {np_f16_airframe field: value}
        if af is not None:
            L.f16o_model_set_airframe(C.c_void_p(self.model), C.byref(af))
        self.cfg = load_cfg(task, solver, overrides)
        self.mode = mode
        if threads:
            L.f16o_set_threads(int(threads))
        self.threads = L.f16o_num_threads()

    def __del__(self):
        try:
            self.lib.f16o_model_free(self.model)
        except Exception:
            pass

    def _set_mode(self):
        self.lib.f16o_set_mode(self.mode)

    # ---- elementary pieces -------------------------------------------------------------
    def sincos(self, x):
        self._set_mode()
        x = _f32(x).reshape(-1)
        s = np.empty_like(x)
        c = np.empty_like(x)
        a, b = C.c_float(), C.c_float()
        for i, v in enumerate(x):
            self.lib.f16o_sincos(C.c_float(v), C.byref(a), C.byref(b))
            s[i], c[i] = a.value, b.value
        return s, c

    def tan(self, x):
        self._set_mode()
        return np.array([self.lib.f16o_tan(C.c_float(v)) for v in _f32(x).reshape(-1)], dtype=np.float32)

    def pow(self, x, y):
        self._set_mode()
        return np.array([self.lib.f16o_pow(C.c_float(v), C.c_float(y)) for v in _f32(x).reshape(-1)], dtype=np.float32)

    def wrap_pi(self, x):
        return np.array([self.lib.f16o_wrap_pi(C.c_float(v)) for v in _f32(x).reshape(-1)], dtype=np.float32)

    def philox(self, ctr, key):
        out = (C.c_uint32 * 4)()
        self.lib.f16o_philox4x32((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out)
        return list(out)

    def rng_uniforms(self, seed, call_idx, row):
        out = (C.c_float * 8)()
        self.lib.f16o_rng_uniforms(C.c_uint64(seed), C.c_uint64(call_idx), C.c_int64(row), out)
        return np.array(out, dtype=np.float32)

    def rng_normals(self, seed, call_idx, row):
        out = (C.c_float * 22)()
        self.lib.f16o_rng_normals(C.c_uint64(seed), C.c_uint64(call_idx), C.c_int64(row), out)
        return np.array(out, dtype=np.float32)

    def aero(self, alpha_deg, beta_deg, el):
        self._set_mode()
        a, b, e = (_f32(v).reshape(-1) for v in (alpha_deg, beta_deg, el))
        out = np.empty((a.size, 43), dtype=np.float32)
        row = (C.c_float * 43)()
        for i in range(a.size):
            self.lib.f16o_aero(C.c_void_p(self.model), C.c_float(a[i]), C.c_float(b[i]), C.c_float(e[i]), row)
            out[i] = row
        return out

    def nlplant(self, x17):
        self._set_mode()
        x = _f32(x17)
        out = np.empty((x.shape[0], 12), dtype=np.float32)
        self.lib.f16o_nlplant(C.c_void_p(self.model), C.c_int64(x.shape[0]), _p(x), _p(out))
        return out

    def get_acceleration(self, s, u):
        self._set_mode()
        s, u = _f32(s), _f32(u)
        out = np.empty((s.shape[0], 3), dtype=np.float32)
        self.lib.f16o_get_acceleration(C.c_void_p(self.model), C.c_int64(s.shape[0]), _p(s), _p(u), _p(out))
        return out

    def get_accels(self, s, u):
        self._set_mode()
        s, u = _f32(s), _f32(u)
        out = np.empty((s.shape[0], 3), dtype=np.float32)
        self.lib.f16o_get_accels(C.c_void_p(self.model), C.c_int64(s.shape[0]), _p(s), _p(u), _p(out))
        return out

    def get_eas2tas(self, s):
        self._set_mode()
        s = _f32(s)
        out = np.empty(s.shape[0], dtype=np.float32)
        self.lib.f16o_get_eas2tas(C.c_void_p(self.model), C.c_int64(s.shape[0]), _p(s), _p(out))
        return out

    def get_atmos(self, s):
        """F16Model.get_atmos (F16_model.py:183-198) -> [n, 3] (mach, qbar, ps)."""
        self._set_mode()
        s = _f32(s)
        out = np.empty((s.shape[0], 3), dtype=np.float32)
        self.lib.f16o_get_atmos(C.c_void_p(self.model), C.c_int64(s.shape[0]), _p(s), _p(out))
        return out

    # ---- env level -----------------------------------------------------------------------
    @staticmethod
    def new_state(n):
        """Arrays as BaseEnv.__init__ leaves them: zeros, all three flags set (env_base.py:28-33)."""
        return dict(s=np.zeros((n, 12), np.float32), u=np.zeros((n, 5), np.float32),
                    tgt=np.zeros((n, 3), np.float32), step_count=np.zeros(n, np.int64),
                    done=np.ones(n, np.uint8), bad=np.ones(n, np.uint8), timeout=np.ones(n, np.uint8))

    def reset(self, st, rand_u=None, noise=None, seed=0, call_idx=0, row0=0, want_obs=True):
        self._set_mode()
        n = st['s'].shape[0]
        obs = np.empty((n, 22), np.float32) if want_obs else None
        ru = None if rand_u is None else _f32(rand_u)
        nz = None if noise is None else _f32(noise)
        rc = self.lib.f16o_reset(C.c_void_p(self.model), C.byref(self.cfg), C.c_int64(n), _p(st['s']), _p(st['u']),
                                 _p(st['tgt']), _p(st['step_count'], C.c_int64), _p(st['done'], C.c_uint8),
                                 _p(st['bad'], C.c_uint8), _p(st['timeout'], C.c_uint8), _p(ru), _p(nz),
                                 C.c_uint64(seed), C.c_uint64(call_idx), C.c_int64(row0), _p(obs))
        assert rc == 0
        return obs

    def termination_reasons(self, st):
        """uint8[n] bit mask of the conditions that fire at the state in `st` (bits: overload, low_altitude, high_speed,
        low_speed, extreme_state, unreach (bad), target reached (done))."""
        self._set_mode()
        n = st['s'].shape[0]
        out = np.zeros(n, np.uint8)
        self.lib.f16o_termination_reasons(C.c_void_p(self.model), C.byref(self.cfg), C.c_int64(n), _p(st['s']), _p(st['u']),
                                          _p(st['tgt']), _p(st['step_count'], C.c_int64), _p(out, C.c_uint8))
        return out

    def lowlevel_obs(self, st, tgt3):
        """PlanningEnv.low_level_obs for tgt3[n,3] = (target_pitch, target_heading, target_vt)."""
        self._set_mode()
        t = _f32(tgt3)
        n = st['s'].shape[0]
        obs = np.empty((n, 22), np.float32)
        self.lib.f16o_lowlevel_obs(C.c_void_p(self.model), C.byref(self.cfg), C.c_int64(n), _p(st['s']), _p(st['u']), _p(t), _p(obs))
        return obs

    def step_inner(self, st, action, noise=None, seed=0, call_idx=0, row0=0):
        """One low-level iteration of PlanningEnv.step (no auto-reset, flagged rows frozen, flags accumulate)."""
        self._set_mode()
        n = st['s'].shape[0]
        a = _f32(action)
        obs = np.empty((n, 22), np.float32)
        rew = np.empty(n, np.float32)
        nz = None if noise is None else _f32(noise)
        rc = self.lib.f16o_step_inner(C.c_void_p(self.model), C.byref(self.cfg), C.c_int64(n), _p(st['s']), _p(st['u']),
                                      _p(st['tgt']), _p(st['step_count'], C.c_int64), _p(st['done'], C.c_uint8),
                                      _p(st['bad'], C.c_uint8), _p(st['timeout'], C.c_uint8), _p(a), C.c_int64(a.shape[1]),
                                      _p(nz), C.c_uint64(seed), C.c_uint64(call_idx), C.c_int64(row0), _p(obs), _p(rew))
        assert rc == 0
        return obs, rew, st['done'].copy(), st['bad'].copy(), st['timeout'].copy()

    def update(self, st, action):
        """F16Model.update(action) alone (F16_model.py:51-67): clamp, control lag, one integrator step for every row."""
        self._set_mode()
        n = st['s'].shape[0]
        a = _f32(action)
        assert a.ndim == 2 and a.shape[0] == n and a.shape[1] >= 4
        rc = self.lib.f16o_update(C.c_void_p(self.model), C.byref(self.cfg), C.c_int64(n), _p(st['s']), _p(st['u']), _p(a), C.c_int64(a.shape[1]))
        assert rc == 0

    def model_reset(self, st, rand_u=None, seed=0, call_idx=0, row0=0):
        """F16Model.reset(env) alone (F16_model.py:33-45): s, u of the flagged rows; targets, counters and flags stay."""
        n = st['s'].shape[0]
        ru = None if rand_u is None else _f32(rand_u)
        rc = self.lib.f16o_model_reset(C.byref(self.cfg), C.c_int64(n), _p(st['s']), _p(st['u']), _p(st['done'], C.c_uint8), _p(st['bad'], C.c_uint8),
                                       _p(st['timeout'], C.c_uint8), _p(ru), C.c_uint64(seed), C.c_uint64(call_idx), C.c_int64(row0))
        assert rc == 0

    def step(self, st, action, rand_u=None, noise=None, seed=0, call_idx=0, row0=0):
        self._set_mode()
        n = st['s'].shape[0]
        a = _f32(action)
        assert a.ndim == 2 and a.shape[0] == n and a.shape[1] >= 4
        obs = np.empty((n, 22), np.float32)
        rew = np.empty(n, np.float32)
        ru = None if rand_u is None else _f32(rand_u)
        nz = None if noise is None else _f32(noise)
        rc = self.lib.f16o_step(C.c_void_p(self.model), C.byref(self.cfg), C.c_int64(n), _p(st['s']), _p(st['u']),
                                _p(st['tgt']), _p(st['step_count'], C.c_int64), _p(st['done'], C.c_uint8),
                                _p(st['bad'], C.c_uint8), _p(st['timeout'], C.c_uint8), _p(a),
                                C.c_int64(a.shape[1]), _p(ru), _p(nz), C.c_uint64(seed), C.c_uint64(call_idx),
                                C.c_int64(row0), _p(obs), _p(rew))
        assert rc == 0
        return obs, rew, st['done'].copy(), st['bad'].copy(), st['timeout'].copy()


# =====================================================================================================
# SingleCombat 1v1 (f16_combat.inc)
# =====================================================================================================
class PidGains(C.Structure):
    _fields_ = [(k, C.c_double) for k in ('Kp', 'Ki', 'Kd', 'Kff', 'Kimax', 'tau', 'rmax_pos', 'rmax_neg')]


class CombatCfg(C.Structure):
    _fields_ = [('solver', C.c_int32), ('inner_steps', C.c_int32), ('dt', C.c_double), ('airspeed', C.c_double),
                ('altitude_limit', C.c_double), ('acceleration_limit', C.c_double), ('max_velocity', C.c_double),
                ('min_velocity', C.c_double), ('min_alpha', C.c_double), ('max_alpha', C.c_double),
                ('min_beta', C.c_double), ('max_beta', C.c_double), ('distance_limit', C.c_double),
                ('max_steps', C.c_int64), ('init_T', C.c_double), ('target_dist', C.c_double),
                ('max_altitude', C.c_double), ('min_altitude', C.c_double), ('max_vt', C.c_double), ('min_vt', C.c_double),
                ('max_heading', C.c_double), ('min_heading', C.c_double), ('max_npos', C.c_double), ('min_npos', C.c_double),
                ('max_epos', C.c_double), ('min_epos', C.c_double),
                ('roll', PidGains), ('pitch', PidGains), ('yaw', PidGains),
                ('roll_ff', C.c_double), ('gravity', C.c_double), ('airspeed_min', C.c_double), ('airspeed_max', C.c_double)]


def load_combat_cfg(config='selfplay', solver=None, overrides=None):
    """selfplay.yaml + pid/*.yaml -> CombatCfg with the defaults of singlecombat_env.py:29-44 / the condition classes."""
    with open(os.path.join(CONFIG_DIR, f'{config}.yaml')) as f:
        y = yaml.safe_load(f)
    y.update(overrides or {})
    g = y.get
    c = CombatCfg()
    c.solver = SOLVERS[solver or g('solver', 'euler')]
    c.inner_steps = 5
    c.dt, c.airspeed = g('dt', 0.02), g('airspeed', 0)
    c.altitude_limit, c.acceleration_limit = g('altitude_limit', 2500.0), g('acceleration_limit', 300.0)
    c.max_velocity, c.min_velocity = g('max_velocity', 3), g('min_velocity', 0.01)
    c.min_alpha, c.max_alpha, c.min_beta, c.max_beta = g('min_alpha', -20), g('max_alpha', 45), g('min_beta', -30), g('max_beta', 30)
    c.distance_limit, c.max_steps = g('distance_limit', 200), g('max_steps', 500)
    c.init_T, c.target_dist = g('init_T', 2000), g('target_dist', 3)
    c.max_altitude, c.min_altitude = g('max_altitude', 20000), g('min_altitude', 19000)
    c.max_vt, c.min_vt = g('max_vt', 1200), g('min_vt', 1000)
    c.max_heading, c.min_heading = g('max_heading', 0.5), g('min_heading', -0.5)
    c.max_npos, c.min_npos, c.max_epos, c.min_epos = g('max_npos', 5000), g('min_npos', -5000), g('max_epos', 5000), g('min_epos', -5000)
    for name in ('roll', 'pitch', 'yaw'):
        with open(os.path.join(CONFIG_DIR, 'pid', f'{name}controller.yaml')) as f:
            p = yaml.safe_load(f)
        gains = getattr(c, name)
        for k in ('Kp', 'Ki', 'Kd', 'Kff', 'Kimax', 'tau'):
            setattr(gains, k, p[k])
        gains.rmax_pos, gains.rmax_neg = p.get('rmax_pos', 0), p.get('rmax_neg', 0)
        if name == 'pitch':
            c.roll_ff, c.gravity = p['roll_ff'], p['gravity']
    c.airspeed_min, c.airspeed_max = 100, 2300   # Controller.__init__ defaults (controller.py:15)
    return c


class CombatOracle(Oracle):
    """1v1 combat macro-step on top of the same model blob.  Rows 2k / 2k+1 = ego / enemy of env k."""

    def __init__(self, config='selfplay', solver=None, overrides=None, blob_path=DEFAULT_BLOB, mode=0, threads=None):
        overrides = dict(overrides or {})
        super().__init__('heading', None, {'airframe': overrides.pop('airframe', None)}, blob_path, mode, threads)
        self.ccfg = load_combat_cfg(config, solver, overrides)
        for fn in ('f16o_acos', 'f16o_atanh', 'f16o_exp'):
            getattr(self.lib, fn).restype = C.c_float
            getattr(self.lib, fn).argtypes = [C.c_float]

    @staticmethod
    def new_state(num_envs):
        n = 2 * num_envs
        return dict(s=np.zeros((n, 12), np.float32), u=np.zeros((n, 5), np.float32), pid=np.zeros((n, 11), np.float32),
                    blood=np.full(n, 100, np.float32), step_count=np.zeros(n, np.int64),
                    done=np.ones(n, np.uint8), bad=np.ones(n, np.uint8), timeout=np.ones(n, np.uint8))

    def unary(self, name, x):
        self._set_mode()
        fn = getattr(self.lib, 'f16o_' + name)
        return np.array([fn(C.c_float(v)) for v in _f32(x).reshape(-1)], dtype=np.float32)

    def pairwise(self, ego_pos, enm_pos, ego_vel, enm_vel):
        self._set_mode()
        a, b, c, d = (_f32(v) for v in (ego_pos, enm_pos, ego_vel, enm_vel))
        out = np.empty((a.shape[0], 11), np.float32)
        self.lib.f16o_pairwise(C.c_int64(a.shape[0]), _p(a), _p(b), _p(c), _p(d), C.c_float(self.ccfg.target_dist), _p(out))
        return out

    def stabilize(self, s, pid, first):
        self._set_mode()
        s = _f32(s)
        out = np.empty((s.shape[0], 3), np.float32)
        self.lib.f16o_stabilize(C.c_void_p(self.model), C.byref(self.ccfg), C.c_int64(s.shape[0]), _p(s), _p(pid), C.c_int(int(first)), _p(out))
        return out

    def combat_reset(self, st, rand_u=None, seed=0, call_idx=0, env0=0):
        self._set_mode()
        n = st['s'].shape[0]
        obs = np.empty((n, 15), np.float32)
        ru = None if rand_u is None else _f32(rand_u)
        rc = self.lib.f16o_combat_reset(C.c_void_p(self.model), C.byref(self.ccfg), C.c_int64(n // 2), _p(st['s']), _p(st['u']),
                                        _p(st['blood']), _p(st['step_count'], C.c_int64), _p(st['done'], C.c_uint8),
                                        _p(st['bad'], C.c_uint8), _p(st['timeout'], C.c_uint8), _p(ru), C.c_uint64(seed),
                                        C.c_uint64(call_idx), C.c_int64(env0), _p(obs))
        assert rc == 0
        return obs

    def combat_step(self, st, action, rand_u=None, pid_first=False, seed=0, call_idx=0, env0=0, term_counts=None):
        self._set_mode()
        n = st['s'].shape[0]
        a = _f32(action)
        assert a.ndim == 2 and a.shape[0] == n and a.shape[1] >= 4
        obs = np.empty((n, 15), np.float32)
        rew = np.empty(n, np.float32)
        ru = None if rand_u is None else _f32(rand_u)
        rc = self.lib.f16o_combat_step(C.c_void_p(self.model), C.byref(self.ccfg), C.c_int64(n // 2), _p(st['s']), _p(st['u']),
                                       _p(st['pid']), _p(st['blood']), _p(st['step_count'], C.c_int64),
                                       _p(st['done'], C.c_uint8), _p(st['bad'], C.c_uint8), _p(st['timeout'], C.c_uint8), _p(a),
                                       C.c_int64(a.shape[1]), _p(ru), C.c_int(int(pid_first)), C.c_uint64(seed),
                                       C.c_uint64(call_idx), C.c_int64(env0), _p(obs), _p(rew), _p(term_counts, C.c_uint32))
        assert rc == 0
        return obs, rew, st['done'].copy(), st['bad'].copy(), st['timeout'].copy()


# =====================================================================================================
# PlanningEnv's frozen low-level controller (f16_actor.inc)
# =====================================================================================================
class ActorOracle:
    """PPOActor.forward(deterministic=True) for the packed weights of neuralplane_amd.actor.pack_ppo_actor.
    numerics: 'fp32' (f16_actor.inc: sequential fmaf chains) or 'i8' (f16_actor_i8.inc: block fixed point, the second spec)."""

    def __init__(self, weights, numerics='fp32'):
        self.lib = C.CDLL(build())
        self.w = _f32(weights).reshape(-1)
        assert self.w.size == self.lib.f16o_actor_num_floats(), (self.w.size, self.lib.f16o_actor_num_floats())
        assert numerics in ('fp32', 'i8'), numerics
        self.numerics = numerics

    def quantised_weights(self, layer):
        """(wq[n_out, n_in] int32, ew[n_out] int32) of layer 0 L1, 1 L2, 2 GI, 3 GH, 4 A1, 5 A2 as the i8 restatement computes them."""
        n_in, n_out = (22, 128, 128, 128, 128, 128)[layer], (128, 128, 384, 384, 128, 128)[layer]
        wq, ew = np.empty((n_out, n_in), np.int32), np.empty(n_out, np.int32)
        assert self.lib.f16o_actor_i8_weights(_p(self.w), C.c_int(layer), _p(wq, C.c_int32), _p(ew, C.c_int32)) == 0
        return wq, ew

    def forward(self, obs, h, masks):
        obs, h = _f32(obs), _f32(h).reshape(-1, 128)
        m = _f32(masks).reshape(-1)
        n = obs.shape[0]
        act = np.empty((n, 4), np.float32)
        h_out = np.empty((n, 128), np.float32)
        if self.numerics == 'i8':
            assert self.lib.f16o_actor_i8_forward(_p(self.w), C.c_int64(n), _p(obs), _p(h), _p(m), _p(act), _p(h_out)) == 0
        else:
            self.lib.f16o_actor_forward(_p(self.w), C.c_int64(n), _p(obs), _p(h), _p(m), _p(act), _p(h_out))
        return act, h_out


class PolicyOracle:
    """PPOPolicy.get_actions / get_values / act (algorithms/ppo/ppo_policy.py:26-57) for two packed networks (f16_actor.inc,
    f16o_policy_act; numerics='i8': f16_actor_i8.inc, f16o_policy_act_i8): `actor_w`, `critic_w` as neuralplane_amd.policy packs them,
    std / log_std [act_dim] float32."""
    ACTOR, CRITIC, DETERMINISTIC = 1, 2, 4

    def __init__(self, actor_w, critic_w, std, log_std, numerics='fp32', obs_dim=22):
        self.lib = C.CDLL(build())
        assert numerics in ('fp32', 'i8'), numerics
        self.numerics, self.obs_dim = numerics, int(obs_dim)
        self.wa, self.wc = _f32(actor_w).reshape(-1), _f32(critic_w).reshape(-1)
        assert self.wa.size == self.wc.size == self.lib.f16o_actor_num_floats()
        self.std, self.log_std = _f32(std).reshape(-1), _f32(log_std).reshape(-1)
        self.act_dim = self.std.size
        assert 1 <= self.act_dim <= 4 and self.log_std.size == self.act_dim

    def run(self, obs, ha, hc, masks, noise=None, flags=3):
        obs, ha, hc = _f32(obs).reshape(-1, self.obs_dim), _f32(ha).reshape(-1, 128), _f32(hc).reshape(-1, 128)
        m = _f32(masks).reshape(-1)
        n, A = obs.shape[0], self.act_dim
        noise = np.zeros((n, A), np.float32) if noise is None else _f32(noise).reshape(n, A)
        values, actions, logp = np.zeros((n, 1), np.float32), np.zeros((n, A), np.float32), np.zeros((n, 1), np.float32)
        ha_out, hc_out = ha.copy(), hc.copy()
        fn = self.lib.f16o_policy_act_i8 if self.numerics == 'i8' else self.lib.f16o_policy_act
        rc = fn(_p(self.wa), _p(self.wc), _p(self.std), _p(self.log_std), C.c_int64(n), C.c_int(self.obs_dim), C.c_int(A), C.c_int(flags), _p(obs), _p(ha),
                _p(hc), _p(m), _p(noise), _p(values), _p(actions), _p(logp), _p(ha_out), _p(hc_out))
        assert self.numerics == 'fp32' or rc == 0
        return values, actions, logp, ha_out, hc_out


# ---------------------------------------------------------------------------------------------
# Rollout storage: ReplayBuffer.compute_returns (f16_rollout.inc)
# ---------------------------------------------------------------------------------------------
def rollout_returns(rewards, value_preds, masks, bad_masks, next_value, gamma, gae_lambda, use_gae, proper):
    """rewards [T, ...], value_preds / masks / bad_masks [T+1, ...] (trailing dims flattened to N columns), next_value [...].
    Returns (returns[T+1, ...], value_preds[T+1, ...]) as the reference's buffer holds them after compute_returns()."""
    lib = C.CDLL(build())
    shape = np.asarray(value_preds).shape
    T = shape[0] - 1
    r = _f32(rewards).reshape(T, -1).copy()
    N = r.shape[1]
    v = _f32(value_preds).reshape(T + 1, N).copy()
    m = _f32(masks).reshape(T + 1, N).copy()
    b = _f32(bad_masks).reshape(T + 1, N).copy()
    nv = _f32(next_value).reshape(N).copy()
    ret = np.zeros((T + 1, N), np.float32)
    lib.f16o_rollout_returns(C.c_int64(T), C.c_int64(N), C.c_double(gamma), C.c_double(gae_lambda), C.c_int(int(use_gae)), C.c_int(int(proper)),
                             _p(r), _p(v), _p(m), _p(b), _p(nv), _p(ret))
    return ret.reshape(shape), v.reshape(shape)
