"""F16Batch — device-resident state of N F-16 aircraft and the fused reset/step launches.

This is the thin host layer between the reference-shaped Python env surface (neuralplane_amd/envs)
and the C ABI (include/neuralplane_amd.h).  PyTorch-ROCm is used for what it is good at — device
memory, streams, the caching allocator — and nothing else: every number is produced by the HIP
kernels in csrc/.  State is structure-of-arrays (`s[12,n]`, `u[5,n]`, `tgt[3,n]`); the `[n,k]`
tensors the reference exposes (`model.s`, `model.u`) are transposed *views* of these buffers.
"""
import ctypes as C
import os

import torch

from . import _lib

ASSET_BLOB = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'assets', 'f16_aero_mlp.bin')
NUM_DERIVED = 23
# Identifier of the numerics spec the kernels implement: bumped whenever a change alters the bits of a run (round 2:
# observation-noise generator v2, packed output layers, Markstein constant divisions = 2).  Checkpoints carry it together with the
# library's ABI version; load_state_dict warns when a checkpoint was written under another spec — it still loads, but the continued
# trajectories will not reproduce the recorded ones bit for bit.
NUMERICS_SPEC = 2
NUM_NETS = 43
NUM_CACHED = 14


def cfg_from_config(config, task, solver=None, aero_1d_tables=None):
    """Scenario attribute bag (parse_config) -> np_f16_cfg, with the reference's getattr defaults."""
    g = lambda k, d: getattr(config, k, d)  # noqa: E731
    c = _lib.NpF16Cfg()
    c.task = _lib.TASKS[task]
    sol = solver or g('solver', 'euler')
    if sol not in _lib.SOLVERS:
        raise NotImplementedError(f"solver '{sol}' (supported: euler, rk4)")
    c.solver = _lib.SOLVERS[sol]
    c.dt = g('dt', 0.02)                                    # F16_model.py:15
    c.airspeed = g('airspeed', 0)                           # F16_model.py:17
    c.noise_scale = g('noise_scale', 0.01)                  # heading_task.py:32
    c.altitude_limit = g('altitude_limit', 2500.0)          # low_altitude.py:13
    c.acceleration_limit = g('acceleration_limit', 300.0)   # overload.py:13
    c.max_velocity = g('max_velocity', 3)                   # high_speed.py:13
    c.min_velocity = g('min_velocity', 0.01)                # low_speed.py:13
    c.min_alpha, c.max_alpha = g('min_alpha', -20), g('max_alpha', 45)   # extreme_state.py:13-16
    c.min_beta, c.max_beta = g('min_beta', -30), g('max_beta', 30)
    c.max_check_interval = g('max_check_interval', 1500)    # unreach_heading.py:16
    c.min_check_interval = g('min_check_interval', 300)     # unreach_heading.py:17
    c.init_T = config.init_state['init_T']                  # F16_model.py:43
    c.max_altitude, c.min_altitude = g('max_altitude', 20000), g('min_altitude', 19000)  # F16_model.py:25-26
    c.max_vt, c.min_vt = g('max_vt', 1200), g('min_vt', 1000)                           # F16_model.py:27-28
    c.max_heading_increment = g('max_heading_increment', 0.3)          # control_task.py:31
    c.max_pitch_increment = g('max_pitch_increment', 0.3)              # control_task.py:30
    c.max_velocities_u_increment = g('max_velocities_u_increment', 100)  # control_task.py:32
    c.max_distance, c.min_distance = g('max_distance', 2000), g('min_distance', 2000)  # tracking_task.py:30-31
    # numerics option of this framework: scenario key `aero_1d_tables`, else env NPF16_AERO_1D_TABLES
    if aero_1d_tables is None:
        aero_1d_tables = g('aero_1d_tables', int(os.environ.get('NPF16_AERO_1D_TABLES', '0')))
    c.aero_1d_tables = 1 if aero_1d_tables else 0
    # the airframe as data (not a reference key: the reference spells these as literals, F16_dynamics.py:61-76): scenario key `airframe`, a
    # mapping of np_f16_airframe fields that differ from the F-16's, e.g. {mass: 700, Jy: 60000}; absent / empty = the F-16
    c.airframe = _lib.airframe(g('airframe', None))
    return c


def _check_numerics_spec(sd):
    spec = sd.get('numerics_spec')
    if spec != NUMERICS_SPEC:
        import warnings
        warnings.warn(f"checkpoint written under numerics spec {spec if spec is not None else '<unrecorded>'} (ABI {sd.get('abi_version', '?')}), "
                      f'this library implements spec {NUMERICS_SPEC} (ABI {_lib.ABI_VERSION}): the state loads, but the continued '
                      'trajectories will differ in the last bits from a run recorded under the old spec', RuntimeWarning, stacklevel=3)


def _check_airframe(sd, airframe):
    """A checkpoint written with another np_f16_airframe block would continue on a different aircraft: refuse it (checkpoints from before
    round 6 carry no block: they were the F-16, i.e. the all-zero block)."""
    def key(b):      # the all-zero block and the spelled-out F-16 defaults are the same aircraft
        b = bytes(b)
        return bytes(_lib.airframe({'Heng': 0.0})) if not any(b) else b
    if key(sd.get('airframe', bytes(len(bytes(airframe))))) != key(airframe):
        raise ValueError('checkpoint was written with a different airframe block (np_f16_airframe) than this env was built with')


class F16Batch:
    """N aircraft on one GPU.  `row0` is the global index of local row 0 (sharded batches)."""

    def __init__(self, n, config, task, device, seed=0, solver=None, row0=0, blob_path=ASSET_BLOB, aero_1d_tables=None):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError(f"neuralplane_amd runs on MI355X (torch device 'cuda:N'), not on '{device}': "
                               'there is no CPU fallback')
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.n = int(n)
        self.task = task
        self.cfg = cfg_from_config(config, task, solver, aero_1d_tables)
        self.aero_1d_tables = bool(self.cfg.aero_1d_tables)
        self.noise_scale = float(self.cfg.noise_scale)
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.row0 = int(row0)
        self.call_idx = 0
        with open(blob_path, 'rb') as f:
            blob = f.read()
        ctx = C.c_void_p()
        _lib.check(self.lib.np_f16_ctx_create(blob, len(blob), C.byref(self.cfg), self.device.index, C.byref(ctx)))
        self._ctx = ctx
        d = self.device
        n = self.n
        self.s = torch.zeros((12, n), dtype=torch.float32, device=d)
        self.u = torch.zeros((5, n), dtype=torch.float32, device=d)
        self.tgt = torch.zeros((3, n), dtype=torch.float32, device=d)
        self.step_count = torch.zeros(n, dtype=torch.int64, device=d)
        # BaseEnv.__init__ leaves all three flags set so that the first reset()/step() initialises
        # every row (env_base.py:31-33)
        self.flags = torch.ones((3, n), dtype=torch.uint8, device=d)
        # cross-step cache of the 14 force-side (alpha, beta)-only aero coefficients (np_f16_io.coef_cache).  The cache carries the
        # (alpha, beta) its coefficients belong to and the step kernels re-evaluate them in every wave that finds the state changed
        # (np_nets.h, round 4): results do not depend on who wrote `s`.  The `_version` bookkeeping below is only a fast path — an edit
        # torch can see (model.s[mask] = ...) sends the whole batch through the un-cached kernel instead of refilling wave by wave.
        self.coef_cache = torch.empty(int(self.lib.np_f16_cache_floats(n)), dtype=torch.float32, device=d)
        self._cache_valid = False
        self._no_cache = bool(os.environ.get('NPF16_NO_CACHE'))
        self._io_cached = None
        self._s_version = self.s._version
        self._derived = None
        self._derived_key = None
        self._version = 0
        # device-side RNG counter base for launches replayed from a HIP graph (np_f16_io.call_idx_base)
        self.call_base = torch.zeros(1, dtype=torch.int64, device=d)
        # per-condition termination counters (np_f16_io.term_counters), accumulated on the device, read lazily
        self.term_counters = torch.zeros(7, dtype=torch.int32, device=d)
        # per-aircraft condition bits of the last step (np_f16_io.term_reasons); allocated by track_termination_reasons()
        self.term_reasons = None
        self.reward_task = None     # the task reward term of the last step (np_f16_io.reward_task); allocated by track_reward_terms()
        # bumped whenever one of the optional output pointers above changes: a HIP graph that baked the old pointers must be
        # re-captured before its next replay (PlanningEnv._step_graph checks it)
        self._io_epoch = 0

    def __del__(self):
        ctx = getattr(self, '_ctx', None)
        if ctx:
            try:
                self.lib.np_f16_ctx_destroy(ctx)
            except Exception:
                pass
            self._ctx = None

    # -- helpers -------------------------------------------------------------------------------
    def _stream(self):
        return _lib.stream_ptr(self.device)

    # -- graph-safe raw launches: caller-owned static buffers, no Python-side state is touched -------------
    def launch_static(self, flags_in, flags_out, call_offset, action=None, obs=None, reward=None, inner=False, cache_valid=False,
                      ll_tgt=None, ll_obs=None):
        """One np_f16_reset (action is None) or np_f16_step launch on fixed buffers, RNG counter = *call_base + call_offset.
        Safe to capture in a HIP graph (torch.cuda.graph): every argument is baked, the counter base lives on the device."""
        io = _lib.NpF16Io()
        io.s, io.u, io.tgt, io.ld = self.s.data_ptr(), self.u.data_ptr(), self.tgt.data_ptr(), self.n
        io.step_count = self.step_count.data_ptr()
        io.done_in, io.bad_in, io.timeout_in = flags_in[0].data_ptr(), flags_in[1].data_ptr(), flags_in[2].data_ptr()
        io.done_out, io.bad_out, io.timeout_out = flags_out[0].data_ptr(), flags_out[1].data_ptr(), flags_out[2].data_ptr()
        if action is not None:
            io.action, io.act_stride = action.data_ptr(), action.stride(0)
        io.obs = obs.data_ptr() if obs is not None else None
        io.reward = reward.data_ptr() if reward is not None else None
        io.coef_cache = self.coef_cache.data_ptr()
        io.cache_valid = 1 if (cache_valid and not os.environ.get('NPF16_NO_CACHE')) else 0
        io.inner_step = 1 if inner else 0
        io.seed, io.call_idx, io.row0 = self.seed, int(call_offset), self.row0
        io.call_idx_base = self.call_base.data_ptr()
        io.term_counters = self.term_counters.data_ptr()
        io.term_reasons = self.term_reasons.data_ptr() if self.term_reasons is not None else None
        io.reward_task = self.reward_task.data_ptr() if self.reward_task is not None else None
        io.ll_tgt = ll_tgt.data_ptr() if ll_tgt is not None else None      # inner step: the step kernel also writes the NEXT low-level observation
        io.ll_obs = ll_obs.data_ptr() if ll_obs is not None else None
        fn = self.lib.np_f16_reset if action is None else self.lib.np_f16_step
        _lib.check(fn(self._ctx, self.n, C.byref(io), self._stream()))

    def lowlevel_obs_into(self, tgt3, obs):
        _lib.check(self.lib.np_f16_lowlevel_obs(self._ctx, self.n, self.s.data_ptr(), self.u.data_ptr(), tgt3.data_ptr(), self.n,
                                                obs.data_ptr(), self._stream()))

    def _io(self, new_flags, action, obs, reward, rand_u, noise, inner=False, ll_tgt=None, ll_obs=None):
        io = self._io_cached
        if io is None:  # the fields that never change are filled once (the struct is re-used: ~10 us less host time per step)
            io = self._io_cached = _lib.NpF16Io()
            io.s, io.u, io.tgt, io.ld = self.s.data_ptr(), self.u.data_ptr(), self.tgt.data_ptr(), self.n
            io.step_count = self.step_count.data_ptr()
            io.coef_cache = self.coef_cache.data_ptr()
            io.call_idx_base = None
            io.term_counters = self.term_counters.data_ptr()
        io.term_reasons = self.term_reasons.data_ptr() if self.term_reasons is not None else None
        io.reward_task = self.reward_task.data_ptr() if self.reward_task is not None else None
        n = self.n
        if not self.flags.is_contiguous():
            self.flags = self.flags.contiguous()
        fi, fo = self.flags.data_ptr(), new_flags.data_ptr()        # [3, n] uint8, contiguous
        io.done_in, io.bad_in, io.timeout_in = fi, fi + n, fi + 2 * n
        io.done_out, io.bad_out, io.timeout_out = fo, fo + n, fo + 2 * n
        if action is not None:
            io.action, io.act_stride = action.data_ptr(), action.stride(0)
        else:
            io.action, io.act_stride = None, 0
        io.obs = obs.data_ptr() if obs is not None else None
        io.reward = reward.data_ptr() if reward is not None else None
        io.rand_u = rand_u.data_ptr() if rand_u is not None else None
        io.noise = noise.data_ptr() if noise is not None else None
        if self.s._version != self._s_version:  # the caller edited the state: cached coefficients are stale
            self._cache_valid = False
            self._s_version = self.s._version
        io.cache_valid = 1 if (self._cache_valid and not self._no_cache) else 0
        io.inner_step = 1 if inner else 0
        io.ll_tgt = ll_tgt.data_ptr() if ll_tgt is not None else None
        io.ll_obs = ll_obs.data_ptr() if ll_obs is not None else None
        io.seed, io.call_idx, io.row0 = self.seed, self.call_idx, self.row0
        return io

    def _inject(self, t, cols):
        if t is None:
            return None
        t = torch.as_tensor(t, dtype=torch.float32, device=self.device).contiguous()
        if tuple(t.shape) != (self.n, cols):
            raise ValueError(f'expected shape ({self.n}, {cols}), got {tuple(t.shape)}')
        return t

    # -- launches ------------------------------------------------------------------------------
    def reset(self, rand_u=None, noise=None, want_obs=True):
        """BaseEnv.reset(): re-initialise flagged rows, clear all flags, return obs[n,22]."""
        obs = torch.empty((self.n, 22), dtype=torch.float32, device=self.device) if want_obs else None
        new_flags = torch.empty((3, self.n), dtype=torch.uint8, device=self.device)
        rand_u, noise = self._inject(rand_u, 5), self._inject(noise, 22)
        io = self._io(new_flags, None, obs, None, rand_u, noise)
        _lib.check(self.lib.np_f16_reset(self._ctx, self.n, C.byref(io), self._stream()))
        self.flags = new_flags
        self.call_idx += 1
        self._version += 1
        return obs

    def observe(self, noise=None):
        """BaseEnv.obs() (env_base.py:58-59 -> task.get_obs): the observation of the CURRENT state, nothing else changes (no row
        is re-initialised, flags and counters stay).  One np_f16_reset launch with an all-clear flag input; the noise is a fresh
        draw (the call counter advances, as the reference's obs() advances torch's generator)."""
        obs = torch.empty((self.n, 22), dtype=torch.float32, device=self.device)
        scratch = torch.zeros((2, 3, self.n), dtype=torch.uint8, device=self.device)   # [0]: all-clear input flags, [1]: discarded output
        keep = self.flags
        self.flags = scratch[0]
        try:
            io = self._io(scratch[1], None, obs, None, None, self._inject(noise, 22))
            # observe-only: a real reset() clears the per-aircraft condition bits "of the last step"; this launch must not
            io.term_reasons = None
            io.reward_task = None
            _lib.check(self.lib.np_f16_reset(self._ctx, self.n, C.byref(io), self._stream()))
        finally:
            self.flags = keep
        self.call_idx += 1
        return obs

    def step(self, action, rand_u=None, noise=None, inner=False, ll_tgt=None, ll_obs=None, want_obs=True, out=None):
        """BaseEnv.step(action): ONE kernel launch.  Returns obs, reward, flags[3,n] (uint8).
        out = (obs[n,22], reward[n], flags[3,n] uint8): caller-owned contiguous device buffers to write into (a rollout storage's slots).
        inner=True: one low-level iteration of PlanningEnv.step (np_f16_io.inner_step); with ll_tgt[3,n] and ll_obs[n,22] the
        launch also writes the low-level controller's NEXT observation into ll_obs (np_f16_io.ll_obs), and want_obs=False then skips
        the task observation (returned as None)."""
        if action.device != self.device or action.dtype != torch.float32:
            action = action.to(device=self.device, dtype=torch.float32)
        if action.dim() != 2 or action.shape[0] != self.n or action.shape[1] < 4:
            raise ValueError(f'action must be [n={self.n}, >=4], got {tuple(action.shape)}')
        if action.stride(1) != 1:
            action = action.contiguous()
        if (ll_obs is None) != (ll_tgt is None) or (ll_obs is not None and not inner):
            raise ValueError('ll_tgt and ll_obs come together, with inner=True')
        if not want_obs and ll_obs is None:
            raise ValueError('want_obs=False needs ll_obs (an inner step that writes the low-level observation instead)')
        if out is not None:
            obs, reward, new_flags = out
            if not (obs.is_contiguous() and reward.is_contiguous() and new_flags.is_contiguous() and obs.numel() == 22 * self.n and reward.numel() == self.n and
                    new_flags.numel() == 3 * self.n and obs.dtype == reward.dtype == torch.float32 and new_flags.dtype == torch.uint8 and
                    obs.device == reward.device == new_flags.device == self.device and new_flags.data_ptr() != self.flags.data_ptr()):
                raise ValueError('out = (obs[n,22] float32, reward[n] float32, flags[3,n] uint8): contiguous device buffers, flags distinct from the current flags')
        else:
            obs = torch.empty((self.n, 22), dtype=torch.float32, device=self.device) if want_obs else None
            reward = torch.empty(self.n, dtype=torch.float32, device=self.device)
            new_flags = torch.empty((3, self.n), dtype=torch.uint8, device=self.device)
        rand_u, noise = self._inject(rand_u, 5), self._inject(noise, 22)
        io = self._io(new_flags, action, obs, reward, rand_u, noise, inner=inner, ll_tgt=ll_tgt, ll_obs=ll_obs)
        _lib.check(self.lib.np_f16_step(self._ctx, self.n, C.byref(io), self._stream()))
        self._cache_valid = True  # every row's coefficients were just rewritten for its new state
        self.flags = new_flags
        self.call_idx += 1
        self._version += 1
        return obs, reward, new_flags

    def update(self, action):
        """F16Model.update(action) on its own (reference envs/models/F16_model.py:51-67; called directly by envs/planning_env.py:160 and
        example/quick_start.ipynb): clamp, control lag and the integrator for every row as ONE launch (np_f16_io.inner_step =
        NP_INNER_UPDATE_ONLY) — no auto-reset, step_count / flags / targets untouched, no observation, no reward.  The cross-step
        coefficient cache is refreshed for the state reached, so update() and step() can be mixed freely."""
        if action.device != self.device or action.dtype != torch.float32:
            action = action.to(device=self.device, dtype=torch.float32)
        if action.dim() != 2 or action.shape[0] != self.n or action.shape[1] < 4:
            raise ValueError(f'action must be [n={self.n}, >=4], got {tuple(action.shape)}')
        if action.stride(1) != 1:
            action = action.contiguous()
        scratch = torch.empty((3, self.n), dtype=torch.uint8, device=self.device)   # *_out = *_in: the env's flags stay where they are
        io = self._io(scratch, action, None, None, None, None)
        io.inner_step = _lib.INNER_UPDATE_ONLY
        io.term_reasons = io.reward_task = None
        try:
            _lib.check(self.lib.np_f16_step(self._ctx, self.n, C.byref(io), self._stream()))
        finally:
            io.inner_step = 0
        self._cache_valid = True
        self.call_idx += 1
        self._version += 1

    def model_reset(self, rand_u=None):
        """F16Model.reset(env) on its own (F16_model.py:33-45): state and controls of the rows the env has flagged are re-initialised
        (altitude and vt drawn) — the reset kernel on the env's flags; what BaseEnv.reset does beyond that (task.reset's targets, the step
        counters, clearing the flags: env_base.py:83-97) is undone around the launch, so flags, targets and counters are as before."""
        keep_tgt, keep_sc = self.tgt.clone(), self.step_count.clone()
        scratch = torch.empty((3, self.n), dtype=torch.uint8, device=self.device)
        io = self._io(scratch, None, None, None, self._inject(rand_u, 5), None)
        io.term_reasons = io.reward_task = None
        _lib.check(self.lib.np_f16_reset(self._ctx, self.n, C.byref(io), self._stream()))
        self.tgt.copy_(keep_tgt)
        self.step_count.copy_(keep_sc)
        self.call_idx += 1
        self._version += 1

    def planning_inner_loop(self, actor_weights, ll_obs, rnn, masks, ll_act, tgt3, flags_scratch, iterations, groups=0, mode=0, waves=0, block=0, check=0):
        """np_planning_inner_loop: the `iterations` low-level iterations of PlanningEnv.step (controller forward + inner FDM step each)
        enqueued by one library call.  ll_obs = (first input [n,22], scratch [n,22]); rnn = (state on entry [n,128], scratch [n,128]);
        flags_scratch [3,n] uint8.  Returns obs (task observation of the last iteration), reward, flags; the final recurrent state is in
        rnn[iterations & 1].  mode: np_planning_loop.mode (0 automatic, 1 launch by launch, 2 the persistent kernel — all iterations in
        one launch, 3 the same with the (tile, iteration) work queue); waves: waves per 32-row tile of the persistent kernel (0, 4, 8);
        check: np_planning_loop.check (0 sync: a stalled guest / queue schedule raises _lib.PlanningStalled with every in-place buffer restored
        and this object's bookkeeping untouched; 1 deferred)."""
        n = self.n
        obs = torch.empty((n, 22), dtype=torch.float32, device=self.device)
        reward = torch.empty(n, dtype=torch.float32, device=self.device)
        io = self._io(flags_scratch, ll_act, obs, reward, None, None, inner=True, ll_tgt=tgt3, ll_obs=ll_obs[1])
        lp = _lib.NpPlanningLoop()
        lp.iterations, lp.groups = int(iterations), int(groups)
        lp.mode, lp.waves, lp.block, lp.check = int(mode), int(waves), int(block), int(check)
        lp.actor_weights = actor_weights.data_ptr()
        lp.actor_weights_floats = int(actor_weights.numel())   # NP_ACTOR_NUM_FLOATS (fp32 numerics) or NP_ACTOR_I8_NUM_FLOATS (block fixed point)
        lp.ll_obs[0], lp.ll_obs[1] = ll_obs[0].data_ptr(), ll_obs[1].data_ptr()
        lp.rnn[0], lp.rnn[1] = rnn[0].data_ptr(), rnn[1].data_ptr()
        lp.masks, lp.ll_act, lp.ll_tgt = masks.data_ptr(), ll_act.data_ptr(), tgt3.data_ptr()
        lp.flags[0], lp.flags[1] = self.flags.data_ptr(), flags_scratch.data_ptr()   # _io made self.flags contiguous
        if self._no_cache:
            # NPF16_NO_CACHE (the cache-off A/B validation switch): the library treats a loop WITHOUT a cache buffer as "never valid" —
            # every one of the iterations re-evaluates all coefficients (launch-by-launch path), not just the first
            io.coef_cache = None
            io.cache_valid = 0
            lp.mode = 1
        try:
            _lib.check(self.lib.np_planning_inner_loop(self._ctx, n, C.byref(io), C.byref(lp), self._stream()))
        finally:
            io.coef_cache = self.coef_cache.data_ptr()   # `io` is the cached struct of _io()
        self._cache_valid = not self._no_cache
        if iterations & 1:
            self.flags = flags_scratch
        self.call_idx += int(iterations)
        self._version += 1
        return obs, reward, self.flags

    def lowlevel_obs(self, tgt3):
        """PlanningEnv.low_level_obs for targets tgt3[3,n] (pitch, heading, vt) -> obs[n,22]."""
        tgt3 = torch.as_tensor(tgt3, dtype=torch.float32, device=self.device).contiguous()
        if tuple(tgt3.shape) != (3, self.n):
            raise ValueError(f'tgt3 must be [3, {self.n}]')
        obs = torch.empty((self.n, 22), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.np_f16_lowlevel_obs(self._ctx, self.n, self.s.data_ptr(), self.u.data_ptr(), tgt3.data_ptr(), self.n,
                                                obs.data_ptr(), self._stream()))
        return obs

    def planning_targets_obs(self, action, tgt3):
        """PlanningEnv.step's prelude (planning_env.py:146-152) in one launch: tgt3[3,n] <- (pitch, heading, vt) + clamp(action, -1, 1) *
        (0.3, 0.3, 30); returns low_level_obs(tgt3) [n,22].  action: float32 [n, >=3] on this device, rows contiguous."""
        if action.dtype != torch.float32 or action.device != self.s.device or action.dim() != 2 or action.shape[0] != self.n or action.shape[1] < 3 \
                or action.stride(1) != 1 or tuple(tgt3.shape) != (3, self.n) or not tgt3.is_contiguous():
            raise ValueError('planning_targets_obs: action must be float32 [n, >=3] with contiguous rows on the env device, tgt3 contiguous [3, n]')
        obs = torch.empty((self.n, 22), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.np_planning_targets_obs(self._ctx, self.n, self.s.data_ptr(), self.u.data_ptr(), action.data_ptr(), action.stride(0),
                                                    tgt3.data_ptr(), self.n, obs.data_ptr(), self._stream()))
        return obs

    def aero_coefficients(self, alpha_deg, beta_deg, el):
        """The 43 aero coefficient surrogates at arbitrary inputs in degrees (hifi_F16_AeroData.py:745-822) -> [43, m] in the
        reference's evaluation order, through the same device code the step kernels run (np_f16_aero_coefficients).  Row 24
        (delta_Czq_lef, never read by nlplant) is 0."""
        a, b, e = (torch.as_tensor(v, dtype=torch.float32, device=self.device).reshape(-1).contiguous() for v in (alpha_deg, beta_deg, el))
        if not (a.numel() == b.numel() == e.numel()):
            raise ValueError('alpha, beta and el must have the same number of elements')
        m = a.numel()
        out = torch.empty((NUM_NETS, m), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.np_f16_aero_coefficients(self._ctx, m, a.data_ptr(), b.data_ptr(), e.data_ptr(), out.data_ptr(), m,
                                                     self._stream()))
        return out

    def invalidate(self):
        """Tell the batch that `s` / `u` were written through a path torch cannot see (`.data`, DLPack, a foreign kernel): the getters'
        cache (`derived`) is dropped.  The step kernels need no such hint — the coefficient cache validates itself (np_nets.h)."""
        self._version += 1
        self._derived = None

    def derived(self):
        """[23,n] derived quantities at the current (s,u) (np_f16_derived), cached per state version."""
        key = (self._version, self.s.data_ptr(), self.u.data_ptr(), self.s._version, self.u._version)
        if self._derived is None or self._derived_key != key:
            out = torch.empty((NUM_DERIVED, self.n), dtype=torch.float32, device=self.device)
            _lib.check(self.lib.np_f16_derived(self._ctx, self.n, self.s.data_ptr(), self.u.data_ptr(), self.n,
                                               out.data_ptr(), self.n, self._stream()))
            self._derived, self._derived_key = out, key
        return self._derived

    # -- checkpoint / resume (the reference never checkpoints env state) ------------
    def state_dict(self):
        """Everything needed to continue the trajectories bit for bit: state, targets, counters, the flags
        left by the last step and the RNG counter (the RNG is counter-based: no generator state)."""
        return {'s': self.s.clone(), 'u': self.u.clone(), 'tgt': self.tgt.clone(), 'step_count': self.step_count.clone(),
                'flags': self.flags.clone(), 'call_idx': int(self.call_idx), 'seed': int(self.seed), 'row0': int(self.row0),
                'task': self.task, 'n': self.n, 'numerics_spec': NUMERICS_SPEC, 'abi_version': _lib.ABI_VERSION,
                'airframe': bytes(self.cfg.airframe)}   # np_f16_airframe as built (all zero = the F-16): a resumed run must fly the same aircraft

    def load_state_dict(self, sd):
        if sd['n'] != self.n or sd['task'] != self.task:
            raise ValueError(f"checkpoint is for n={sd['n']}, task={sd['task']}; this batch is n={self.n}, task={self.task}")
        _check_numerics_spec(sd)
        _check_airframe(sd, self.cfg.airframe)
        for k in ('s', 'u', 'tgt', 'step_count'):
            getattr(self, k).copy_(sd[k].to(self.device))
        self.flags = sd['flags'].to(self.device).clone()
        self.call_idx, self.seed, self.row0 = int(sd['call_idx']), int(sd['seed']), int(sd['row0'])
        self._cache_valid = False  # cached coefficients belong to the state that was just overwritten
        self._version += 1

    # -- kernel timing (bench.py) ---------------------------------------------------------------
    def set_timing(self, enable):
        _lib.check(self.lib.np_f16_set_timing(self._ctx, int(bool(enable))))

    def get_timing(self):
        ms, cnt = C.c_double(), C.c_int64()
        _lib.check(self.lib.np_f16_get_timing(self._ctx, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def get_timing_samples(self):
        """Durations (ms) of the individual launches timed since set_timing(True), in launch order."""
        cnt = C.c_int64()
        _lib.check(self.lib.np_f16_get_timing_samples(self._ctx, None, 0, C.byref(cnt)))
        buf = (C.c_float * max(1, cnt.value))()
        _lib.check(self.lib.np_f16_get_timing_samples(self._ctx, buf, cnt.value, C.byref(cnt)))
        return [float(buf[i]) for i in range(cnt.value)]

    TERM_NAMES = ('overload', 'low_altitude', 'high_speed', 'low_speed', 'extreme_state', 'unreach', 'reached')

    def track_termination_reasons(self, enable=True):
        """Have every following step also write, per aircraft, WHICH conditions fired (uint8[n], bit k = TERM_NAMES[k]) — the
        per-row version of the counters below; one more byte stored per aircraft and step.  Returns the tensor (zeros until the
        next step)."""
        if enable and self.term_reasons is None:
            self.term_reasons = torch.zeros(self.n, dtype=torch.uint8, device=self.device)
            self._io_epoch += 1
        elif not enable and self.term_reasons is not None:
            self.term_reasons = None
            self._io_epoch += 1
        return self.term_reasons

    def track_reward_terms(self, enable=True):
        """Have every following step also write the task's own reward term per aircraft (float32[n]; `reward` = it + the event
        term -200 * bad_done + 200 * done).  Returns the tensor (zeros until the next step)."""
        if enable and self.reward_task is None:
            self.reward_task = torch.zeros(self.n, dtype=torch.float32, device=self.device)
            self._io_epoch += 1
        elif not enable and self.reward_task is not None:
            self.reward_task = None
            self._io_epoch += 1
        return self.reward_task

    def termination_counts(self, reset=False):
        """{condition: aircraft that tripped it since the last reset of the counters} — one small D2H copy, on demand (the
        reference prints these sums from inside every condition on every step, a host sync each)."""
        c = self.term_counters.cpu().tolist()
        if reset:
            self.term_counters.zero_()
        return dict(zip(self.TERM_NAMES, c))

    def set_kernel_variant(self, variant):
        """'auto' (default: 'latency8' while n <= 16384, 'latency' while n <= 49152, 'latency2' while n <= 98304, then 'pair'),
        'latency8', 'latency', 'latency2', 'throughput', 'pair' — bit-identical results."""
        _lib.check(self.lib.np_f16_set_kernel_variant(self._ctx, _lib.KERNEL_VARIANTS[variant]))


# =====================================================================================================
# SingleCombat 1v1
# =====================================================================================================
PID_CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'envs', 'configs', 'pid')
NUM_PID = 11
NUM_OBS_COMBAT = 15


def combat_cfg_from_config(config, solver=None, aero_1d_tables=None, pid_dir=PID_CONFIG_DIR):
    """selfplay attribute bag + pid/*.yaml -> np_f16_combat_cfg (defaults: singlecombat_env.py:29-44, the condition classes)."""
    import yaml
    g = lambda k, d: getattr(config, k, d)  # noqa: E731
    c = _lib.NpF16CombatCfg()
    sol = solver or g('solver', 'euler')
    if sol not in _lib.SOLVERS:
        raise NotImplementedError(f"solver '{sol}' (supported: euler, rk4)")
    c.solver = _lib.SOLVERS[sol]
    c.inner_steps = 5                                          # singlecombat_env.py:243
    c.dt, c.airspeed = g('dt', 0.02), g('airspeed', 0)
    c.altitude_limit, c.acceleration_limit = g('altitude_limit', 2500.0), g('acceleration_limit', 300.0)
    c.max_velocity, c.min_velocity = g('max_velocity', 3), g('min_velocity', 0.01)
    c.min_alpha, c.max_alpha = g('min_alpha', -20), g('max_alpha', 45)
    c.min_beta, c.max_beta = g('min_beta', -30), g('max_beta', 30)
    c.distance_limit = g('distance_limit', 200)                # crash.py:17
    c.max_steps = g('max_steps', 500)                          # timeout.py:15
    c.init_T, c.target_dist = g('init_T', 2000), g('target_dist', 3)
    c.max_altitude, c.min_altitude = g('max_altitude', 20000), g('min_altitude', 19000)
    c.max_vt, c.min_vt = g('max_vt', 1200), g('min_vt', 1000)
    c.max_heading, c.min_heading = g('max_heading', 0.5), g('min_heading', -0.5)
    c.max_npos, c.min_npos = g('max_npos', 5000), g('min_npos', -5000)
    c.max_epos, c.min_epos = g('max_epos', 5000), g('min_epos', -5000)
    for name in ('roll', 'pitch', 'yaw'):
        path = os.path.join(pid_dir, f'{name}controller.yaml')
        assert os.path.exists(path), f'config path {path} does not exist.'
        with open(path, 'r', encoding='utf-8') as f:
            p = yaml.safe_load(f)
        gains = getattr(c, name)
        for k in ('Kp', 'Ki', 'Kd', 'Kff', 'Kimax', 'tau'):
            setattr(gains, k, p[k])
        gains.rmax_pos, gains.rmax_neg = p.get('rmax_pos', 0), p.get('rmax_neg', 0)
        if name == 'pitch':
            c.roll_ff, c.gravity = p['roll_ff'], p['gravity']
    c.airspeed_min, c.airspeed_max = 100, 2300                 # controller.py:15
    c.airframe = _lib.airframe(g('airframe', None))             # see cfg_from_config
    if aero_1d_tables is None:
        aero_1d_tables = g('aero_1d_tables', int(os.environ.get('NPF16_AERO_1D_TABLES', '0')))
    c.aero_1d_tables = 1 if aero_1d_tables else 0
    return c


class F16CombatBatch:
    """num_envs 1v1 engagements (2*num_envs aircraft, rows 2k / 2k+1 = ego / enemy of env k) on one GPU.
    `env0` is the global index of local env 0 (sharded batches: partition by env, never by aircraft)."""

    def __init__(self, num_envs, config, device, seed=0, solver=None, env0=0, blob_path=ASSET_BLOB, aero_1d_tables=None):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError(f"neuralplane_amd runs on MI355X (torch device 'cuda:N'), not on '{device}': "
                               'there is no CPU fallback')
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.num_envs = int(num_envs)
        self.n = 2 * self.num_envs
        self.cfg = combat_cfg_from_config(config, solver, aero_1d_tables)
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.env0 = int(env0)
        self.call_idx = 0
        with open(blob_path, 'rb') as f:
            blob = f.read()
        ctx = C.c_void_p()
        _lib.check(self.lib.np_f16_combat_ctx_create(blob, len(blob), C.byref(self.cfg), self.device.index, C.byref(ctx)))
        self._ctx = ctx
        d, n = self.device, self.n
        self.s = torch.zeros((12, n), dtype=torch.float32, device=d)
        self.u = torch.zeros((5, n), dtype=torch.float32, device=d)
        self.pid = torch.zeros((NUM_PID, n), dtype=torch.float32, device=d)   # Controller state (controller.py:27-40: zeros)
        self.blood = torch.full((n,), 100.0, dtype=torch.float32, device=d)   # singlecombat_env.py:45
        self.step_count = torch.zeros(n, dtype=torch.int64, device=d)
        self.flags = torch.ones((3, n), dtype=torch.uint8, device=d)
        self.pid_first = True                                                 # PID.reset (pid.py:14)
        self.term_counters = torch.zeros(9, dtype=torch.int32, device=d)     # np_f16_combat_io.term_counters
        self._version = 0
        self._derived_ctx = None

    def __del__(self):
        ctx = getattr(self, '_ctx', None)
        if ctx:
            try:
                self.lib.np_f16_ctx_destroy(ctx)
            except Exception:
                pass
            self._ctx = None

    def _stream(self):
        return _lib.stream_ptr(self.device)

    def _io(self, new_flags, action, obs, reward, rand_u, action_opp=None, obs_opp=None):
        io = _lib.NpF16CombatIo()
        io.action_opp = action_opp.data_ptr() if action_opp is not None else None
        io.obs_opp = obs_opp.data_ptr() if obs_opp is not None else None
        io.s, io.u, io.pid, io.blood, io.ld = self.s.data_ptr(), self.u.data_ptr(), self.pid.data_ptr(), self.blood.data_ptr(), self.n
        io.step_count = self.step_count.data_ptr()
        f, g = self.flags, new_flags
        io.done_in, io.bad_in, io.timeout_in = f[0].data_ptr(), f[1].data_ptr(), f[2].data_ptr()
        io.done_out, io.bad_out, io.timeout_out = g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr()
        if action is not None:
            io.action, io.act_stride = action.data_ptr(), action.stride(0)
        io.obs = obs.data_ptr() if obs is not None else None
        io.reward = reward.data_ptr() if reward is not None else None
        io.rand_u = rand_u.data_ptr() if rand_u is not None else None
        io.pid_first = 1 if self.pid_first else 0
        io.term_counters = self.term_counters.data_ptr()
        io.seed, io.call_idx, io.row0 = self.seed, self.call_idx, 2 * self.env0
        return io

    def _inject(self, t):
        if t is None:
            return None
        t = torch.as_tensor(t, dtype=torch.float32, device=self.device).contiguous()
        if tuple(t.shape) != (self.n, 5):
            raise ValueError(f'expected shape ({self.n}, 5), got {tuple(t.shape)}')
        return t

    def reset(self, rand_u=None):
        """reset_done_envs + obs: both aircraft of every flagged env are re-initialised; returns obs[n,15]."""
        obs = torch.empty((self.n, NUM_OBS_COMBAT), dtype=torch.float32, device=self.device)
        new_flags = torch.empty((3, self.n), dtype=torch.uint8, device=self.device)
        io = self._io(new_flags, None, obs, None, self._inject(rand_u))
        _lib.check(self.lib.np_f16_combat_reset(self._ctx, self.num_envs, C.byref(io), self._stream()))
        self.flags = new_flags
        self.call_idx += 1
        self._version += 1
        return obs

    def observe(self):
        """SingleCombatEnv.obs() (singlecombat_env.py:64-138): the pairwise observation of the CURRENT state; no env is
        re-initialised, flags, blood, counters and controller state stay (one np_f16_combat_reset launch with all-clear flags)."""
        obs = torch.empty((self.n, NUM_OBS_COMBAT), dtype=torch.float32, device=self.device)
        scratch = torch.zeros((2, 3, self.n), dtype=torch.uint8, device=self.device)
        keep = self.flags
        self.flags = scratch[0]
        try:
            io = self._io(scratch[1], None, obs, None, None)
            _lib.check(self.lib.np_f16_combat_reset(self._ctx, self.num_envs, C.byref(io), self._stream()))
        finally:
            self.flags = keep
        return obs

    def step(self, action, rand_u=None):
        """SingleCombatEnv.step(action[n,>=4]): ONE kernel launch (5 FDM steps).  Returns obs, reward, flags[3,n]."""
        if action.device != self.device or action.dtype != torch.float32:
            action = action.to(device=self.device, dtype=torch.float32)
        if action.dim() != 2 or action.shape[0] != self.n or action.shape[1] < 4:
            raise ValueError(f'action must be [n={self.n}, >=4], got {tuple(action.shape)}')
        if action.stride(1) != 1:
            action = action.contiguous()
        obs = torch.empty((self.n, NUM_OBS_COMBAT), dtype=torch.float32, device=self.device)
        reward = torch.empty(self.n, dtype=torch.float32, device=self.device)
        new_flags = torch.empty((3, self.n), dtype=torch.uint8, device=self.device)
        io = self._io(new_flags, action, obs, reward, self._inject(rand_u))
        _lib.check(self.lib.np_f16_combat_step(self._ctx, self.num_envs, C.byref(io), self._stream()))
        self.flags = new_flags
        self.pid_first = False
        self.call_idx += 1
        self._version += 1
        return obs, reward, new_flags

    # -- split (self-play) layout: ego / opponent halves as separate contiguous per-env arrays (np_f16_combat_io.action_opp /
    # obs_opp) — what the self-play runner slices out of the interleaved arrays (runner/selfplay_F16sim_runner.py:62-67, 96-100),
    # produced and consumed by the kernel directly so that the opponent exchange needs no split / stack / contiguous copies
    def _check_half(self, a, name):
        if a.device != self.device or a.dtype != torch.float32:
            a = a.to(device=self.device, dtype=torch.float32)
        if a.dim() != 2 or a.shape[0] != self.num_envs or a.shape[1] < 4:
            raise ValueError(f'{name} must be [num_envs={self.num_envs}, >=4], got {tuple(a.shape)}')
        return a if a.stride(1) == 1 else a.contiguous()

    def reset_split(self, rand_u=None, out=None):
        """reset() with the observation as (ego[E,15], opponent[E,15]); `out` = two caller-owned buffers to write into."""
        oe, oo = out if out is not None else (torch.empty((self.num_envs, NUM_OBS_COMBAT), dtype=torch.float32, device=self.device) for _ in range(2))
        new_flags = torch.empty((3, self.n), dtype=torch.uint8, device=self.device)
        io = self._io(new_flags, None, oe, None, self._inject(rand_u), obs_opp=oo)
        _lib.check(self.lib.np_f16_combat_reset(self._ctx, self.num_envs, C.byref(io), self._stream()))
        self.flags = new_flags
        self.call_idx += 1
        self._version += 1
        return oe, oo

    def step_split(self, ego_action, opp_action, rand_u=None, out=None):
        """step() on the two action halves ego_action[E,>=4], opp_action[E,>=4] (equal row strides) -> obs_ego[E,15],
        obs_opp[E,15], reward[n], flags[3,n]: ONE kernel launch, no copies on either side."""
        ego_action, opp_action = self._check_half(ego_action, 'ego_action'), self._check_half(opp_action, 'opp_action')
        if ego_action.stride(0) != opp_action.stride(0):
            # the kernel reads both halves with ONE row stride: contiguous() is a no-op on halves that are already contiguous but of
            # different widths ([E,4] next to [E,6]), so both are cut to the four columns the kernel reads
            opp_action = opp_action[:, :4].contiguous()
            ego_action = ego_action[:, :4].contiguous()
        if ego_action.stride(0) != opp_action.stride(0):
            raise ValueError(f'ego_action / opp_action row strides differ ({ego_action.stride(0)} vs {opp_action.stride(0)})')
        oe, oo = out if out is not None else (torch.empty((self.num_envs, NUM_OBS_COMBAT), dtype=torch.float32, device=self.device) for _ in range(2))
        reward = torch.empty(self.n, dtype=torch.float32, device=self.device)
        new_flags = torch.empty((3, self.n), dtype=torch.uint8, device=self.device)
        io = self._io(new_flags, ego_action, oe, reward, self._inject(rand_u), action_opp=opp_action, obs_opp=oo)
        _lib.check(self.lib.np_f16_combat_step(self._ctx, self.num_envs, C.byref(io), self._stream()))
        self.flags = new_flags
        self.pid_first = False
        self.call_idx += 1
        self._version += 1
        return oe, oo, reward, new_flags

    def state_dict(self):
        return {'s': self.s.clone(), 'u': self.u.clone(), 'pid': self.pid.clone(), 'blood': self.blood.clone(),
                'step_count': self.step_count.clone(), 'flags': self.flags.clone(), 'pid_first': bool(self.pid_first),
                'call_idx': int(self.call_idx), 'seed': int(self.seed), 'env0': int(self.env0), 'num_envs': self.num_envs,
                'numerics_spec': NUMERICS_SPEC, 'abi_version': _lib.ABI_VERSION, 'airframe': bytes(self.cfg.airframe)}

    def load_state_dict(self, sd):
        if sd['num_envs'] != self.num_envs:
            raise ValueError(f"checkpoint is for num_envs={sd['num_envs']}; this batch has {self.num_envs}")
        _check_numerics_spec(sd)
        _check_airframe(sd, self.cfg.airframe)
        for k in ('s', 'u', 'pid', 'blood', 'step_count'):
            getattr(self, k).copy_(sd[k].to(self.device))
        self.flags = sd['flags'].to(self.device).clone()
        self.pid_first = bool(sd['pid_first'])
        self.call_idx, self.seed, self.env0 = int(sd['call_idx']), int(sd['seed']), int(sd['env0'])
        self._version += 1

    def set_timing(self, enable):
        _lib.check(self.lib.np_f16_set_timing(self._ctx, int(bool(enable))))

    def get_timing(self):
        ms, cnt = C.c_double(), C.c_int64()
        _lib.check(self.lib.np_f16_get_timing(self._ctx, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def get_timing_samples(self):
        """Durations (ms) of the individual launches timed since set_timing(True), in launch order."""
        cnt = C.c_int64()
        _lib.check(self.lib.np_f16_get_timing_samples(self._ctx, None, 0, C.byref(cnt)))
        buf = (C.c_float * max(1, cnt.value))()
        _lib.check(self.lib.np_f16_get_timing_samples(self._ctx, buf, cnt.value, C.byref(cnt)))
        return [float(buf[i]) for i in range(cnt.value)]

    def set_kernel_variant(self, variant):
        """'auto' (default, Euler and MLP numerics, 256 CUs: 'latency' while n <= 16384 aircraft, 'dual8' while n <= 32768, 'dual4' while
        n <= 65536, then 'pair'; rk4 / table numerics: 'latency' while n <= 40000), 'latency', 'dual8', 'dual4', 'throughput', 'pair' —
        bit-identical results."""
        _lib.check(self.lib.np_f16_set_kernel_variant(self._ctx, _lib.KERNEL_VARIANTS[variant]))

    TERM_NAMES = ('overload', 'low_altitude', 'high_speed', 'low_speed', 'extreme_state', 'crash', 'timeout', 'shutdown_bad',
                  'shutdown_done')

    def termination_counts(self, reset=False):
        """{condition: evaluations that fired since the counters were reset} (one evaluation per aircraft and inner FDM step)."""
        c = self.term_counters.cpu().tolist()
        if reset:
            self.term_counters.zero_()
        return dict(zip(self.TERM_NAMES, c))
