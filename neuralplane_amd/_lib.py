"""ctypes binding of the C ABI in include/neuralplane_amd.h (libneuralplane_hip.so).

PyTorch is imported first on purpose: the extension's DT_NEEDED `libamdhip64.so.7` then resolves to
the HIP runtime PyTorch-ROCm already loaded (same SONAME), so tensors and kernels share one runtime.
There is no CPU fallback: if the extension is missing or fails to load, importing this module raises.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL below)

from . import build as _build

TASKS = {'heading': 0, 'control': 1, 'tracking': 2}
SOLVERS = {'euler': 0, 'rk4': 1}
ABI_VERSION = 16
INNER_UPDATE_ONLY = 2   # np_f16_io.inner_step: F16Model.update(action) on its own (NP_INNER_UPDATE_ONLY)


class NpF16Airframe(C.Structure):
    """include/neuralplane_amd.h: np_f16_airframe.  All zero = the F-16 literals of the reference (F16_dynamics.py:22-35,61-76,114-116,
    F16_model.py:52-62); `airframe(**overrides)` below builds a block from the defaults."""
    _fields_ = [(k, C.c_double) for k in ('g', 'mass', 'B', 'S', 'cbar', 'xcgr', 'xcg', 'Heng', 'Jy', 'Jxz', 'Jz', 'Jx', 'ail_ref', 'rud_ref',
                                          'atm_lapse', 'atm_exp', 'rho0', 'lag_keep', 'lag_new', 'thrust_frac', 'thrust_max', 'thrust_unit')] + [('surf_max', C.c_double * 3)]


AIRFRAME_KEYS = tuple(k for k, _ in NpF16Airframe._fields_)


def airframe(overrides=None):
    """np_f16_airframe with the reference's F-16 values (np_f16_airframe_default) and `overrides` ({field: value}; surf_max: three values) on top.
    None / {} -> the all-zero block, which the library reads as "the F-16"."""
    a = NpF16Airframe()
    if not overrides:
        return a
    lib = load()
    lib.np_f16_airframe_default.argtypes = [C.POINTER(NpF16Airframe)]
    lib.np_f16_airframe_default.restype = None
    lib.np_f16_airframe_default(C.byref(a))
    for k, v in dict(overrides).items():
        if k not in AIRFRAME_KEYS:
            raise ValueError(f'airframe: unknown field {k!r} (fields: {", ".join(AIRFRAME_KEYS)})')
        if k == 'surf_max':
            for j, x in enumerate(v):
                a.surf_max[j] = float(x)
        else:
            setattr(a, k, float(v))
    return a


class NpF16Cfg(C.Structure):
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
                ('max_distance', C.c_double), ('min_distance', C.c_double),
                ('aero_1d_tables', C.c_int32), ('reserved_cfg_', C.c_int32), ('airframe', NpF16Airframe)]


class NpF16Io(C.Structure):
    _fields_ = [('s', C.c_void_p), ('u', C.c_void_p), ('tgt', C.c_void_p), ('ld', C.c_int64),
                ('step_count', C.c_void_p),
                ('done_in', C.c_void_p), ('bad_in', C.c_void_p), ('timeout_in', C.c_void_p),
                ('done_out', C.c_void_p), ('bad_out', C.c_void_p), ('timeout_out', C.c_void_p),
                ('action', C.c_void_p), ('act_stride', C.c_int64),
                ('obs', C.c_void_p), ('reward', C.c_void_p),
                ('rand_u', C.c_void_p), ('noise', C.c_void_p),
                ('coef_cache', C.c_void_p), ('cache_valid', C.c_int32), ('inner_step', C.c_int32),
                ('seed', C.c_uint64), ('call_idx', C.c_uint64), ('row0', C.c_int64), ('call_idx_base', C.c_void_p), ('term_counters', C.c_void_p),
                ('term_reasons', C.c_void_p), ('reward_task', C.c_void_p), ('ll_tgt', C.c_void_p), ('ll_obs', C.c_void_p)]


class NpPidGains(C.Structure):
    _fields_ = [(k, C.c_double) for k in ('Kp', 'Ki', 'Kd', 'Kff', 'Kimax', 'tau', 'rmax_pos', 'rmax_neg')]


class NpF16CombatCfg(C.Structure):
    _fields_ = [('solver', C.c_int32), ('inner_steps', C.c_int32), ('dt', C.c_double), ('airspeed', C.c_double),
                ('altitude_limit', C.c_double), ('acceleration_limit', C.c_double), ('max_velocity', C.c_double),
                ('min_velocity', C.c_double), ('min_alpha', C.c_double), ('max_alpha', C.c_double),
                ('min_beta', C.c_double), ('max_beta', C.c_double), ('distance_limit', C.c_double),
                ('max_steps', C.c_int64), ('init_T', C.c_double), ('target_dist', C.c_double),
                ('max_altitude', C.c_double), ('min_altitude', C.c_double), ('max_vt', C.c_double), ('min_vt', C.c_double),
                ('max_heading', C.c_double), ('min_heading', C.c_double), ('max_npos', C.c_double), ('min_npos', C.c_double),
                ('max_epos', C.c_double), ('min_epos', C.c_double),
                ('roll', NpPidGains), ('pitch', NpPidGains), ('yaw', NpPidGains),
                ('roll_ff', C.c_double), ('gravity', C.c_double), ('airspeed_min', C.c_double), ('airspeed_max', C.c_double),
                ('aero_1d_tables', C.c_int32), ('reserved_cfg_', C.c_int32), ('airframe', NpF16Airframe)]


class NpPlanningLoop(C.Structure):
    _fields_ = [('iterations', C.c_int32), ('groups', C.c_int32), ('actor_weights', C.c_void_p),
                ('ll_obs', C.c_void_p * 2), ('rnn', C.c_void_p * 2), ('masks', C.c_void_p), ('ll_act', C.c_void_p),
                ('flags', C.c_void_p * 2), ('ll_tgt', C.c_void_p), ('mode', C.c_int32), ('waves', C.c_int32), ('block', C.c_int32), ('check', C.c_int32), ('actor_weights_floats', C.c_int64)]


class NpF16CombatIo(C.Structure):
    _fields_ = [('s', C.c_void_p), ('u', C.c_void_p), ('pid', C.c_void_p), ('blood', C.c_void_p), ('ld', C.c_int64),
                ('step_count', C.c_void_p),
                ('done_in', C.c_void_p), ('bad_in', C.c_void_p), ('timeout_in', C.c_void_p),
                ('done_out', C.c_void_p), ('bad_out', C.c_void_p), ('timeout_out', C.c_void_p),
                ('action', C.c_void_p), ('act_stride', C.c_int64), ('obs', C.c_void_p), ('reward', C.c_void_p),
                ('rand_u', C.c_void_p), ('pid_first', C.c_int32), ('reserved_io_', C.c_int32),
                ('seed', C.c_uint64), ('call_idx', C.c_uint64), ('row0', C.c_int64), ('term_counters', C.c_void_p),
                ('action_opp', C.c_void_p), ('obs_opp', C.c_void_p)]


class NpDispatchInfo(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ('pair', 'pair3', 'latency', 'latency8', 'latency2', 'latency4w', 'block', 'planning_groups', 'actor_tile32',
                                          'combat_latency')] + [('grid', C.c_int64), ('planning_mode', C.c_int32), ('planning_mode_i8', C.c_int32)]


def dispatch_plan(n, num_cus, step=True, solver=0, tables=False, variant=0):
    """np_dispatch_plan as a dict: the variant / tiling / row-group selection for n rows on a device with num_cus CUs (no GPU needed)."""
    info = NpDispatchInfo()
    lib = load()
    lib.np_dispatch_plan.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(NpDispatchInfo)]
    check(lib.np_dispatch_plan(int(n), int(num_cus), int(bool(step)), int(solver), int(bool(tables)), int(variant), C.byref(info)))
    return {k: getattr(info, k) for k, _ in NpDispatchInfo._fields_ if k != 'reserved_'}


EXPORTS = ('np_abi_version', 'np_f16_airframe_default', 'np_f16_cache_floats', 'np_last_error', 'np_f16_ctx_create', 'np_f16_ctx_destroy', 'np_f16_reset',
           'np_f16_step', 'np_f16_derived', 'np_f16_aero_coefficients', 'np_f16_lowlevel_obs', 'np_f16_set_timing', 'np_f16_get_timing', 'np_f16_get_timing_samples', 'np_f16_set_trace', 'np_selfcheck_divc',
           'np_f16_combat_ctx_create', 'np_f16_combat_reset', 'np_f16_combat_step', 'np_f16_set_kernel_variant', 'np_actor_forward', 'np_rollout_returns', 'np_planning_inner_loop', 'np_planning_check', 'np_actor_pack_i8', 'np_rollout_insert', 'np_policy_act', 'np_planning_targets_obs', 'np_dispatch_plan')
KERNEL_VARIANTS = {'auto': 0, 'latency': 1, 'throughput': 2, 'pair': 3, 'latency8': 4, 'latency2': 5, 'latency4w': 6, 'dual8': 7, 'dual4': 8}

_lib = None


def so_path():
    # NPF16_LIB: A/B timing of experimental builds inside one gpurun session (tools/microbench/ab_libs.py); never set in production.
    # An experimental build may carry other numerics under the same ABI number, so the override is announced on every load.
    override = os.environ.get('NPF16_LIB')
    if override:
        import warnings
        warnings.warn(f'NPF16_LIB is set: loading the HIP extension from {override} instead of the built library {_build.SO} '
                      '(measurement builds only: results may differ from the shipped numerics)', RuntimeWarning, stacklevel=3)
        return override
    return _build.SO


def load():
    """dlopen the extension (built in-tree by neuralplane_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    path = so_path()
    if not os.path.exists(path):
        raise RuntimeError(f'{path} is missing: run `python -m neuralplane_amd.build` (hipcc, gfx950). '
                           'neuralplane_amd has no CPU fallback.')
    lib = C.CDLL(path)
    lib.np_abi_version.restype = C.c_int
    lib.np_f16_cache_floats.restype = C.c_int64
    lib.np_f16_cache_floats.argtypes = [C.c_int64]
    lib.np_last_error.restype = C.c_char_p
    lib.np_f16_ctx_create.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(NpF16Cfg), C.c_int, C.POINTER(C.c_void_p)]
    lib.np_f16_ctx_destroy.argtypes = [C.c_void_p]
    lib.np_f16_ctx_destroy.restype = None
    lib.np_f16_reset.argtypes = [C.c_void_p, C.c_int64, C.POINTER(NpF16Io), C.c_void_p]
    lib.np_f16_step.argtypes = [C.c_void_p, C.c_int64, C.POINTER(NpF16Io), C.c_void_p]
    lib.np_f16_derived.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                   C.c_void_p]
    lib.np_f16_aero_coefficients.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.np_f16_lowlevel_obs.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    lib.np_planning_targets_obs.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    lib.np_f16_set_timing.argtypes = [C.c_void_p, C.c_int]
    lib.np_f16_set_kernel_variant.argtypes = [C.c_void_p, C.c_int]
    lib.np_actor_forward.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_void_p]
    lib.np_planning_inner_loop.argtypes = [C.c_void_p, C.c_int64, C.POINTER(NpF16Io), C.POINTER(NpPlanningLoop), C.c_void_p]
    lib.np_planning_check.argtypes = [C.c_void_p]
    lib.np_actor_pack_i8.argtypes = [C.c_void_p, C.c_void_p]
    lib.np_rollout_insert.argtypes = [C.POINTER(NpRolloutStep), C.c_int, C.c_void_p]
    lib.np_policy_act.argtypes = [C.c_void_p, C.c_int, C.c_void_p]   # np_policy_step * (policy.py: NpPolicyStep)
    lib.np_rollout_returns.argtypes = [C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.np_f16_combat_ctx_create.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(NpF16CombatCfg), C.c_int, C.POINTER(C.c_void_p)]
    lib.np_f16_combat_reset.argtypes = [C.c_void_p, C.c_int64, C.POINTER(NpF16CombatIo), C.c_void_p]
    lib.np_f16_combat_step.argtypes = [C.c_void_p, C.c_int64, C.POINTER(NpF16CombatIo), C.c_void_p]
    lib.np_f16_get_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.np_f16_set_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.np_selfcheck_divc.argtypes = [C.c_float, C.POINTER(C.c_uint64), C.c_int]
    lib.np_f16_get_timing_samples.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    if lib.np_abi_version() != ABI_VERSION:
        raise RuntimeError('libneuralplane_hip.so ABI version mismatch')
    _lib = lib
    return lib


class NpRolloutStep(C.Structure):   # include/neuralplane_amd.h: np_rollout_step
    _fields_ = [('num_envs', C.c_int64), ('num_agents', C.c_int64), ('step', C.c_int64), ('obs_dim', C.c_int32), ('act_dim', C.c_int32), ('rnn_dim', C.c_int32),
                ('reserved_', C.c_int32)] + [(k, C.c_void_p) for k in ('obs', 'actions', 'rewards', 'masks', 'bad_masks', 'action_log_probs', 'value_preds',
                                                                      'rnn_states_actor', 'rnn_states_critic', 'obs_in', 'actions_in', 'rewards_in',
                                                                      'action_log_probs_in', 'values_in', 'rnn_states_actor_in', 'rnn_states_critic_in', 'done_in',
                                                                      'bad_done_in', 'exceed_time_limit_in')]


E_PLANNING_STALLED, E_PLANNING_STALLED_LOST = 2, 3   # include/neuralplane_amd.h


class PlanningStalled(RuntimeError):
    """np_planning_inner_loop: a progress-word wait of the guest / queue schedule expired (NP_E_PLANNING_STALLED: the in-place buffers were
    restored, `restored` is True — re-run the macro-step launch by launch; NP_E_PLANNING_STALLED_LOST: check = deferred, nothing was kept)."""

    def __init__(self, msg, restored):
        super().__init__(msg)
        self.restored = restored


_raw_stream = None


def stream_ptr(device):
    """The current torch stream of `device` as the void* the C ABI takes.  torch.cuda.current_stream(device).cuda_stream builds a Stream object per
    call (4.5 us of the 16.5 us of Python behind an env.step, tools/microbench/host_step_profile.py); torch's own raw accessor returns the handle."""
    global _raw_stream
    if _raw_stream is None:
        import torch
        _raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', False)
    if _raw_stream:
        return C.c_void_p(_raw_stream(device.index))
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def check(rc):
    if rc != 0:
        msg = 'neuralplane_amd: ' + load().np_last_error().decode()
        if rc in (E_PLANNING_STALLED, E_PLANNING_STALLED_LOST):
            raise PlanningStalled(msg, rc == E_PLANNING_STALLED)
        raise RuntimeError(msg)
