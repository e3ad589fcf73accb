/*
 * neuralplane_amd.h — C ABI of the MI355X-native NeuralPlane F-16 env.step hot path.
 *
 * The reference (xuecy22/NeuralPlane @ 2024-12-18) is 100 % Python/PyTorch and has NO native or
 * FFI layer; its boundary for this path is the duck-typed Python env surface.  This header is the
 * native boundary a replacement must export underneath that surface: each entry point names the
 * reference Python interface it replaces (paths relative to the reference root).  The Python
 * mirror of the surface (neuralplane_amd/envs/) binds these symbols with ctypes; INTEGRATION.md
 * shows the stub.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / HIP types in signatures (the stream is a
 *     `void*` holding a hipStream_t, NULL = default stream).
 *   - All buffers are DEVICE pointers owned by the caller; the library never allocates per call
 *     and never frees caller memory.  Kernels are enqueued asynchronously on `stream`.
 *   - Every function returns 0 on success, non-zero on failure; np_last_error() gives the
 *     thread-local message.  No exceptions cross the ABI.  There is NO CPU fallback: without a
 *     gfx950 device every compute entry point fails.
 *   - State is structure-of-arrays: component k of aircraft i is at base[k*ld + i].
 */
#ifndef NEURALPLANE_AMD_H
#define NEURALPLANE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NP_ABI_VERSION 16

#define NP_NUM_STATES 12   /* npos epos alt | roll pitch yaw | vt alpha beta | P Q R   (F16_dynamics.py:39-51) */
#define NP_NUM_CONTROLS 5  /* T el ail rud lef                                         (F16_dynamics.py:53-58) */
#define NP_NUM_TARGETS 3   /* task targets (heading: alt,heading,vt | control: pitch,heading,vt | tracking: n,e,alt) */
#define NP_NUM_OBS 22      /* envs/tasks/heading_task.py:71-152                                                   */
#define NP_NUM_DERIVED 23  /* rows written by np_f16_derived()                                                    */
#define NP_NUM_NETS 43     /* rows written by np_f16_aero_coefficients(): the aero surrogates, hifi_F16_AeroData.py    */
#define NP_NUM_TERM_COUNTERS 7 /* per-condition termination counters (np_f16_io.term_counters)                      */
#define NP_NUM_CACHED 14   /* coefficients per aircraft in the cross-step cache (np_f16_io.coef_cache); the cache also carries two key
                            * values per aircraft since ABI 14: size it with np_f16_cache_floats()                 */

enum { NP_TASK_HEADING = 0, NP_TASK_CONTROL = 1, NP_TASK_TRACKING = 2 }; /* envs/control_env.py:28-35 */
enum { NP_SOLVER_EULER = 0, NP_SOLVER_RK4 = 1 };                          /* envs/models/F16_model.py:16,64-67 */
#define NP_INNER_UPDATE_ONLY 2 /* np_f16_io.inner_step: F16Model.update(action) on its own (envs/models/F16_model.py:51-67) */

/* The airframe as data (ABI 16).  What the reference spells as literals inside F16Dynamics.nlplant / atmos and F16Model.update
 * (envs/models/F16/F16_dynamics.py:22-35,61-76,114-116; envs/models/F16_model.py:52-62).  An ALL-ZERO block means "the F-16": every
 * field then takes the reference's literal (np_f16_airframe_default() writes them out), and a context computes exactly what it computed
 * when these were compile-time constants.  Together with a second weights blob of the same net topology (np_f16_ctx_create) this is how
 * another aircraft is expressed — mass and inertias, reference areas and lengths, c.g. positions, engine angular momentum, control
 * scalings, atmosphere constants, command scales.  Derived constants (xcgr - xcg, cbar / B, the inertia products, Jx Jz - Jxz^2,
 * reciprocals of divisors) are folded in double on the host and rounded once, as the literal expressions were.
 * PARITY of anything but the defaults is UNPINNED: the reference holds no second airframe; HIP == CPU oracle holds for any block. */
typedef struct np_f16_airframe {
    double g, mass, B, S, cbar, xcgr, xcg, Heng;  /* F16_dynamics.py:61-68: 32.17, 636.94, 30, 300, 11.32, 0.35, 0.30, 0 */
    double Jy, Jxz, Jz, Jx;                        /* :71-74: 55814, 982, 63100, 9496 */
    double ail_ref, rud_ref;                       /* :114-115: dail = ail / 21.5, drud = rud / 30 (dlef = 1 - lef / 25 with lef == 0) */
    double atm_lapse, atm_exp, rho0;               /* atmos :22-35: tfac = 1 - 0.703e-5 alt, rho = 2.377e-3 tfac^4.14 */
    double lag_keep, lag_new;                      /* F16_model.py:53-56: u' = 0.9 u + 0.1 a * scale */
    double thrust_frac, thrust_max, thrust_unit;   /* :53: scale_T = 0.225 * 76300 / 0.3048 */
    double surf_max[3];                            /* :54-56: 45, 45, 45 (el, ail, rud) */
} np_f16_airframe;

/* Scenario constants = the keys of envs/configs/{heading,control,tracking}.yaml, with the
 * defaults of the reference's getattr(config, key, default) calls.  Values are passed as Python
 * holds them (double / int); the library rounds them to fp32 where the reference's tensor
 * arithmetic does. */
typedef struct np_f16_cfg {
    int32_t task;   /* NP_TASK_*   */
    int32_t solver; /* NP_SOLVER_* */
    double dt, airspeed, noise_scale;                               /* F16_model.py:15-17, heading_task.py:32 */
    double altitude_limit, acceleration_limit, max_velocity, min_velocity; /* termination_conditions/ __init__ */
    double min_alpha, max_alpha, min_beta, max_beta;                /* extreme_state.py:13-16 */
    int64_t max_check_interval, min_check_interval;                 /* unreach_heading.py:16-17 */
    double init_T, max_altitude, min_altitude, max_vt, min_vt;      /* F16_model.py:25-29 */
    double max_heading_increment, max_pitch_increment, max_velocities_u_increment; /* control_task.py:30-32 */
    double max_distance, min_distance;                              /* tracking_task.py:30-31 */
    /* Numerics option (not a reference key): evaluate the 22 single-input aero nets through their exact
     * piecewise-linear tables (PWL section of the version-2 weights blob) instead of the MLP FMA chains.
     * Same functions (a ReLU MLP of one input IS piecewise linear), 0.4 ppm from their fp64 value; results
     * differ from the default by ~1e-5 relative per coefficient, i.e. by the fp32 noise of the MLP itself. */
    int32_t aero_1d_tables;
    int32_t reserved_cfg_;
    np_f16_airframe airframe;   /* all zero = the F-16 literals (see np_f16_airframe) */
} np_f16_cfg;

/* Buffers of one reset()/step() call.  n aircraft, global row index of local row i = row0 + i
 * (used only to key the counter-based RNG, so that a batch sharded over several GPUs draws the
 * same numbers as the unsharded batch). */
typedef struct np_f16_io {
    float *s;              /* [12][ld]  F16Model.s            (F16_model.py:19)  in/out */
    float *u;              /* [5][ld]   F16Model.u            (F16_model.py:21)  in/out (row 4 = lef stays 0) */
    float *tgt;            /* [3][ld]   task.target_*         (heading_task.py:26-28) in/out */
    int64_t ld;            /* leading dimension of s/u/tgt (>= n, < 2^30: more rows than fit in 288 GB of HBM) */
    int64_t *step_count;   /* [n]       BaseEnv.step_count    (env_base.py:28) in/out */
    const uint8_t *done_in, *bad_in, *timeout_in;   /* [n] flags left by the previous step (env_base.py:31-33) */
    uint8_t *done_out, *bad_out, *timeout_out;      /* [n] new flags; may NOT alias the *_in buffers */
    const float *action;   /* [n][act_stride] row-major, columns 0..3 read (F16_model.py:52-56); NULL for reset */
    int64_t act_stride;
    float *obs;            /* [n][22] row-major (what the policy consumes); may be NULL for reset */
    float *reward;         /* [n]; unused by reset */
    /* Parity hooks (normally NULL -> in-kernel counter-based RNG keyed by seed/call_idx/row):   */
    const float *rand_u;   /* [n][5] uniforms (alt, vt, task0, task1, task2) consumed by flagged rows */
    const float *noise;    /* [n][22] standard normals added as obs + noise*noise_scale */
    /* Cross-step coefficient cache (optional, may be NULL): np_f16_cache_floats(n) floats owned by the caller,
     * layout private to the library (tiled per workgroup).  36 of the
     * 42 aero MLPs depend on (alpha, beta) only; their values after the integrator step are exactly what the
     * next step's integrator needs, so np_f16_step writes them here and, when cache_valid != 0, reads them
     * back instead of re-evaluating (results are bit-identical either way).  cache_valid = "the last np_f16_step / np_f16_reset on these
     * buffers wrote the cache" (0 for a fresh or foreign buffer).  Since ABI 14 the cache also records the (alpha, beta) its
     * coefficients belong to, and a step that finds a row's state edited since (by whatever path: the caller need not tell) re-evaluates
     * the coefficients of that wave before it goes on: clearing cache_valid after editing `s` is an optimisation (the whole batch then
     * runs the un-cached kernel), no longer a correctness requirement. */
    float *coef_cache;
    int32_t cache_valid;
    /* inner_step != 0 selects the semantics of ONE of the 50 low-level iterations inside PlanningEnv.step
     * (envs/planning_env.py:153-176): no auto-reset, rows whose *_in flags are already set keep their state
     * (`s[reset] = recent_s[reset]`, controls still advance), step_count += 1 for every row, and the *_out
     * flags ACCUMULATE (out = in | new), as BaseEnv.done does between two reset() calls (env_base.py:70-75).
     * inner_step == NP_INNER_UPDATE_ONLY (ABI 16): F16Model.update(action) on its own (envs/models/F16_model.py:51-67, called directly by
     * envs/planning_env.py:160 and example/quick_start.ipynb): clamp, control lag and the integrator for EVERY row — no auto-reset, no hold,
     * step_count untouched, *_out = *_in, no termination condition or reward reported (obs, reward, term_* may be NULL; obs, when given,
     * receives task.get_obs of the new state).  The cross-step coefficient cache is refreshed for the state reached, as by any step. */
    int32_t inner_step;
    uint64_t seed;         /* RNG key */
    uint64_t call_idx;     /* RNG counter word: the caller increments it once per reset()/step() call */
    int64_t row0;
    /* Optional DEVICE word added to call_idx inside the kernel (NULL = 0).  Lets a fixed sequence of launches be
     * captured in a HIP graph once and replayed: the captured call_idx values are offsets, the base advances on the
     * device between replays (PlanningEnv's 1 + 50 launches per step). */
    const uint64_t *call_idx_base;
    /* Optional DEVICE counters [NP_NUM_TERM_COUNTERS] (uint32, caller-zeroed, accumulated by np_f16_step): how many aircraft
     * tripped each termination condition at the state reached by the step — what the reference prints per condition
     * (`print(torch.sum(bad_done), ...)`, envs/termination_conditions/<condition>.py) at the price of a host sync each.  Order:
     * overload, low_altitude, high_speed, low_speed, extreme_state, unreach_* (bad), target reached (done). */
    uint32_t *term_counters;
    /* Optional DEVICE bytes [n] written by np_f16_step: the same conditions per aircraft (bit k = counter k above) at the state
     * reached by this step — which of the termination-condition classes (envs/termination_conditions/<condition>.py, called from
     * task_base.py:75-96) fired for which row.  NULL = not wanted.  np_f16_reset clears the bytes of every row (all flags are
     * cleared, no condition is evaluated); with inner_step set the bits accumulate (OR) over the launches of one PlanningEnv.step,
     * like the done / bad_done flags they explain. */
    uint8_t *term_reasons;
    /* Optional DEVICE floats [n] written by np_f16_step: the value of the task's own reward function (HeadingReward /
     * PostureReward / PositionReward, envs/reward_functions/<function>.py) before EventDrivenReward's -200 * bad_done + 200 * done is added
     * (task_base.py:60-73): `reward` = this + that, in fp32.  NULL = not wanted. */
    float *reward_task;
    /* PlanningEnv's inner loop (inner_step set), both optional: ll_tgt [3][ld] = the low-level controller's targets (pitch, heading,
     * vt: planning_env.py:150-152), ll_obs [n][22] receives PlanningEnv.low_level_obs (planning_env.py:60-142) of the state this
     * launch REACHES — the controller's input for the next inner iteration, written by the step kernel itself (same values as
     * np_f16_lowlevel_obs on the new state), so the 50 iterations of a macro-step need that kernel once instead of 50 times.  With
     * ll_obs set, `obs` may be NULL (the task observation of an intermediate inner iteration is never read: planning_env.py:153-176). */
    const float *ll_tgt;
    float *ll_obs;
} np_f16_io;

typedef struct np_f16_ctx np_f16_ctx;

int np_abi_version(void);
/* the reference's literals (the values an all-zero np_f16_airframe stands for) */
void np_f16_airframe_default(np_f16_airframe *out);
/* number of floats np_f16_io.coef_cache must hold for n aircraft */
int64_t np_f16_cache_floats(int64_t n);
const char *np_last_error(void);

/* Upload the 43-MLP asset blob (NPF16MLP v1, neuralplane_amd/assets/f16_aero_mlp.bin) and the
 * scenario constants to `device`.  Replaces F16Dynamics/hifi_F16 construction
 * (envs/models/F16/hifi_F16_AeroData.py:40-129) + parse_config (envs/utils/utils.py:12-27).
 * A context owns its packed weights (one device allocation): contexts created from different blobs — a second aircraft
 * type with the same 43-net topology, BASELINE's "second aero-table set, same kernel template" — coexist on a device;
 * scenario constants, task, solver and numerics options are per context as well. */
int np_f16_ctx_create(const void *weights_blob, size_t nbytes, const np_f16_cfg *cfg, int device, np_f16_ctx **out);
void np_f16_ctx_destroy(np_f16_ctx *ctx);

/* BaseEnv.reset() — envs/env_base.py:83-97 (F16Model.reset F16_model.py:33-45, task.reset
 * heading_task.py:49-69 / control_task.py:49-68 / tracking_task.py:48-71, then obs()).
 * Rows with any *_in flag set are re-initialised; all *_out flags are written as 0. */
int np_f16_reset(np_f16_ctx *ctx, int64_t n, const np_f16_io *io, void *stream);

/* BaseEnv.step(action) — envs/env_base.py:99-109: auto-reset, F16Model.update
 * (F16_model.py:51-67, one integrator step of F16Dynamics.nlplant F16_dynamics.py:37-228),
 * step_count += 1, task.get_obs, the six termination conditions (task_base.py:75-96) and the
 * reward (task_base.py:60-73), fused into ONE kernel launch. */
int np_f16_step(np_f16_ctx *ctx, int64_t n, const np_f16_io *io, void *stream);

/* Derived quantities behind F16Model's getters (F16_model.py:47-49, 132-198), SoA out[23][ld_out]:
 *   rows 0..11  get_extended_state()[:, :12]  (xdot)
 *   rows 12..14 get_acceleration()            (ax, ay, az)
 *   rows 15..17 get_accels()                  (nx_cg, ny_cg, nz_cg)
 *   row  18     get_EAS2TAS()
 *   row  19     get_EAS()
 *   rows 20..22 get_atmos()                   (mach, qbar, ps; F16_model.py:183-198)      */
int np_f16_derived(np_f16_ctx *ctx, int64_t n, const float *s, const float *u, int64_t ld, float *out,
                   int64_t ld_out, void *stream);

/* The aero coefficient surrogates on their own — hifi_F16.hifi_C(alpha, beta, el), hifi_damping(alpha), hifi_C_lef(alpha, beta),
 * hifi_damping_lef(alpha), hifi_rudder(alpha, beta), hifi_ailerons(alpha, beta), hifi_other_coeffs(alpha, el)
 * (envs/models/F16/hifi_F16_AeroData.py:745-822; inputs in DEGREES as F16_dynamics.py:135-137 passes them): out[43][ld_out],
 * rows in the reference's evaluation order
 *   0-5 Cx Cz Cm Cy Cn Cl | 6-14 Cxq Cyr Cyp Czq Clr Clp Cmq Cnr Cnp | 15-20 delta_{Cx,Cz,Cm,Cy,Cn,Cl}_lef |
 *   21-29 delta_{Cxq,Cyr,Cyp,Czq,Clr,Clp,Cmq,Cnr,Cnp}_lef | 30-32 delta_{Cy,Cn,Cl}_r30 |
 *   33-38 delta_Cy_a20, delta_Cy_a20_lef, delta_Cn_a20, delta_Cn_a20_lef, delta_Cl_a20, delta_Cl_a20_lef |
 *   39-42 delta_Cnbeta, delta_Clbeta, delta_Cm, eta_el
 * evaluated by the same device code (normalisation + net bodies, or the 1-D tables when the context has them on) that the
 * step kernels run.  Row 24 (delta_Czq_lef) is returned as 0: F16Dynamics.nlplant never reads it (F16_dynamics.py:199 uses
 * delta_Cz_lef) and the device weights do not carry it.  This is the entry the reference's surrogate check
 * (envs/models/F16/model/test_model.py:58-338: MLPs against the table values of model/coefs.csv) binds to. */
int np_f16_aero_coefficients(np_f16_ctx *ctx, int64_t n, const float *alpha_deg, const float *beta_deg, const float *el, float *out,
                             int64_t ld_out, void *stream);

/* PlanningEnv.low_level_obs(target_pitch, target_heading, target_vt) — envs/planning_env.py:60-142: the 22-float
 * observation of the low-level controller (same layout as ControlTask.get_obs, no noise) for caller-supplied
 * targets tgt3[3][ld].  obs: [n][22] row-major. */
int np_f16_lowlevel_obs(np_f16_ctx *ctx, int64_t n, const float *s, const float *u, const float *tgt3, int64_t ld,
                        float *obs, void *stream);
/* The prelude of PlanningEnv.step — envs/planning_env.py:146-152 — in one launch (ABI 14): action [n][act_stride >= 3] is clamped to
 * [-1, 1], tgt3[3][ld] = (pitch, heading, vt) + action * (0.3, 0.3, 30) (one fp32 product and one fp32 sum per element, as the reference's
 * torch expressions) is written, and obs [n][22] = low_level_obs(tgt3) as np_f16_lowlevel_obs computes it. */
int np_planning_targets_obs(np_f16_ctx *ctx, int64_t n, const float *s, const float *u, const float *action, int64_t act_stride, float *tgt3,
                            int64_t ld, float *obs, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * SingleCombat 1v1 (envs/singlecombat_env.py).  n = 2*num_envs aircraft; rows 2k / 2k+1 are the ego /
 * enemy aircraft of env k (singlecombat_env.py:98-99).  One launch = one SingleCombatEnv.step: pairwise
 * auto-reset, 5 x {demand filters, attitude PID stack (algorithms/pid/controller.py:43-74), one FDM step,
 * step_count += 1, terminations}, observation + reward at the final state, blood update.
 * ------------------------------------------------------------------------------------------------- */
#define NP_NUM_OBS_COMBAT 15 /* singlecombat_env.py:64-138 */
#define NP_NUM_COMBAT_TERM_COUNTERS 9
#define NP_NUM_PID 11        /* roll_dem, pitch_dem, {roll, pitch, yaw} x {error, integrator, last_out} */

/* algorithms/pid/config/{roll,pitch,yaw}controller.yaml */
typedef struct np_pid_gains {
    double Kp, Ki, Kd, Kff, Kimax, tau, rmax_pos, rmax_neg;
} np_pid_gains;

/* envs/configs/selfplay.yaml (+ the getattr defaults of singlecombat_env.py:29-44 and of the condition
 * classes), the three PID YAMLs and Controller.__init__'s airspeed bounds (controller.py:15) */
typedef struct np_f16_combat_cfg {
    int32_t solver;      /* NP_SOLVER_* */
    int32_t inner_steps; /* FDM steps per env.step (singlecombat_env.py:243: 5); 1..16 */
    double dt, airspeed;
    double altitude_limit, acceleration_limit, max_velocity, min_velocity;
    double min_alpha, max_alpha, min_beta, max_beta;
    double distance_limit;  /* crash.py:17 */
    int64_t max_steps;      /* timeout.py:15 */
    double init_T, target_dist;
    double max_altitude, min_altitude, max_vt, min_vt, max_heading, min_heading;
    double max_npos, min_npos, max_epos, min_epos;
    np_pid_gains roll, pitch, yaw;
    double roll_ff, gravity;           /* pitchcontroller.yaml */
    double airspeed_min, airspeed_max; /* controller.py:15 */
    int32_t aero_1d_tables;            /* see np_f16_cfg.aero_1d_tables */
    int32_t reserved_cfg_;
    np_f16_airframe airframe;          /* all zero = the F-16 literals */
} np_f16_combat_cfg;

typedef struct np_f16_combat_io {
    float *s;            /* [12][ld] in/out */
    float *u;            /* [5][ld]  in/out (T, el, ail, rud, lef) */
    float *pid;          /* [11][ld] in/out attitude-controller state; never reset by the env (as in the reference) */
    float *blood;        /* [ld]     in/out SingleCombatEnv.blood (singlecombat_env.py:45) */
    int64_t ld;
    int64_t *step_count; /* [n] */
    const uint8_t *done_in, *bad_in, *timeout_in;
    uint8_t *done_out, *bad_out, *timeout_out; /* may NOT alias the *_in buffers */
    const float *action; /* [n][act_stride]: throttle, roll, pitch, yaw demands of each aircraft; NULL for reset */
    int64_t act_stride;
    float *obs;          /* [n][15] row-major */
    float *reward;       /* [n] */
    const float *rand_u; /* parity hook: [n][5] uniforms (npos, epos, alt, yaw, vt) or NULL -> counter RNG */
    int32_t pid_first;   /* != 0 on the first step after the controller was created (PID.reset, pid.py:14,23-28) */
    int32_t reserved_io_;
    uint64_t seed, call_idx;
    int64_t row0;        /* global aircraft row of local row 0 (= 2 * first env of this shard) */
    /* optional DEVICE counters [NP_NUM_COMBAT_TERM_COUNTERS] (see np_f16_io.term_counters), one evaluation per inner FDM step:
     * overload, low_altitude, high_speed, low_speed, extreme_state, crash, timeout, shutdown (bad), shutdown (done) */
    uint32_t *term_counters;
    /* Split ("self-play") layout, both optional: the two halves of the self-play runner's data (runner/selfplay_F16sim_runner.py:62-67
     * `obs[:, :A//2]` / `obs[:, A//2:]`, :96-100 `actions = concatenate((ego, opponent), axis=1)`) as SEPARATE contiguous per-env
     * arrays, so that the opponent exchange (an all-gather over the ranks) reads and writes the kernel's own buffers — no split /
     * stack / contiguous copies around the launch.
     *   action_opp != NULL: `action` holds the EGO rows only, [num_envs][act_stride], `action_opp` the opponent rows, same stride
     *   obs_opp    != NULL: `obs` receives the EGO rows only, [num_envs][15], `obs_opp` the opponent rows [num_envs][15]
     * NULL (the default): the interleaved [n][...] layout above (rows 2k / 2k+1 = ego / opponent of env k). */
    const float *action_opp;
    float *obs_opp;
} np_f16_combat_io;

/* Same context type as np_f16_ctx_create; the weights blob is shared. */
int np_f16_combat_ctx_create(const void *weights_blob, size_t nbytes, const np_f16_combat_cfg *cfg, int device,
                             np_f16_ctx **out);
/* SingleCombatEnv.reset_done_envs + obs (singlecombat_env.py:183-238): both aircraft of every env in which a
 * *_in flag is set are re-initialised (blood = 100, step_count = 0); all *_out flags are written as 0. */
int np_f16_combat_reset(np_f16_ctx *ctx, int64_t num_envs, const np_f16_combat_io *io, void *stream);
/* SingleCombatEnv.step(action) (singlecombat_env.py:240-274), one kernel launch. */
int np_f16_combat_step(np_f16_ctx *ctx, int64_t num_envs, const np_f16_combat_io *io, void *stream);

/* PlanningEnv's frozen low-level controller: PPOActor.forward(obs, rnn_states, masks, deterministic=True)
 * (algorithms/ppo/ppo_actor.py:38-64 in the configuration of envs/planning_env.py:18-29: feature LayerNorm, MLP 22-128-128,
 * GRU 128, act MLP 128-128, tanh mean head) as ONE kernel launch.  `weights`: NP_ACTOR_NUM_FLOATS floats on the device, the
 * actor's state_dict in kernel order (neuralplane_amd/actor.py::pack_ppo_actor documents the layout).  obs [n][22],
 * h_in / h_out [n][128] (rnn_states with the layer dimension squeezed; may not alias; 16-byte aligned), masks [n],
 * actions [n][4].  Two tilings with identical results: 32-row tiles up to 16 384 rows (latency: 32 us per call up to 8 192
 * rows; needs `weights` 16-byte aligned), 64-row tiles above (throughput).  Environment variable NP_ACTOR_TILE=32|64, read at
 * every call, forces one (benchmarks and the parity tests of both).
 * Second numerics spec (ABI 15), "block fixed point": the six Linear layers with N >= 128 on the i8 matrix pipe (v_mfma_i32_32x32x32_i8) —
 * activations quantised per row to sign + 22 bits, weights per output feature to sign + 29 bits, integer limb products, exact class
 * sums combined by three fused multiply-adds (the CPU restatement f16_actor_i8.inc states it operation by operation; against the reference's recordings
 * it is as close as the fp32 chains: actions 5e-6, 150 closed-loop steps 4e-5).  It is selected by the WEIGHT BUFFER: np_actor_pack_i8
 * (host) turns the NP_ACTOR_NUM_FLOATS floats into NP_ACTOR_I8_NUM_FLOATS (the same floats, then per-output scales and the limb bytes in
 * the kernel's fragment order); upload that and pass its size as num_floats — here and in np_planning_loop.actor_weights_floats. */
#define NP_ACTOR_NUM_FLOATS 153392
#define NP_ACTOR_I8_NUM_FLOATS 309312
int np_actor_pack_i8(const float *packed_fp32_host, float *out_host);
int np_actor_forward(const float *weights, int64_t num_floats, int64_t n, const float *obs, const float *h_in, const float *masks,
                     float *actions, float *h_out, int device, void *stream);

/* The 50 low-level iterations of PlanningEnv.step (reference envs/planning_env.py:153-176: ego_actions = self.controller(low_level_obs,
 * rnn_states, masks); self.model.update(ego_actions); terminations / reward; next low_level_obs) enqueued by ONE call: per iteration
 * np_actor_forward on ll_obs[k & 1] / rnn[k & 1] -> ll_act, rnn[(k + 1) & 1], then np_f16_step with inner_step set, action = ll_act,
 * flags[k & 1] -> flags[(k + 1) & 1], which also writes the controller's next observation into ll_obs[(k + 1) & 1].  `io` carries what all
 * iterations share (s, u, tgt, ld, step_count, coef_cache, reward, term_*, reward_task, seed, row0, call_idx_base; call_idx = the first
 * iteration's, cache_valid = whether the cache is valid for the first iteration); io->obs receives the task observation of the LAST
 * iteration only (planning_env.py overwrites it every iteration) and the last iteration writes no low-level observation.  The results
 * are those of the 2 x iterations separate calls, bit for bit.
 * groups: the rows are processed as that many independent row groups (boundaries at multiples of 64 rows), group 0 on `stream`, the
 * others on streams the context owns, forked behind and joined back into `stream` by events: one group's controller call overlaps
 * another's FDM step and every group's controller call runs on the 32-row tiles.  0 = the library chooses by n (one group up to
 * 8 192 rows and from ~80 000 on, two or three in between: PlanningEnv.step 3.51 -> 3.23 ms at n = 1e4, 6.43 -> 4.44 ms at 20 000,
 * 10.7 -> 9.2 ms at 49 152; profiles/r03g_planning_groups.log).
 * A stream that is being captured into a graph always gets one group.
 * mode (ABI 13): NP_PLANNING_PERSISTENT runs ALL iterations in ONE kernel launch — a workgroup owns a tile of 32 rows and loops
 * {controller call, inner FDM step} with no kernel boundary in between (np_planning.hip; Euler solver, MLP numerics); with more
 * tiles than resident workgroups NP_PLANNING_PERSISTENT_QUEUE has the resident workgroups pull (tile, iteration) items from a device
 * counter instead, so that e.g. 313 tiles on 256 CUs take 62 rounds of items rather than two lock-step passes.  Both are bit-identical
 * to the launches they replace.  NP_PLANNING_PERSISTENT_GUESTS (ABI 14) is the static counterpart of the queue for resident < tiles <= 2 x
 * resident: every resident workgroup owns a tile, the remaining "guest" tiles are cut into blocks of iterations and each block is hosted
 * by a different workgroup between two stretches of its own tile (makespan = iterations + one block instead of 2 x iterations; `block`
 * = the slack in iterations per block index, 0 = the library chooses).  NP_PLANNING_PERSISTENT_DUAL (ABI 14) gives every eight-wave
 * workgroup TWO tiles: their controller calls run in lock-step on waves 0..3 / 4..7 and one 64-lane FDM step serves both (a single 32-row
 * tile fills the FDM code's 64-lane waves only half); any n, the choice above 1.5 tiles per CU.  NP_PLANNING_AUTO picks by n, solver and
 * numerics; environment NP_PLANNING_MODE=launches|persistent|queue|guests|dual (read per call) overrides it for benchmarks and the parity tests.
 * The guest and queue schedules cannot be captured into a graph (their counter / progress-word bases advance per launch on the host): the
 * call fails on a capturing stream, and NP_PLANNING_AUTO never picks them there.
 * Bounded waits (ABI 15).  In the guest and queue schedules a workgroup may wait for a tile's progress word, raised by another workgroup.
 * That ends when the grid is resident, or — observed, not promised by the architecture — when workgroups are dispatched in index order.
 * A wait that lasts longer than 2 s (environment NP_PLANNING_WAIT_MS) therefore gives up: the kernel drains and ENDS, and
 *   check = NP_PLANNING_CHECK_SYNC (0, the default): the call waits for the launch (it is the only part of this API that synchronises
 *     the stream; the copy of the in-place buffers it keeps meanwhile costs ~10 us per macro-step) and, had a wait expired, restores s, u,
 *     step_count, flags[0], rnn[0], ll_obs[0], coef_cache, term_reasons and term_counters to their values before the call and returns
 *     NP_E_PLANNING_STALLED with np_last_error() naming workgroup, tile and iteration: call again with mode = NP_PLANNING_LAUNCHES
 *     (neuralplane_amd/envs/planning_env.py does exactly that);
 *   check = NP_PLANNING_CHECK_DEFERRED: the call returns at once; the verdict is delivered by the context's NEXT np_planning_inner_loop call
 *     or by np_planning_check(ctx) (both wait for that launch first) as NP_E_PLANNING_STALLED_LOST — nothing was kept, the state buffers
 *     hold a partially advanced macro-step.
 * The static schedule (_PERSISTENT), the dual workgroups and the launches wait for nothing and are not affected. */
#define NP_E_PLANNING_STALLED 2
#define NP_E_PLANNING_STALLED_LOST 3
enum { NP_PLANNING_CHECK_SYNC = 0, NP_PLANNING_CHECK_DEFERRED = 1 };
enum { NP_PLANNING_AUTO = 0, NP_PLANNING_LAUNCHES = 1, NP_PLANNING_PERSISTENT = 2, NP_PLANNING_PERSISTENT_QUEUE = 3, NP_PLANNING_PERSISTENT_GUESTS = 4, NP_PLANNING_PERSISTENT_DUAL = 5 };
typedef struct np_planning_loop {
    int32_t iterations;        /* planning_env.py:153: 50 */
    int32_t groups;            /* 0 = automatic */
    const float *actor_weights;/* np_actor_forward's packed weights [NP_ACTOR_NUM_FLOATS] */
    float *ll_obs[2];          /* [n][22] each; [0] holds the first iteration's input (np_f16_lowlevel_obs) */
    float *rnn[2];             /* [n][128] each, 16-byte aligned; [0] holds the recurrent state on entry; the final state ends in rnn[iterations & 1] */
    const float *masks;        /* [n] */
    float *ll_act;             /* [n][4] scratch: the controller's actions of the current iteration */
    uint8_t *flags[2];         /* [3][n] each (done, bad_done, exceed_time_limit); [0] = the flags on entry; the final flags end in flags[iterations & 1] */
    const float *ll_tgt;       /* [3][ld] the controller's targets (np_f16_io.ll_tgt) */
    int32_t mode;              /* NP_PLANNING_AUTO / _LAUNCHES / _PERSISTENT / _PERSISTENT_QUEUE (ABI 13) / _PERSISTENT_GUESTS (ABI 14) */
    int32_t waves;             /* persistent kernel: waves per 32-row tile: 8 (0 = the library chooses; the four-wave builds were retired in ABI 15) */
    int32_t block;             /* queue schedule: iterations per (tile, block) item, 1 .. iterations; 0 = the library chooses */
    int32_t check;             /* ABI 15 (was reserved, 0): NP_PLANNING_CHECK_SYNC / _DEFERRED — guest and queue schedules only, see above */
    int64_t actor_weights_floats; /* ABI 15: what actor_weights holds — 0 or NP_ACTOR_NUM_FLOATS: the fp32 numerics; NP_ACTOR_I8_NUM_FLOATS: the
                                * block-fixed-point numerics (np_actor_pack_i8), every schedule except the dual workgroups */
} np_planning_loop;
int np_planning_inner_loop(np_f16_ctx *ctx, int64_t n, const np_f16_io *io, const np_planning_loop *loop, void *stream);
/* check = deferred: wait for the context's last guest / queue launch and return its verdict (0 or NP_E_PLANNING_STALLED_LOST). */
int np_planning_check(np_f16_ctx *ctx);

/* Returns of one rollout for the device-resident rollout storage — replaces ReplayBuffer.compute_returns
 * (reference algorithms/utils/buffer.py:139-173; a Python loop over time with numpy array operations there): one launch, a
 * backward scan per column, the reference's float32 arithmetic operation by operation.  T = buffer_size, N =
 * n_rollout_threads x num_agents; rewards [T][N]; value_preds, masks, bad_masks, returns [T+1][N] (row-major, device
 * pointers); next_value [N].  use_gae: value_preds[T] = next_value is written and returns[0..T-1] are the GAE returns
 * (returns[T] untouched); otherwise returns[T] = next_value and returns[t] are the discounted sums.  bad_masks may be NULL
 * unless use_proper_time_limits.  gamma / gae_lambda are the Python floats of the reference's argument bag. */
int np_rollout_returns(int64_t T, int64_t N, double gamma, double gae_lambda, int use_gae, int use_proper_time_limits,
                       const float *rewards, float *value_preds, const float *masks, const float *bad_masks, const float *next_value,
                       float *returns, int device, void *stream);

/* One collect step into the device-resident rollout storage (ABI 15) — F16SimRunner.insert (reference runner/F16sim_runner.py:131-154: the
 * recurrent states of the envs that ended are zeroed, masks = 0 where an env is done, bad_masks = 0 where it is bad_done, `any` over the env's
 * agents) + ReplayBuffer.insert (algorithms/utils/buffer.py:76-112: obs / masks / bad_masks / recurrent states into slot step + 1, actions /
 * rewards / log-probs / values into slot step) as ONE launch.  All pointers are device pointers; storage arrays are the buffer's, row-major
 * [T or T + 1][num_envs * num_agents][dim]; inputs are [num_envs * num_agents][dim]; done / bad_done / exceed_time_limit are bytes (bool). */
typedef struct np_rollout_step {
    int64_t num_envs, num_agents, step;
    int32_t obs_dim, act_dim, rnn_dim, reserved_;
    float *obs, *actions, *rewards, *masks, *bad_masks, *action_log_probs, *value_preds, *rnn_states_actor, *rnn_states_critic;
    const float *obs_in, *actions_in, *rewards_in, *action_log_probs_in, *values_in, *rnn_states_actor_in, *rnn_states_critic_in;
    const uint8_t *done_in, *bad_done_in, *exceed_time_limit_in;
} np_rollout_step;
int np_rollout_insert(const np_rollout_step *step, int device, void *stream);

/* The rollout policy's inference step (ABI 15) — PPOPolicy.get_actions (reference algorithms/ppo/ppo_policy.py:26-32; the call of
 * F16SimRunner.collect, runner/F16sim_runner.py:123-129): PPOActor.forward with sampled actions and their log-probabilities
 * (ppo_actor.py:38-64, act.py:56-101, distributions.py:36-44,77-101) and PPOCritic.forward (ppo_critic.py:38-50) on the same observation,
 * as ONE launch (the reference: ~110 small torch kernels per step).  For the networks the training scripts build (scripts/train_heading.sh:17,
 * train_tracking.sh:17: hidden "128 128", act-hidden "128 128", GRU 128 x 1, feature normalisation, ReLU — config.py's defaults) on 22
 * observations (15 for the 1v1 combat env's policies, runner/selfplay_F16sim_runner.py:76-100: np_policy_step.obs_dim) with 1..4
 * continuous actions — the frozen controller's shapes, so both networks travel in np_actor_forward's packed layout (NP_ACTOR_NUM_FLOATS
 * floats each, 16-byte aligned; a head narrower than four columns is zero-padded; the critic's value_out is column 0 of its head block:
 * neuralplane_amd/policy.py packs both from the state_dicts).  std = exp(log_std) and log_std [act_dim] are host values.
 *   noise [n][act_dim]: the standard normal draws of this step (the caller's generator — the reference's sample() draws them the same
 *   way); actions = fl(fl(noise * std) + tanh(mu)), or tanh(mu) with NP_POLICY_DETERMINISTIC (noise may be NULL then);
 *   log_probs [n] = sum over the actions of Normal(mean, std).log_prob(action); values [n]; rnn states [n][128], in != out.
 * flags choose the networks (get_actions = ACTOR | CRITIC; act = ACTOR [| DETERMINISTIC]; get_values = CRITIC); buffers of a network
 * that is not evaluated may be NULL.  Results equal the CPU restatement (f16_actor.inc; f16_actor_i8.inc for the second numerics) bit for
 * bit and the reference's recording within 2e-5 (measured: actions 1.1e-6 / 3.4e-6, values 1.4e-5 / 2.6e-5, log-probabilities 2e-6). */
enum { NP_POLICY_ACTOR = 1, NP_POLICY_CRITIC = 2, NP_POLICY_DETERMINISTIC = 4 };
typedef struct np_policy_step {
    int64_t n;
    int32_t act_dim, flags;
    const float *actor_weights, *critic_weights;
    float std[4], log_std[4];
    const float *obs, *masks, *noise, *rnn_states_actor_in, *rnn_states_critic_in;
    float *values, *actions, *action_log_probs, *rnn_states_actor_out, *rnn_states_critic_out;
    int64_t weights_floats;   /* what actor_weights / critic_weights hold, each: 0 or NP_ACTOR_NUM_FLOATS = the fp32 chains; NP_ACTOR_I8_NUM_FLOATS =
                               * np_actor_pack_i8's output: both networks in the block-fixed-point numerics (as np_actor_forward) */
    int32_t obs_dim;          /* observations per row: 0 or 22 (control / heading / tracking: envs/configs/*.yaml num_observation), or 15 (the 1v1 combat
                               * env, selfplay.yaml) — obs is [n][obs_dim]; the packed layout is the same, the first layer's rows >= obs_dim are zero */
    int32_t reserved_;
    /* optional (NULL: off) — the runner's insert rule applied by this launch instead of a separate one (F16SimRunner.insert, reference
     * runner/F16sim_runner.py:131-154): prev_flags [3][n] = done, bad_done, exceed_time_limit of the env step that produced `obs`.  Then `masks`
     * is not read: masks_out[n] = !done and bad_masks_out[n] = !bad_done are WRITTEN (the rollout storage's slot), the recurrent state of
     * every env with any flag set is taken as zero and zeroed IN PLACE in rnn_states_*_in (which therefore must be writable: the slot the
     * previous launch wrote as its rnn_states_*_out).  Results equal np_rollout_insert followed by this call without prev_flags, bit for bit. */
    const uint8_t *prev_flags;
    float *masks_out, *bad_masks_out;
} np_policy_step;
int np_policy_act(const np_policy_step *step, int device, void *stream);

/* np_f16_step has five bit-identical kernel variants: "latency" (four waves share a tile of 64 aircraft and split the 44 net
 * evaluations of a step, the serial fp64 chains of the state and the observation noise; chosen automatically for n <= 49152 —
 * one generation of 768 tiles at three waves per SIMD; "latency4w" = the same kernel built for four waves per SIMD, 1 024 tiles in
 * one generation, chosen for 49152 < n <= 65536 — Euler solver), "latency8" (eight waves per tile — two per SIMD, so one
 * wave's scalar-load latency is the other's FMA time; chosen automatically for n <= 16384, where every tile still has a CU of
 * its own), "latency2" (two waves per tile: two to three waves on every SIMD for 65536 < n <= 98304, where the pair variant
 * leaves a SIMD one or two), "pair" (the two waves of a 128-aircraft
 * workgroup split the nets and evaluate them for each other's aircraft: half the scalar weight traffic per aircraft; the default
 * above that size with the MLP numerics; built twice, for two and for three waves per SIMD — the second with ~20 cold registers per
 * lane in scratch — and picked per grid size, environment NPF16_PAIR_WAVES=2|3 pins one) and "throughput" (two independent waves per
 * workgroup; the 1-D table mode).  This call pins the choice for a context (tests, tuning); np_f16_combat_step has latency, pair and
 * throughput. */
enum { NP_KERNEL_AUTO = 0, NP_KERNEL_LATENCY = 1, NP_KERNEL_THROUGHPUT = 2, NP_KERNEL_PAIR = 3, NP_KERNEL_LATENCY8 = 4, NP_KERNEL_LATENCY2 = 5, NP_KERNEL_LATENCY4W = 6,
       NP_KERNEL_DUAL8 = 7, NP_KERNEL_DUAL4 = 8 /* SingleCombat contexts only: eight / four waves per 128-aircraft tile (Euler, MLP numerics; otherwise the automatic rule applies) */ };
int np_f16_set_kernel_variant(np_f16_ctx *ctx, int variant);

/* Which variant / tiling / row grouping a launch of n rows gets on a device with num_cus compute units (ABI 14): the selection
 * np_f16_step, np_f16_combat_step, np_actor_forward and np_planning_inner_loop apply, as a pure function (no device needed, the NPF16_*
 * environment overrides not applied).  Every threshold is a number of tiles per CU or wave slots per SIMD, measured on the 256-CU device
 * and scaled by multiProcessorCount (csrc/np_dispatch.h), so a partitioned MI355X (CPX 32 CUs, DPX 128) takes the same decisions per CU.
 * step: 1 = np_f16_step, 0 = np_f16_reset; solver: 0 euler, 1 rk4; tables: the 1-D table numerics; variant: NP_KERNEL_*. */
typedef struct np_dispatch_info {
    int32_t pair, pair3, latency, latency8, latency2, latency4w;  /* np_f16_step: which kernel family / build */
    int32_t block;            /* threads per workgroup */
    int32_t planning_groups;  /* np_planning_inner_loop, launch-by-launch mode: row groups */
    int32_t actor_tile32;     /* np_actor_forward: 32-row tiles (1) or 64-row tiles (0) */
    int32_t combat_latency;   /* np_f16_combat_step with n aircraft: 0 = pair / throughput, 1 = latency variant (four waves per 64 aircraft), 2 / 3 = dual8 / dual4 (eight / four waves per 128-aircraft tile, two-set bodies; automatic choice, Euler, MLP numerics only) */
    int64_t grid;             /* workgroups of the np_f16_step launch */
    int32_t planning_mode;    /* np_planning_inner_loop, NP_PLANNING_AUTO with the fused controller (Euler, MLP numerics): the NP_PLANNING_* it resolves to,
                               * assuming one resident eight-wave workgroup per CU */
    int32_t planning_mode_i8; /* ABI 15 (was reserved): the same with the block-fixed-point controller weights (no dual workgroups: queue / guests there) */
} np_dispatch_info;
int np_dispatch_plan(int64_t n, int32_t num_cus, int32_t step, int32_t solver, int32_t tables, int32_t variant, np_dispatch_info *out);

/* Average duration in ms of the `count` most recent np_f16_step / np_f16_combat_step launches on this context, measured with
 * HIP events on the launch stream: a start / stop pair attached to each kernel's own dispatch (hipExtLaunchKernelGGL — the
 * timestamps of the kernel's packet, no extra barrier packets between back-to-back launches; 0 disables, enable with
 * np_f16_set_timing(ctx, 1), which starts a new series; disabling keeps the series recorded so far readable).  Synchronises on
 * the recorded events.  A caller that enables timing and never polls is bounded:
 * after 8192 unresolved launches the next launch resolves them itself (it waits for them), and the per-launch sample list
 * below stops growing at 2^20 entries (the running average continues). */
int np_f16_set_timing(np_f16_ctx *ctx, int enable);
int np_f16_get_timing(np_f16_ctx *ctx, double *avg_ms, int64_t *count);
/* The individual durations (ms, launch order) behind np_f16_get_timing since timing was enabled: the first
 * min(capacity, *count) of them are copied to ms_out; *count = launches timed.  For the median / first-launch
 * numbers bench.py reports beside the mean (the reference's own benchmark, envs/measure_env.py:65-78, reports a mean only). */
int np_f16_get_timing_samples(np_f16_ctx *ctx, float *ms_out, int64_t capacity, int64_t *count);

/* Self-check of the numerics spec on the device at hand: divisions by a constant c run as q = x*rc; r = fma(-q, c, x);
 * q' = fma(r, rc, q) instead of the 12-instruction IEEE sequence.  This sweeps ALL 2^32 bit patterns of x
 * and compares with the IEEE quotient: counts3[0] = mismatches with a normal quotient and |x| >= 2^-100 (the spec promises 0),
 * counts3[1] = mismatches with a denormal / underflowing quotient or |x| < 2^-100 (last place only), counts3[2] = inputs compared
 * (2^32).  Blocking; about 10 ms per constant. */
int np_selfcheck_divc(float c, uint64_t *counts3, int device);

/* Profiling hook: while a device buffer of capacity_workgroups x NP_TRACE_WORDS uint64 is set (NULL clears), every
 * np_f16_step launch on this context writes one record per workgroup: [0] shader-clock counter at entry, [1] after the
 * de-phasing delay, [2] at exit, [3] / [4] the constant-rate (100 MHz) counter at entry / exit, [5] XCC id << 32 | HW_ID.
 * capacity must cover ceil(n / 64) workgroups.  tools/wg_timeline.py turns the records into the per-CU occupancy timeline
 * (start-up, tail, effective shader clock) quoted in DESIGN.md. */
#define NP_TRACE_WORDS 6
int np_f16_set_trace(np_f16_ctx *ctx, uint64_t *dev_buf, int64_t capacity_workgroups);

#ifdef __cplusplus
}
#endif
#endif
