/*
 * oracle/f16_oracle.h — CPU restatement of the NeuralPlane F-16 env.step hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle: a plain-C, scalar, fp32 restatement of the
 * reference's algorithm (xuecy22/NeuralPlane @ 2024-12-18; file:line citations are in f16_oracle.c
 * next to each function).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it; the product (neuralplane_amd/) never does and has no CPU fallback.
 *
 * Parity pin: tests/golden/ (.npz) are produced by importing the reference itself in the build
 * container (tools/gen_golden.py); tests/test_oracle_vs_golden.py checks this file against them,
 * including a "pin mode" in which everything except the fp32 operation order is made
 * implementation-independent so the comparison is bit-exact.
 *
 * Layout here is the REFERENCE's (array-of-structs, row-major [n][k]) on purpose: the oracle
 * restates the reference, not the HIP kernel's SoA layout.
 */
#ifndef F16_ORACLE_H
#define F16_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define F16O_NUM_NETS 43
#define F16O_NS 12 /* states   */
#define F16O_NU 5  /* controls */
#define F16O_NOBS 22

enum { F16O_TASK_HEADING = 0, F16O_TASK_CONTROL = 1, F16O_TASK_TRACKING = 2 };
enum { F16O_SOLVER_EULER = 0, F16O_SOLVER_RK4 = 1 };

/* YAML values exactly as Python holds them (doubles); rounded to fp32 where the reference's
 * tensor arithmetic rounds them (envs/configs/ YAML, read via getattr(config, key, default)). */
typedef struct f16o_cfg {
    int32_t task;   /* F16O_TASK_*   */
    int32_t solver; /* F16O_SOLVER_* */
    double dt, airspeed, noise_scale;
    double altitude_limit, acceleration_limit, max_velocity, min_velocity;
    double min_alpha, max_alpha, min_beta, max_beta;
    int64_t max_check_interval, min_check_interval;
    double init_T, max_altitude, min_altitude, max_vt, min_vt;
    double max_heading_increment, max_pitch_increment, max_velocities_u_increment; /* control */
    double max_distance, min_distance;                                              /* tracking */
} f16o_cfg;

typedef struct f16o_model f16o_model;

/* The airframe as data (device twin: include/neuralplane_amd.h::np_f16_airframe, same fields): the literals of F16Dynamics.nlplant / atmos
 * (envs/models/F16/F16_dynamics.py:22-35,61-76,114-116) and of F16Model.update (envs/models/F16_model.py:52-62).  All zero / NULL = the
 * F-16.  A model carries one (f16o_model_set_airframe); every function that evaluates the dynamics reads it from there.  Anything but
 * the defaults is parity-UNPINNED: the reference has no second aircraft. */
typedef struct f16o_airframe {
    double g, mass, B, S, cbar, xcgr, xcg, Heng, Jy, Jxz, Jz, Jx, ail_ref, rud_ref, atm_lapse, atm_exp, rho0;
    double lag_keep, lag_new, thrust_frac, thrust_max, thrust_unit, surf_max[3];
} f16o_airframe;
void f16o_airframe_default(f16o_airframe *a);
void f16o_model_set_airframe(f16o_model *m, const f16o_airframe *a);

/* Parse the NPF16MLP v1 blob (tools/export_weights.py).  Returns NULL on malformed input. */
f16o_model *f16o_model_load(const void *blob, size_t nbytes);
void f16o_model_free(f16o_model *m);

/* mode bits (f16o_set_mode): default 0 = the shipped numerics spec. */
#define F16O_MODE_MLP_F64 1  /* pin mode: MLPs evaluated in fp64 from fp32 inputs, rounded once */
#define F16O_MODE_LIBM 2     /* transcendental functions from the host libm (sinf/cosf/tanf/powf) */
#define F16O_MODE_PWL 4      /* single-input nets through their exact piecewise-linear tables (blob PWL section) */
#define F16O_MODE_DIV_IEEE 8 /* divisions by constants as plain IEEE `x / c` (the reference's operator) instead of the spec's
                              * Markstein sequence f16o_divc — identical results wherever f16o_divc_check holds (tests run both) */
#define F16O_MODE_BIAS_LAST 16 /* EXPERIMENT (never the spec, never compared with the kernels): the aero MLPs' Linear layers as acc = 0; fmaf chain;
                               * + bias — the order of a GEMM with a bias epilogue (profiles/r06h_mlp_order_probe.log) */
void f16o_set_mode(int mode);
int f16o_get_mode(void);

/* 43 aero coefficients for one (alpha_deg, beta_deg, el_deg); order = blob order. */
void f16o_aero(const f16o_model *m, float alpha_deg, float beta_deg, float el, float out[F16O_NUM_NETS]);

/* xdot[n][12] = F16Dynamics.nlplant(x[n][17])[:, :12] */
void f16o_nlplant(const f16o_model *m, int64_t n, const float *x17, float *xdot12);

/* getters of F16Model that need the dynamics (s[n][12], u[n][5]) */
void f16o_get_acceleration(const f16o_model *m, int64_t n, const float *s, const float *u, float *a3);
void f16o_get_accels(const f16o_model *m, int64_t n, const float *s, const float *u, float *n3);
void f16o_get_eas2tas(const f16o_model *m, int64_t n, const float *s, float *out);
void f16o_get_atmos(const f16o_model *m, int64_t n, const float *s, float *out3); /* (mach, qbar, ps) per row */

/* Elementary functions of the numerics spec, exposed for unit tests. */
void f16o_sincos(float x, float *s, float *c);
float f16o_tan(float x);
float f16o_pow(float x, float y);
float f16o_wrap_pi(float x);
void f16o_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* reset uniforms u[8] and 22 standard normals for (seed, call_idx, global_row) */
void f16o_rng_uniforms(uint64_t seed, uint64_t call_idx, int64_t row, float u8[8]);
void f16o_rng_normals(uint64_t seed, uint64_t call_idx, int64_t row, float z22[F16O_NOBS]);

/*
 * One BaseEnv.reset() (env_base.py:83-97).
 *   flags in/out: done/bad/timeout uint8[n]; rows with any flag set are re-initialised, then ALL
 *   flags are cleared.  rand_u: NULL -> counter-based RNG (seed, call_idx, row0+i); else
 *   float[n][5] = (U_alt, U_vt, U_task0, U_task1, U_task2) consumed for flagged rows only.
 *   noise: NULL -> counter-based RNG (skipped entirely when noise_scale==0); else float[n][22]
 *   standard normals (obs + noise*noise_scale, as the reference writes it).
 */
int f16o_reset(const f16o_model *m, const f16o_cfg *cfg, int64_t n, float *s, float *u, float *tgt,
               int64_t *step_count, uint8_t *done, uint8_t *bad, uint8_t *timeout, const float *rand_u,
               const float *noise, uint64_t seed, uint64_t call_idx, int64_t row0, float *obs);

/*
 * One BaseEnv.step(action) (env_base.py:99-109): auto-reset -> control lag + integrator ->
 * step_count+=1 -> obs -> terminations -> reward.  action: float[n][act_stride], columns 0..3 used.
 * On return done/bad/timeout hold the new flags; obs float[n][22]; reward float[n].
 */
int f16o_step(const f16o_model *m, const f16o_cfg *cfg, int64_t n, float *s, float *u, float *tgt,
              int64_t *step_count, uint8_t *done, uint8_t *bad, uint8_t *timeout, const float *action,
              int64_t act_stride, const float *rand_u, const float *noise, uint64_t seed, uint64_t call_idx,
              int64_t row0, float *obs, float *reward);

/* One of the 50 low-level iterations inside PlanningEnv.step (envs/planning_env.py:153-176): no auto-reset,
 * rows whose flags are already set keep their state (controls still advance), flags accumulate. */
int f16o_step_inner(const f16o_model *m, const f16o_cfg *cfg, int64_t n, float *s, float *u, float *tgt,
                    int64_t *step_count, uint8_t *done, uint8_t *bad, uint8_t *timeout, const float *action,
                    int64_t act_stride, const float *noise, uint64_t seed, uint64_t call_idx, int64_t row0,
                    float *obs, float *reward);
/* F16Model.update(action) / F16Model.reset(env) on their own (envs/models/F16_model.py:51-67, :33-45) — see f16_oracle.c */
int f16o_update(const f16o_model *m, const f16o_cfg *cfg, int64_t n, float *s, float *u, const float *action, int64_t act_stride);
int f16o_model_reset(const f16o_cfg *cfg, int64_t n, float *s, float *u, const uint8_t *done, const uint8_t *bad, const uint8_t *timeout,
                     const float *rand_u, uint64_t seed, uint64_t call_idx, int64_t row0);
/* per-condition termination bits of the state (s, u, tgt, step_count) — see f16_oracle.c */
void f16o_termination_reasons(const f16o_model *m, const f16o_cfg *cfg, int64_t n, const float *s, const float *u,
                              const float *tgt, const int64_t *step_count, uint8_t *reasons);
/* PlanningEnv.low_level_obs (envs/planning_env.py:60-142); tgt3[n][3] = (target_pitch, target_heading, target_vt) */
void f16o_lowlevel_obs(const f16o_model *m, const f16o_cfg *cfg, int64_t n, const float *s, const float *u, const float *tgt3, float *obs);

/* ------------------------------------------------------------------------------------------ */
/* SingleCombat 1v1 restatement (f16_combat.inc) — envs/singlecombat_env.py + algorithms/pid/  */
/* ------------------------------------------------------------------------------------------ */
#define F16O_NPID 11        /* roll_dem, pitch_dem, {roll,pitch,yaw} x {error, integrator, last_out} */
#define F16O_NOBS_COMBAT 15
#define F16O_NUM_COMBAT_TERM 9 /* overload, low_altitude, high_speed, low_speed, extreme_state, crash, timeout, shutdown(bad), shutdown(done) */

/* algorithms/pid/config/{roll,pitch,yaw}controller.yaml */
typedef struct f16o_pid_gains {
    double Kp, Ki, Kd, Kff, Kimax, tau, rmax_pos, rmax_neg;
} f16o_pid_gains;

/* envs/configs/selfplay.yaml + the PID YAMLs + Controller.__init__ defaults (controller.py:15-23) */
typedef struct f16o_combat_cfg {
    int32_t solver, inner_steps;
    double dt, airspeed;
    double altitude_limit, acceleration_limit, max_velocity, min_velocity;
    double min_alpha, max_alpha, min_beta, max_beta;
    double distance_limit;
    int64_t max_steps;
    double init_T, target_dist;
    double max_altitude, min_altitude, max_vt, min_vt, max_heading, min_heading;
    double max_npos, min_npos, max_epos, min_epos;
    f16o_pid_gains roll, pitch, yaw;
    double roll_ff, gravity;
    double airspeed_min, airspeed_max;
} f16o_combat_cfg;

float f16o_acos(float x);
float f16o_atanh(float x);
float f16o_exp(float x);
/* pure pairwise functions (envs/utils/utils.py:156-249); out11 per pair = AO, TA, R, AO2d, TA2d, R2d, side,
 * orientation_reward('v2'), range_reward('v3', R km), orientation_fn(AO), distance_fn(R km) */
void f16o_pairwise(int64_t n, const float *ego_pos, const float *enm_pos, const float *ego_vel, const float *enm_vel,
                   float target_dist, float *out11);
/* Controller.stabilize for n aircraft (s[n][12], pid[n][F16O_NPID] in/out); out3 = (el, ail, rud) outputs */
void f16o_stabilize(const f16o_model *m, const f16o_combat_cfg *cfg, int64_t n, const float *s, float *pid, int first, float *out3);
/* Pairwise reset: both aircraft of an env in which any flag is set are re-initialised (rand_u[n][5] =
 * U_npos, U_epos, U_alt, U_yaw, U_vt or NULL -> counter RNG keyed by global aircraft row 2*env0+i), then all
 * flags are cleared; obs[n][15] may be NULL. */
int f16o_combat_reset(const f16o_model *m, const f16o_combat_cfg *cfg, int64_t num_envs, float *s, float *u, float *blood,
                      int64_t *step_count, uint8_t *done, uint8_t *bad, uint8_t *timeout, const float *rand_u, uint64_t seed,
                      uint64_t call_idx, int64_t env0, float *obs);
/* One SingleCombatEnv.step (n = 2*num_envs aircraft, rows 2k = ego, 2k+1 = enemy of env k). */
int f16o_combat_step(const f16o_model *m, const f16o_combat_cfg *cfg, int64_t num_envs, float *s, float *u, float *pid,
                     float *blood, int64_t *step_count, uint8_t *done, uint8_t *bad, uint8_t *timeout, const float *action,
                     int64_t act_stride, const float *rand_u, int pid_first, uint64_t seed, uint64_t call_idx, int64_t env0,
                     float *obs, float *reward, uint32_t *term_counts /* nullable [F16O_NUM_COMBAT_TERM], accumulated */);

/* PlanningEnv's frozen low-level controller (f16_actor.inc): PPOActor.forward(obs[n][22], rnn_states[n][128], masks[n],
 * deterministic=True) -> act[n][4], h_out[n][128]; w = f16o_actor_num_floats() packed weights (neuralplane_amd/actor.py) */
int f16o_actor_num_floats(void);
void f16o_actor_forward(const float *w, int64_t n, const float *obs, const float *h_in, const float *mask, float *act,
                        float *h_out);

/* The rollout policy's inference step (f16_actor.inc): PPOPolicy.get_actions (algorithms/ppo/ppo_policy.py:26-32) — actor with sampled
 * actions + log-probabilities and critic on the same observation.  wa / wc: the two networks as f16o_actor_num_floats() packed weights
 * (the critic's value_out in column 0 of the head block); obs [n][obs_dim], obs_dim = 22 or fewer (15: the 1v1 combat env); std / log_std
 * [act_dim]; noise [n][act_dim] the standard normal draws;
 * flags: 1 actor, 2 critic, 4 deterministic.  values [n], actions [n][act_dim], log_probs [n], ha_out / hc_out [n][128]. */
void f16o_policy_act(const float *wa, const float *wc, const float *std, const float *log_std, int64_t n, int obs_dim, int act_dim, int flags,
                     const float *obs, const float *ha_in, const float *hc_in, const float *mask, const float *noise, float *values,
                     float *actions, float *log_probs, float *ha_out, float *hc_out);

/* the same with both networks in the controller's block-fixed-point numerics (f16_actor_i8.inc); returns non-zero when out of memory */
int f16o_policy_act_i8(const float *wa, const float *wc, const float *std, const float *log_std, int64_t n, int obs_dim, int act_dim, int flags,
                       const float *obs, const float *ha_in, const float *hc_in, const float *mask, const float *noise, float *values,
                       float *actions, float *log_probs, float *ha_out, float *hc_out);

/* ReplayBuffer.compute_returns (f16_rollout.inc; reference algorithms/utils/buffer.py:139-173): rewards [T][N], value_preds / masks /
 * bad_masks / returns [T+1][N], next_value [N]; GAE modes write value_preds[T], the others returns[T] */
void f16o_rollout_returns(int64_t T, int64_t N, double gamma, double gae_lambda, int use_gae, int proper, const float *rewards,
                          float *value_preds, const float *masks, const float *bad_masks, const float *next_value, float *returns);

int f16o_num_threads(void);
void f16o_set_threads(int n); /* OpenMP threads used by the batched entry points */

#ifdef __cplusplus
}
#endif
#endif
