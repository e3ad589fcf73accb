// np_f16_combat.h — fused SingleCombat (1v1) macro-step kernel for gfx950.
//
// Reference: envs/singlecombat_env.py:64-274 (obs, reward, reset_done_envs, step), the attitude PID stack
// algorithms/pid/{controller,pid,rollController,pitchController,yawController}.py and
// envs/termination_conditions/{crash,shutdown,timeout}.py.  The reference env file is stale against today's
// BaseEnv; DESIGN.md §10 lists the composition decisions.
//
// Mapping: one lane per aircraft; the two aircraft of an env are ADJACENT lanes (rows 2k, 2k+1), so every
// pairwise quantity (Crash distance, AO/TA/R, blood) is one DPP/bpermute lane exchange — no LDS, no second
// kernel.  One launch runs all `inner_steps` FDM steps of an env.step with the state in registers: HBM is
// touched once per env.step (not once per FDM step), and the 14 force-side alpha/beta-only aero
// coefficients the Overload check evaluates at the new state stay in this lane's LDS column for the next
// inner step's integrator (the same reuse the single-aircraft kernel routes through its HBM cache).
#pragma once
#include "np_f16_device.h"

namespace npf16 {

struct PidDev {
    float Kp, Ki, Kd, Kff, Kimax, tau, rmax_pos, rmax_neg;
    int ki_on;  // `self.Ki != 0 and self.dt > 0` (pid.py:38)
};

struct CombatDevCfg {
    float dt;        // integrator step t1 - t0 (F16_model.py:66)
    float dt_pid;    // Controller(dt=...) (singlecombat_env.py:46)
    float airspeed;
    float altitude_limit, acceleration_limit, max_velocity, min_velocity;
    float min_alpha, max_alpha, min_beta, max_beta;
    float dist_limit_sq;
    long long max_steps;
    float init_T;
    float alt_span, min_altitude, vt_span, min_vt, yaw_span, min_heading, npos_span, min_npos, epos_span, min_epos;
    PidDev roll, pitch, yaw;
    float roll_ff, gravity, scale_min, scale_max;
    int inner_steps;
    int aero_1d_tables;
    Airframe af;   // np_f16_combat_cfg.airframe (defaults: the F-16 literals)
};

struct CombatArgs;
typedef const CombatArgs __attribute__((address_space(4))) *CombatArgsC;
#ifndef NP_REREAD_ARGS
#define NP_REREAD_ARGS(ap) asm volatile("" : "+s"(ap) : : "memory")
#endif

struct CombatArgs {
    float *s, *u, *pid, *blood;
    long long ld;
    long long *step_count;
    const uint8_t *fin0, *fin1, *fin2;
    uint8_t *fout0, *fout1, *fout2;
    const float *action;
    long long act_stride;
    float *obs, *reward;
    const float *rand_u;
    int pid_first;
    uint64_t seed, call_idx;
    long long row0, n;
    unsigned *term_counters;  // optional [NP_NUM_COMBAT_TERM_COUNTERS]
    AeroWeights wt;
    CombatDevCfg cfg;
    // split (self-play) layout, np_f16_combat_io: per-env ego / opponent halves in separate contiguous arrays
    const float *action_opp;  // non-null: `action` = ego rows [E][act_stride], this = opponent rows
    float *obs_opp;           // non-null: `obs` = ego rows [E][15], this = opponent rows
};

enum { PID_ROLL_DEM = 0, PID_PITCH_DEM, PID_R_ERR, PID_R_INT, PID_R_LAST, PID_P_ERR, PID_P_INT, PID_P_LAST, PID_Y_ERR,
       PID_Y_INT, PID_Y_LAST, NUM_PID };

__device__ __forceinline__ float clampf(float v, float lo, float hi) {  // torch.clamp: NaN stays NaN
    v = v < lo ? lo : v;
    return v > hi ? hi : v;
}
__device__ __forceinline__ float partner(float v) { return __shfl_xor(v, 1); }

// {Roll,Pitch,Yaw}Controller.get_rate_out + PID.update_all / update_i (pid.py:18-42).  A row whose target or
// measurement is non-finite holds its previous output (the reference skips the update of the whole batch).
template <class PID>  // PidDev, generic or constant address space
__device__ __forceinline__ float rate_out(const PID &g, float dt, float desired, float scaler, float eas2tas, float rate,
                                          float &err, float &integ, float &last_out, bool strict, bool first) {
    const bool limit = strict ? fabsf(last_out) > 45.0f : fabsf(last_out) >= 45.0f;
    const float target = (desired * scaler) * scaler;
    const float meas = (rate * scaler) * scaler;
    const bool ok = ((target - target) == 0.0f) && ((meas - meas) == 0.0f);
    const float held = clampf(last_out, -45.0f, 45.0f);
    const float last_error = err;
    const float e = target - meas;
    const float deriv = first ? 0.0f : (e - last_error) / dt;
    float in = first ? 0.0f : integ;
    if (g.ki_on) {
        const bool gate = (!limit) | ((e * dt) < 0.0f);
        in = in + ((e * g.Ki) * dt) * (gate ? 1.0f : 0.0f);
        in = clampf(in, -g.Kimax, g.Kimax);
    } else {
        in = 0.0f;
    }
    const float ff = (target * g.Kff) / (scaler * eas2tas + 1e-8f);
    float out = ((ff + e * g.Kp) + in) + deriv * g.Kd;
    out = NP_DIVC(180.0f * out, 3.14159265358979323846f);
    err = ok ? e : err;
    integ = ok ? in : integ;
    last_out = ok ? out : last_out;
    return ok ? clampf(out, -45.0f, 45.0f) : held;
}

// Controller.stabilize (controller.py:35-74) for one aircraft; rates = Euler-angle rates xdot[3..5]
template <class CFG>  // CombatDevCfg, or CombatDevCfg in the constant address space (scalar loads)
__device__ __forceinline__ void stabilize(const CFG &cfg, const float (&s)[12], const Trig &tr, float tt,
                                          float (&pid)[NUM_PID], bool first, float &el, float &ail, float &rud) {
    const float PI_F = 3.14159265358979323846f;
    const float P = s[9], Q = s[10], R = s[11];
    const float roll_rate = P + tt * (Q * tr.sphi + R * tr.cphi);   // F16_dynamics.py:136-138
    const float pitch_rate = Q * tr.cphi - R * tr.sphi;
    const float yaw_rate = (Q * tr.sphi + R * tr.cphi) / tr.ct;
    const float eas2tas = eas2tas_of(cfg.af, s[2]);
    const float TAS = s[6] + cfg.airspeed * 1.0f;
    float scaler = (1.0f / (TAS + 1e-8f)) * 1000.0f;                // calc_speed_scaler :35-40
    scaler = clampf(scaler, cfg.scale_min, cfg.scale_max);
    const float roll = s[3], pitch = s[4];
    {   // stabilize_roll :42-46, rollController.py:43-49
        const float angle_err = np_wrap_pi(pid[PID_ROLL_DEM] - roll);
        float desired = angle_err / cfg.roll.tau;
        if (cfg.roll.rmax_pos != 0.0f) desired = clampf(desired, -cfg.roll.rmax_pos, cfg.roll.rmax_pos);
        ail = rate_out(cfg.roll, cfg.dt_pid, desired, scaler, eas2tas, roll_rate, pid[PID_R_ERR], pid[PID_R_INT], pid[PID_R_LAST],
                       false, first);
    }
    {   // stabilize_pitch :48-52, pitchController.py:47-94
        const float angle_err = np_wrap_pi(pid[PID_PITCH_DEM] - pitch);
        float desired = angle_err / cfg.pitch.tau;
        const float pio2 = (float)(3.141592653589793 / 2.0);
        const bool m1 = fabsf(roll) < pio2, m2 = roll >= pio2, m3 = roll <= -pio2;
        const float r1 = clampf(roll, (float)(-4.0 * 3.141592653589793 / 9.0), (float)(4.0 * 3.141592653589793 / 9.0));
        const float r2 = clampf(roll, (float)(5.0 * 3.141592653589793 / 9.0), PI_F);
        const float r3 = clampf(roll, -PI_F, (float)(-5.0 * 3.141592653589793 / 9.0));
        const bool inverted = !m1;
        const float rollc = ((m1 ? 1.0f : 0.0f) * r1 + (m2 ? 1.0f : 0.0f) * r2) + (m3 ? 1.0f : 0.0f) * r3;
        const bool mp = fabsf(pitch) <= (float)(7.0 * 3.141592653589793 / 18.0);
        float sr, cr, trl;
        np_sincostan(rollc, sr, cr, trl);
        float off = ((((1.0f / TAS) * cfg.gravity) * trl) * sr) * eas2tas;
        off = (((mp ? 1.0f : 0.0f) * tr.ct) * fabsf(off)) * cfg.roll_ff;
        off = off * (inverted ? 0.0f : 1.0f) - off * (inverted ? 1.0f : 0.0f);
        float d1 = desired + off;
        if (cfg.pitch.rmax_pos != 0.0f) d1 = d1 > cfg.pitch.rmax_pos ? cfg.pitch.rmax_pos : d1;
        if (cfg.pitch.rmax_neg != 0.0f) d1 = d1 < -cfg.pitch.rmax_neg ? -cfg.pitch.rmax_neg : d1;
        desired = (inverted ? 0.0f : 1.0f) * d1 + (inverted ? 1.0f : 0.0f) * (off - desired);
        float rw = fabsf(roll);
        const float pw = fabsf(pitch);
        const bool mk = rw > pio2;
        rw = (mk ? 1.0f : 0.0f) * (PI_F - rw) + (mk ? 0.0f : 1.0f) * rw;
        const bool mq = (rw > (float)(5.0 * 3.141592653589793 / 18.0)) & (pw < (float)(7.0 * 3.141592653589793 / 18.0));
        float prop = (rw - (float)(5.0 * 3.141592653589793 / 18.0)) / (float)(4.0 * 3.141592653589793 / 18.0);
        prop = prop * (mq ? 1.0f : 0.0f);
        desired = desired * (1.0f - prop);
        el = rate_out(cfg.pitch, cfg.dt_pid, desired, scaler, eas2tas, pitch_rate, pid[PID_P_ERR], pid[PID_P_INT], pid[PID_P_LAST],
                      true, first);
    }
    // stabilize_yaw :54-57: YawController.get_rate_out(yaw_rate_dem == 0)
    rud = rate_out(cfg.yaw, cfg.dt_pid, 0.0f, scaler, eas2tas, yaw_rate, pid[PID_Y_ERR], pid[PID_Y_INT], pid[PID_Y_LAST], false, first);
}

// ground velocity xdot[0..2] (F16_dynamics.py:129-135) — the `es[:, :3]` of the pairwise geometry
__device__ __forceinline__ void ground_velocity(const float (&s)[12], const Trig &tr, float spsi, float cpsi, float (&v)[3]) {
    float vt = s[6];
    vt = (vt <= 0.01f ? 1.0f : 0.0f) * 0.01f + (vt > 0.01f ? 1.0f : 0.0f) * vt;
    const float U = (vt * tr.ca) * tr.cb, V = vt * tr.sb, W = (vt * tr.sa) * tr.cb;
    const float st = tr.st, ct = tr.ct, sphi = tr.sphi, cphi = tr.cphi;
    v[0] = (U * (ct * cpsi) + V * ((sphi * cpsi) * st - cphi * spsi)) + W * ((cphi * st) * cpsi + sphi * spsi);
    v[1] = (U * (ct * spsi) + V * ((sphi * spsi) * st + cphi * cpsi)) + W * ((cphi * st) * spsi - sphi * cpsi);
    v[2] = (U * st - V * (sphi * ct)) - W * (cphi * ct);
}

// envs/utils/utils.py:156-205 get_AO_TA_R / get2d_AO_TA_R from the EGO aircraft's point of view
template <int DIMS>
__device__ __forceinline__ void ao_ta_r(const float (&ep)[3], const float (&mp)[3], const float (&ev)[3], const float (&mv)[3],
                                        float &AO, float &TA, float &R, float &side) {
    const float e2 = DIMS == 3 ? ev[2] : 0.0f, m2 = DIMS == 3 ? mv[2] : 0.0f;
    const float d0 = mp[0] - ep[0], d1 = mp[1] - ep[1], d2 = DIMS == 3 ? mp[2] - ep[2] : 0.0f;
    const float ego_v = np_norm3(ev[0], ev[1], e2), enm_v = np_norm3(mv[0], mv[1], m2);
    const float dist = np_norm3(d0, d1, d2);
    float proj = np_dot3(d0, d1, d2, ev[0], ev[1], e2);
    AO = np_acos(clampf(proj / (dist * ego_v + 1e-8f), -1.0f, 1.0f));
    proj = np_dot3(d0, d1, d2, mv[0], mv[1], m2);
    TA = np_acos(clampf(proj / (dist * enm_v + 1e-8f), -1.0f, 1.0f));
    R = dist;
    const float c = ev[0] * d1 - ev[1] * d0;
    side = (c != c) ? c : (float)((c > 0.0f) - (c < 0.0f));
}

__device__ __forceinline__ float orientation_reward_v2(float AO, float TA) {  // utils.py:207-218
    const float PI_F = 3.14159265358979323846f;
    const float a = (1.0f / (NP_DIVC(AO * 50.0f, PI_F) + 2.0f)) * 1.0f + 0.5f;
    float m = NP_DIVC(TA * 1.9f, PI_F);
    const float floor_ = 1e-4f * 1.0f;
    m = (m != m) ? m : (m > floor_ ? m : floor_);
    float t = NP_DIVC(np_atanh(1.0f - m), (float)(2.0 * 3.141592653589793));
    t = (t != t) ? t : (t < 0.0f ? t : 0.0f);
    return (a + t) + 0.5f;
}
__device__ __forceinline__ float range_reward_v3(float R) {  // utils.py:220-233, R in km
    const float near_ = R < 5.0f ? 1.0f : 0.0f;
    const float poly = clampf((-0.032f * (R * R) + 0.284f * R) + 0.38f, 0.0f, 1.0f);
    const float tail = clampf(np_exp(-0.16f * R), 0.0f, 0.2f);
    return (near_ + (R >= 5.0f ? 1.0f : 0.0f) * poly) + tail;
}
__device__ __forceinline__ float orientation_fn(float AO) {  // utils.py:235-243
    const float PI_F = 3.14159265358979323846f;
    const float pi6 = (float)(3.141592653589793 / 6.0);
    const float m3 = ((AO >= 0.0f) & (AO <= pi6)) ? 1.0f : 0.0f, m4 = ((AO <= 0.0f) & (AO >= -pi6)) ? 1.0f : 0.0f;
    const float q = NP_DIVC(6.0f * AO, PI_F);
    return (1.0f - q) * m3 + (1.0f + q) * m4;
}
__device__ __forceinline__ float distance_fn(float R) {  // utils.py:245-249
    const float m1 = R <= 1.0f ? 1.0f : 0.0f, m2 = ((R > 1.0f) & (R <= 3.0f)) ? 1.0f : 0.0f;
    return m1 + ((3.0f - R) / 2.0f) * m2;
}

#ifndef NPF16_COMBAT_MINWAVES
#define NPF16_COMBAT_MINWAVES 2  // waves per SIMD the register allocator must leave room for (256 VGPRs at 2)
#endif
constexpr int COMBAT_OBS = 15;
constexpr int COMBAT_BLOCK = 128;
constexpr int COMBAT_DUAL_TILE = 128;  // aircraft per workgroup of the dual8 / dual4 variants (np_combat_lat.hip)
static_assert(NUM_LIVE_NETS >= COMBAT_OBS, "the coefficient columns double as the observation transpose tile");

// STEP=true: SingleCombatEnv.step; STEP=false: reset_done_envs + obs
// TILE, WPT: as f16_env_kernel — (128, 1) throughput variant; (128, 2) pair variant: the two waves of the workgroup split the nets
// of every evaluation and evaluate their half for both waves' aircraft (dual asm bodies, half the scalar weight traffic);
// (64, 4) latency variant: four waves hold the same 64 aircraft (32 engagements), split the net evaluations and repeat the
// rest; (128, WPT_DUAL8) dual8 variant (round 4, np_combat_lat.hip; up to one tile per CU): eight waves per tile of 128 aircraft — waves
// 0..3 hold rows 0..63, waves 4..7 rows 64..127 — and wave w evaluates slice w of the eight-wave plans for both halves with the two-set
// bodies.  The engagement's pair exchange stays inside each wave in every variant.
// PW: waves per SIMD the pair variant is built for (3: 168 VGPRs with ~46 dwords per lane in scratch — only worth it where six
// workgroups per CU hold a whole grid in ONE generation, see launch_combat)
template <int SOLVER, bool STEP, int TILE = COMBAT_BLOCK, int WPT = 1, int PW = NPF16_COMBAT_MINWAVES>
__global__ __launch_bounds__(WPT == 4 ? TILE * 4 : dual_waves(WPT) ? 64 * dual_waves(WPT) : TILE, PW) void f16_combat_kernel(const CombatArgs a) {
    constexpr int DW = dual_waves(WPT);  // dual family: waves per 128-aircraft tile (the first DW / 2 hold rows 0..63, the others rows 64..127)
    static_assert(DW == 0 || TILE == 128, "the dual family shares tiles of 128 aircraft");
    constexpr int B = TILE;
    constexpr bool SPLIT = WPT == 4 || DW != 0;  // several waves hold the same rows and split the nets
    constexpr int THREADS = WPT == 4 ? TILE * 4 : DW ? 64 * DW : TILE;
    constexpr int COLS = NUM_LDS_SLOTS + ((WPT == 2 || DW) ? NUM_NORM_GROUPS : 0);  // two-set bodies: nine more columns carry the inputs to the other half
    __shared__ float lds[COLS * TILE];  // > TILE * COMBAT_OBS
    const int t = WPT == 4 ? (int)(threadIdx.x % TILE) : DW ? (int)((threadIdx.x & 63) + 64 * ((int)threadIdx.x / (32 * (DW ? DW : 1)))) : (int)threadIdx.x;
    // latency variant: which quarter of the nets / dual family: which eighth or quarter / pair variant: which wave of the pair (wave-uniform);
    // stores are done by one wave per set of rows in the latency and dual variants and by every wave otherwise
    const int part = WPT == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x / 64)) % (WPT == 4 ? 4 : DW ? DW : 2);
    const bool storer = WPT == 4 ? part == 0 : DW ? part % ((DW ? DW : 2) / 2) == 0 : true;
    float *coef = lds + t;
    const long long i0 = (long long)blockIdx.x * B;
    const long long i = i0 + t;
    const bool valid = i < a.n;
    const long long ic = valid ? i : a.n - 1;  // n is even and B is even: a pair never straddles workgroups
    // scalars are not kept live across the asm phases (which own s2-s101): every phase re-reads what it needs from the
    // kernel-argument segment (NP_REREAD_ARGS, see f16_env_kernel)
    CombatArgsC ap = (CombatArgsC)__builtin_amdgcn_kernarg_segment_ptr();  // `a` is the only kernel parameter
    const CombatDevCfg &cfg = a.cfg;
    const bool tables = cfg.aero_1d_tables != 0;
    const bool is_ego = (t & 1) == 0;
    const float PI_F = 3.14159265358979323846f;

    float s[12], u[4], pid[NUM_PID];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = a.s[k * a.ld + ic];
#pragma unroll
    for (int k = 0; k < 4; k++) u[k] = a.u[k * a.ld + ic];
    float blood = a.blood[ic];
    long long sc = a.step_count[ic];
    const int own_flag = (a.fin0[ic] | a.fin1[ic] | a.fin2[ic]) != 0 ? 1 : 0;
    const bool flagged = (own_flag | __shfl_xor(own_flag, 1)) != 0;

    // ---- reset_done_envs (singlecombat_env.py:207-238): both aircraft of a flagged env ----
    if (flagged) {
        float ru[5];
        if (a.rand_u) {
#pragma unroll
            for (int k = 0; k < 5; k++) ru[k] = a.rand_u[ic * 5 + k];
        } else {
            uint32_t w0[4], w1[4];
            rng_block(a.seed, a.call_idx, a.row0 + ic, 0, w0);
            rng_block(a.seed, a.call_idx, a.row0 + ic, 1, w1);
#pragma unroll
            for (int k = 0; k < 4; k++) ru[k] = (float)(w0[k] >> 8) * 5.9604644775390625e-08f;
            ru[4] = (float)(w1[0] >> 8) * 5.9604644775390625e-08f;
        }
#pragma unroll
        for (int k = 0; k < 12; k++) s[k] = 0.0f;
        u[1] = u[2] = u[3] = 0.0f;
        s[0] = ru[0] * cfg.npos_span + cfg.min_npos;
        s[1] = ru[1] * cfg.epos_span + cfg.min_epos;
        s[2] = ru[2] * cfg.alt_span + cfg.min_altitude;
        s[5] = ru[3] * cfg.yaw_span + cfg.min_heading;
        s[6] = ru[4] * cfg.vt_span + cfg.min_vt;
        u[0] = cfg.init_T;
        blood = 100.0f;
        sc = 0;
    }

    Trig tr;
    float tt, spsi, cpsi;
    trig_of(s, tr, tt);
    np_sincos(s[5], spsi, cpsi);

    bool f_done = false, f_bad = false, f_to = false;
    if (STEP) {
#pragma unroll
        for (int k = 0; k < NUM_PID; k++) pid[k] = a.pid[k * a.ld + ic];
        float act[3];
        {
            // interleaved rows [n][stride], or the split layout: ego / opponent rows of env ic / 2 from their own arrays
            const float *arow = a.action_opp ? ((ic & 1) ? a.action_opp : a.action) + (ic >> 1) * a.act_stride : a.action + ic * a.act_stride;
#pragma unroll
            for (int k = 0; k < 3; k++) act[k] = clampf(arow[k], -1.0f, 1.0f);
        }
        // shutdown.py:31-38 reads the blood of the previous env.step for all inner steps
        const float blood_o = partner(blood);
        const bool m1 = (is_ego ? blood : blood_o) <= 0.0f, m2 = (is_ego ? blood_o : blood) <= 0.0f;
        // the 14 force-side alpha/beta-only coefficients at the current state -> this lane's LDS column; from
        // here on every integrator evaluation finds them there (left by the previous Overload evaluation)
        {
            const float r2d = (float)(180.0 / 3.141592653589793);
            float xn[NUM_NORM_GROUPS];
            normalise_inputs(a.wt, s[7] * r2d, s[8] * r2d, u[1], xn);
            // pair variant: the Overload phase (the 14 nets + the force-side Cx, Cz, whose values are simply not used here)
            // (the latency family too: its waves split these nets like those of every later evaluation — until round 4 each of the four
            // waves evaluated all fourteen)
            if constexpr (WPT == 2 || SPLIT) eval_nets<B, AB_FORCE, false, WPT>(a.wt, xn, coef, tables, part);
            else eval_ab<B, AB_FORCE>(a.wt, xn, coef, tables);
        }
        NP_REREAD_ARGS(ap);
#pragma nounroll
        for (int it = 0; it < ap->cfg.inner_steps; it++) {
            // ---- demand filters (:245-246) and Controller.stabilize ----
            pid[PID_ROLL_DEM] = 0.9f * pid[PID_ROLL_DEM] + NP_DIVC(((0.1f * act[1]) * 4.0f) * PI_F, 9.0f);
            pid[PID_PITCH_DEM] = 0.9f * pid[PID_PITCH_DEM] + NP_DIVC((0.1f * act[2]) * PI_F, 12.0f);
            float el, ail, rud;
            stabilize(ap->cfg, s, tr, tt, pid, ap->pid_first != 0 && it == 0, el, ail, rud);
            u[0] = ap->cfg.af.lag_keep * u[0] + np_divc(((ap->cfg.af.lag_new * act[0]) * ap->cfg.af.thrust_frac) * ap->cfg.af.thrust_max, ap->cfg.af.thrust_unit, ap->cfg.af.r_thrust_unit);  // :251
            u[1] = -el;                                                                 // :252-255, written straight to u
            u[2] = -ail;
            u[3] = -rud;
            // ---- one integrator step (F16_model.py:64-67) ----
            const AeroWeights wt1 = {ap->wt.kblob, ap->wt.kblob_dual, ap->wt.pwl, ap->wt.pwl_unnorm};
            const bool tables1 = ap->cfg.aero_1d_tables != 0;
            if (SOLVER == 0) {
                float k1[12];
                nlplant<true, AB_REST, B, WPT>(wt1, airframe_via(ap), s, u, tr, tt, spsi, cpsi, coef, tables1, k1, part);
                NP_REREAD_ARGS(ap);
                const float dt = ap->cfg.dt;
#pragma unroll
                for (int k = 0; k < 12; k++) s[k] = s[k] + dt * k1[k];
            } else {  // torchdiffeq 0.2.3 rk4_alt_step_func (3/8 rule)
                const float dt = cfg.dt;
                const float third = (float)(1.0 / 3.0);
                float y[12], k1[12], k2[12], k3[12];
#pragma unroll
                for (int k = 0; k < 12; k++) y[k] = s[k];
#pragma nounroll
                for (int stage = 0; stage < 4; stage++) {
                    float kk[12];
                    if (stage == 0) nlplant<true, AB_REST, B, WPT>(a.wt, airframe_via(ap), y, u, tr, tt, spsi, cpsi, coef, tables, kk, part);
                    else xdot_full<AB_ALL, B, WPT>(a.wt, airframe_via(ap), y, u, coef, tables, kk, part);
                    if (stage == 0) {
#pragma unroll
                        for (int k = 0; k < 12; k++) {
                            k1[k] = kk[k];
                            y[k] = s[k] + (dt * k1[k]) * third;
                        }
                    } else if (stage == 1) {
#pragma unroll
                        for (int k = 0; k < 12; k++) {
                            k2[k] = kk[k];
                            y[k] = s[k] + dt * (k2[k] - k1[k] * third);
                        }
                    } else if (stage == 2) {
#pragma unroll
                        for (int k = 0; k < 12; k++) {
                            k3[k] = kk[k];
                            y[k] = s[k] + dt * ((k1[k] - k2[k]) + k3[k]);
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 12; k++) y[k] = s[k] + (((k1[k] + 3.0f * (k2[k] + k3[k])) + kk[k]) * dt) * 0.125f;
                    }
                }
#pragma unroll
                for (int k = 0; k < 12; k++) s[k] = y[k];
            }
            sc += 1;
            // ---- terminations at the new state ----
            trig_of(s, tr, tt);
            np_sincos(s[5], spsi, cpsi);
            if constexpr (WPT == 2) {
                // as in f16_env_kernel: without these the compiler sinks the moment equations of the integrator evaluation and the
                // tails of the fp64 sine / cosine sequences past the Overload statement (which owns v70-v157) and keeps their
                // operands live across it
#pragma unroll
                for (int k = 0; k < 12; k++) asm volatile("" : "+v"(s[k]));
                asm volatile("" : "+v"(tr.sa), "+v"(tr.ca), "+v"(tr.sb), "+v"(tr.cb), "+v"(tr.st), "+v"(tr.ct), "+v"(tr.sphi), "+v"(tr.cphi));
                asm volatile("" : "+v"(tt), "+v"(spsi), "+v"(cpsi));
            }
            float xd[12], acc3[3];
            {
                const AeroWeights wt2 = {ap->wt.kblob, ap->wt.kblob_dual, ap->wt.pwl, ap->wt.pwl_unnorm};
                nlplant<false, AB_FORCE, B, WPT>(wt2, airframe_via(ap), s, u, tr, 0.0f, 0.0f, 0.0f, coef, ap->cfg.aero_1d_tables != 0, xd, part);
            }
            NP_REREAD_ARGS(ap);
            body_acceleration(s, tr, xd, acc3);
            const float acc = sqrtf((acc3[0] * acc3[0] + acc3[1] * acc3[1]) + acc3[2] * acc3[2]);
            const bool r_over = (acc - ap->cfg.acceleration_limit) > 0.0f;   // overload.py:37-42
            const bool r_low = (s[2] - ap->cfg.altitude_limit) < 0.0f;        // low_altitude.py:29-30
            const float TAS = s[6] + ap->cfg.airspeed * 1.0f;
            const float vel = NP_DIVC(TAS * 0.3048f, 340.0f);
            const bool r_fast = (vel - ap->cfg.max_velocity) >= 0.0f;         // high_speed.py:29-30
            const bool r_slow = (vel - ap->cfg.min_velocity) <= 0.0f;         // low_speed.py:29-30
            const float alpha = NP_DIVC(s[7] * 180.0f, PI_F), beta = NP_DIVC(s[8] * 180.0f, PI_F);
            const bool r_ext = ((alpha < ap->cfg.min_alpha) | (alpha > ap->cfg.max_alpha)) | ((beta < ap->cfg.min_beta) | (beta > ap->cfg.max_beta));  // extreme_state.py:32-36
            bool b = (((r_over | r_low) | r_fast) | r_slow) | r_ext;
            // crash.py:33-43 (ego - enemy, squared distance in fp32)
            const float on = partner(s[0]), oe = partner(s[1]), oa = partner(s[2]);
            const float dn = is_ego ? s[0] - on : on - s[0], de = is_ego ? s[1] - oe : oe - s[1], da = is_ego ? s[2] - oa : oa - s[2];
            const bool r_crash = ((dn * dn + de * de) + da * da) <= ap->cfg.dist_limit_sq;
            const bool r_tmo = (sc - ap->cfg.max_steps) >= 0;           // timeout.py:29
            b |= r_crash;
            b |= m1;                                               // shutdown.py:36-38
            f_bad |= b;
            f_done |= m2 & !m1;
            f_to |= r_tmo;
            if (ap->term_counters) {  // per-condition sums (the reference prints them per evaluation): ballot + popcount + 1 atomic
                const unsigned bits = (r_over ? 1u : 0u) | (r_low ? 2u : 0u) | (r_fast ? 4u : 0u) | (r_slow ? 8u : 0u) | (r_ext ? 16u : 0u) |
                                      (r_crash ? 32u : 0u) | (r_tmo ? 64u : 0u) | (m1 ? 128u : 0u) | ((m2 & !m1) ? 256u : 0u);
                const bool counted = valid && storer;
#pragma unroll
                for (int k = 0; k < NP_NUM_COMBAT_TERM_COUNTERS; k++) {
                    const unsigned long long mk = __ballot(counted && ((bits >> k) & 1u));
                    if (mk != 0 && (threadIdx.x & 63) == 0) atomicAdd(ap->term_counters + k, (unsigned)__popcll(mk));
                }
            }
        }
    }

    // ---- observation (:64-138), reward (:140-181) and blood (:264-271) at the final state ----
    float gv[3];
    ground_velocity(s, tr, spsi, cpsi, gv);
    const float vel_u = (s[6] * tr.cb) * tr.ca, vel_v = s[6] * tr.sb, vel_w = (s[6] * tr.cb) * tr.sa;  // get_velocity
    float op[3], ogv[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        op[k] = partner(s[k]);
        ogv[k] = partner(gv[k]);
    }
    const float o_vel_u = partner(vel_u);
    float ep[3], mp[3], ev[3], mv[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        ep[k] = is_ego ? s[k] : op[k];
        mp[k] = is_ego ? op[k] : s[k];
        ev[k] = is_ego ? gv[k] : ogv[k];
        mv[k] = is_ego ? ogv[k] : gv[k];
    }
    float AO2, TA2, R2, side;
    ao_ta_r<2>(ep, mp, ev, mv, AO2, TA2, R2, side);
    float o[COMBAT_OBS];
    o[0] = NP_DIVC(s[2] * 0.3048f, 5000.0f);
    o[1] = tr.sphi;
    o[2] = tr.cphi;
    o[3] = tr.st;
    o[4] = tr.ct;
    o[5] = NP_DIVC(vel_u * 0.3048f, 340.0f);
    o[6] = NP_DIVC(vel_v * 0.3048f, 340.0f);
    o[7] = NP_DIVC(vel_w * 0.3048f, 340.0f);
    o[8] = NP_DIVC(s[6] * 0.3048f, 340.0f);
    o[9] = NP_DIVC((o_vel_u - vel_u) * 0.3048f, 340.0f);
    o[10] = NP_DIVC((op[2] - s[2]) * 0.3048f, 1000.0f);
    o[11] = is_ego ? AO2 : PI_F - TA2;
    o[12] = is_ego ? TA2 : PI_F - AO2;
    o[13] = NP_DIVC(R2 * 0.3048f, 10000.0f);
    o[14] = is_ego ? side : -side;

    float reward = 0.0f;
    if (STEP) {
        float AO, TA, R, side3;
        ao_ta_r<3>(ep, mp, ev, mv, AO, TA, R, side3);
        const float Rkm = NP_DIVC(R * 0.3048f, 1000.0f);
        const float rr = range_reward_v3(Rkm);
        const float orient = is_ego ? orientation_reward_v2(AO, TA) : orientation_reward_v2(PI_F - TA, PI_F - AO);
        reward = 0.01f * (orient * rr);
        const float dfn = distance_fn(Rkm);
        // blood[enm] -= orientation_fn(AO) * dfn; blood[ego] -= orientation_fn(pi - TA) * dfn
        blood = blood - (is_ego ? orientation_fn(PI_F - TA) : orientation_fn(AO)) * dfn;
    }

    if (valid && storer) {
        long long iw = i;
        asm volatile("" : "+v"(iw));  // re-derive the store addresses here instead of keeping the load addresses alive
#pragma unroll
        for (int k = 0; k < 12; k++) ap->s[k * ap->ld + iw] = s[k];
#pragma unroll
        for (int k = 0; k < 4; k++) ap->u[k * ap->ld + iw] = u[k];
        ap->u[4 * ap->ld + iw] = 0.0f;
        ap->blood[iw] = blood;
        ap->step_count[iw] = sc;
        ap->fout0[iw] = f_done ? 1 : 0;
        ap->fout1[iw] = f_bad ? 1 : 0;
        ap->fout2[iw] = f_to ? 1 : 0;
        if (STEP) {
#pragma unroll
            for (int k = 0; k < NUM_PID; k++) ap->pid[k * ap->ld + iw] = pid[k];
            ap->reward[iw] = reward;
        }
    }

    // ---- [n][15] observation rows: transpose through LDS, store coalesced ----
    if (ap->obs) {
        __syncthreads();
        const bool split = ap->obs_opp != nullptr;  // workgroup-uniform
        // split layout: the tile holds its TILE / 2 ego rows first, then its TILE / 2 opponent rows — two contiguous blocks
        const int trow = split ? (t >> 1) + (t & 1) * (TILE / 2) : t;
        if (storer) {
#pragma unroll
            for (int k = 0; k < COMBAT_OBS; k++) lds[trow * COMBAT_OBS + k] = o[k];  // pitch 15 is odd: conflict-free
        }
        __syncthreads();
        const long long rows = (ap->n - i0) < B ? (ap->n - i0) : B;
        if (!split) {
            const int total = (int)rows * COMBAT_OBS;
            float *dst = ap->obs + i0 * COMBAT_OBS;
#pragma unroll
            for (int itr = 0; itr < (COMBAT_OBS * TILE + THREADS - 1) / THREADS; itr++) {
                const int L = itr * THREADS + (int)threadIdx.x;
                if (L < total) dst[L] = lds[L];
            }
        } else {
            constexpr int HALF = COMBAT_OBS * (TILE / 2);
            const int half_total = (int)(rows / 2) * COMBAT_OBS;  // n is even: a tile holds whole engagements
            float *dst_e = ap->obs + (i0 / 2) * COMBAT_OBS, *dst_o = ap->obs_opp + (i0 / 2) * COMBAT_OBS;
#pragma unroll
            for (int itr = 0; itr < (COMBAT_OBS * TILE + THREADS - 1) / THREADS; itr++) {
                const int L = itr * THREADS + (int)threadIdx.x;
                if (L < HALF) {
                    if (L < half_total) dst_e[L] = lds[L];
                } else if (L < 2 * HALF) {
                    if (L - HALF < half_total) dst_o[L - HALF] = lds[L];
                }
            }
        }
    }
}

// np_combat_lat.hip: one launch of the dual8 / dual4 variant (`waves` = 8 / 4; Euler step, MLP numerics); `start` / `stop` are attached to the dispatch when `timed`
void launch_combat_dual(const CombatArgs &a, int waves, unsigned grid, hipStream_t st, bool timed, hipEvent_t start, hipEvent_t stop);

}  // namespace npf16
