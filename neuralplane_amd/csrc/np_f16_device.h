// np_f16_device.h — per-aircraft device functions of the fused F-16 env.step kernel (gfx950).
//
// One lane = one aircraft.  All MLP weights are wave-uniform and live in __constant__ memory in
// the order the FMA chains consume them, so the compiler feeds them through s_load_dwordxN and
// the FMAs are `v_fmac_f32 v_acc, s_weight, v_x` — no VGPR and no LDS spent on weights.
//
// Arithmetic follows the reference operator by operator (see the citations) and
// the numerics spec of DESIGN.md for the implementation-defined pieces.  Build with
// -ffp-contract=off: the only fused operations are the explicit fmaf() chains of the Linear layers.
#pragma once
#include <hip/hip_runtime.h>

#include "np_math.h"
#include "np_nets.h"

namespace npf16 {

// The packed weights of one aircraft type (np_nets.h KBLOB layout + the optional PWL tables) live in device buffers owned
// by a context, so contexts with different weight sets (a second aircraft with the same net topology) coexist on a device.
// device code reads them through the CONSTANT address space (read-only for the lifetime of a launch): invariant loads the
// compiler may reorder, scalarise and merge exactly as it did when the data were __constant__ symbols
typedef const float __attribute__((address_space(4))) *cf32_ptr;
#define NPF16_CONST(p) ((cf32_ptr)(unsigned long long)(p))
struct AeroWeights {
    const float *kblob;       // [KBLOB_FLOATS]
    const float *kblob_dual;  // [KBLOB_DUAL_FLOATS] the same nets in the record layout of the two-set bodies (np_nets.h)
    const float *pwl;         // [NUM_PWL_TABLES * PWL_TABLE_FLOATS] or null
    const float *pwl_unnorm;  // [NUM_PWL_TABLES * 2] or null
};

// The airframe as data (np_f16_airframe, ABI 16): what F16Dynamics.nlplant / atmos and F16Model.update spell as literals
// (envs/models/F16/F16_dynamics.py:22-35,61-76,114-116; envs/models/F16_model.py:52-62), rounded to fp32 on the host exactly where the
// literal expressions round (derived constants folded in double first: np_f16_kernels.hip::make_airframe).  r_* = RN(1 / x) of a divisor
// (np_divc).  The defaults are the F-16 literals: a context built without an airframe block computes what the literals computed, bit for bit.
struct Airframe {
    float g, mass, r_mass, B, S, cbar, Heng, pad_;
    float Jy, r_Jy, Jxz, Jz, Jx;
    float xc, cbar_over_B, c1, c2, c3, c4, denom, r_denom;       // xcgr - xcg, cbar / B, the inertia products of the moment equations
    float ail_ref, r_ail_ref, rud_ref, r_rud_ref;                 // dail = ail / 21.5, drud = rud / 30
    float atm_lapse, rho0;                                        // tfac = 1 - 0.703e-5 alt; rho = 2.377e-3 tfac^4.14
    double atm_exp;                                               // (double)(float)4.14: the double np_pow computes in; > 0 (host check)
    float lag_keep, lag_new, thrust_frac, thrust_max, thrust_unit, r_thrust_unit, surf_max[3];   // u' = 0.9 u + 0.1 a * scale
    __device__ __forceinline__ const Airframe &get() const { return *this; }
};
// How nlplant reaches the airframe.  Inside the step kernels nothing scalar may stay live across the two MLP phases (asm statements that own
// s4-s101), so the constants are (re-)read from the kernel-argument segment AFTER the phase, through the pointer the kernel already keeps
// for that purpose (`ap`, NP_REREAD_ARGS): ~35 scalar loads per evaluation, no VALU work, no register held across the statement.
// AirframeVia<AP>{ap}.get() = ap->cfg.af behind such a re-read; a plain `Airframe` (kernels off the hot path) is its own get().
#ifndef NP_REREAD_ARGS
#define NP_REREAD_ARGS(ap) asm volatile("" : "+s"(ap) : : "memory")
#endif
// (np_cfg_of: where an argument record keeps its DevCfg / CombatDevCfg — `ap->cfg` unless the record's own header says otherwise)
template <class AP>
__device__ __forceinline__ auto np_cfg_of(AP ap) -> decltype(&ap->cfg) { return &ap->cfg; }
template <class AP>
struct AirframeVia {
    AP &ap;
    __device__ __forceinline__ auto &get() const {
        NP_REREAD_ARGS(ap);
        return np_cfg_of(ap)->af;
    }
};
template <class AP>
__device__ __forceinline__ AirframeVia<AP> airframe_via(AP &ap) { return AirframeVia<AP>{ap}; }

// Scenario constants, pre-rounded on the host exactly where the reference rounds them.
struct DevCfg {
    float dt, airspeed, noise_scale;
    float altitude_limit, acceleration_limit, max_velocity, min_velocity;
    float min_alpha, max_alpha, min_beta, max_beta;
    long long max_check_interval, min_check_interval;
    float init_T, alt_span, min_altitude, vt_span, min_vt;
    float max_heading_increment, max_pitch_increment, max_velocities_u_increment;
    float dist_span, min_distance;
    int aero_1d_tables;  // single-input nets through their piecewise-linear tables instead of the MLP bodies
    Airframe af;
};

// ---------------------------------------------------------------------------------------------
// MLP evaluation — hifi_F16_AeroData.py:12-37 (MLP.forward, normalize, unnormalize)
//   Linear: acc = bias; acc = fmaf(W[j][k], x[k], acc), k ascending (numerics spec)
// ---------------------------------------------------------------------------------------------
// Weight stream of one net: the KBLOB record is consumed strictly front to back, one weight per
// FMA.  Weights are wave-uniform, so they travel through the scalar unit: a CHUNK of 16 weights is
// one `s_load_dwordx16` into 16 SGPRs and each FMA names its weight as an SGPR operand
// (`v_fmac_f32 v_acc, s_w, v_x` / `v_pk_fma_f32`).  Two chunks ahead are kept in flight.  Left to
// itself hipcc hoists dozens of these loads, overflows the 102 SGPRs and spills them lane-by-lane
// into VGPRs (thousands of v_readlane/v_writelane); the empty asm "pins" below tie the stream
// pointer to each consumed chunk so that at most NBUF chunks are ever live.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef f32x16 f32x16_u __attribute__((aligned(4)));

template <int NBUF, int LEN>  // LEN = floats in the record
struct WStream {
    const float *kb;  // KBLOB base (wave-uniform)
    int off;  // KBLOB offset of the record, wave-uniform
    f32x16 buf[NBUF];
    int p;  // position in the record; a compile-time constant after full unrolling

    __device__ __forceinline__ void start(const float *kblob, int base) {
        kb = kblob;
        off = base;
        p = 0;
#pragma unroll
        for (int b = 0; b < NBUF; b++)
            if (16 * b < LEN) buf[b] = *(const f32x16_u *)(kb + off + 16 * b);
    }
    __device__ __forceinline__ float next() {
        const int c = p / 16, l = p % 16;
        if (l == 0) {
            // chunk c is about to be consumed: it must have landed, and nothing issued after this
            // point may be scheduled above it (the later loads depend on the pinned offset).
            // Two statements, 32-bit offset: hipcc treats a multi-output asm, and a 64-bit pointer
            // that went through asm, as divergent values (-> vector loads, "illegal VGPR to SGPR copy").
            asm volatile("" : "+s"(buf[c % NBUF]));
            asm volatile("" : "+s"(off));
            if (c > 0 && 16 * (c - 1 + NBUF) < LEN)
                buf[(c - 1) % NBUF] = *(const f32x16_u *)(kb + off + 16 * (c - 1 + NBUF));
        }
        p++;
        return buf[c % NBUF][l];
    }
};

// One Linear(+ReLU) layer fed from the weight stream: bias[out] then W^T[in][out] (k-major).
template <int IN, int OUT, bool RELU, class WS>
__device__ __forceinline__ void dense(WS &ws, const float (&x)[IN], float (&y)[OUT]) {
#pragma unroll
    for (int j = 0; j < OUT; j++) y[j] = ws.next();
    if (OUT & 1) (void)ws.next();  // rows are padded to an even length (np_nets.h::asm_record_len)
#pragma unroll
    for (int k = 0; k < IN; k++) {
#pragma unroll
        for (int j = 0; j < OUT; j++) y[j] = fmaf(ws.next(), x[k], y[j]);
        if (OUT & 1) (void)ws.next();
    }
    if (RELU) {
#pragma unroll
        for (int j = 0; j < OUT; j++) {  // ReLU on activations carried / 2^ACT_SHIFT: saturates at 1 (= the asm bodies' clamp modifier)
            y[j] = y[j] > 0.0f ? y[j] : 0.0f;
            y[j] = y[j] > 1.0f ? 1.0f : y[j];
        }
    }
}

constexpr int WS_NBUF = 3;

template <int IN, int H1, int H2, int H3>
__device__ __forceinline__ float mlp_body(const float *kblob, int w, const float (&x)[IN]) {
    constexpr int LEN = asm_record_len(IN, H1, H2, H3);
    WStream<WS_NBUF, LEN> ws;
    ws.start(kblob, w);
    float h1[H1], h2[H2];
    dense<IN, H1, true>(ws, x, h1);
    dense<H1, H2, true>(ws, h1, h2);
    // output layer: two interleaved partial chains from (bias, 0), even inputs to the first, odd to the second; an odd last
    // input joins the first; y = lo + hi (numerics spec, DESIGN.md §4 — one packed FMA per input pair in the asm bodies)
    auto out_layer = [&ws](const float *h, int n) {
        float lo = ws.next(), hi = ws.next();
        for (int k = 0; k + 1 < n; k += 2) {
            lo = fmaf(ws.next(), h[k], lo);
            hi = fmaf(ws.next(), h[k + 1], hi);
        }
        if (n & 1) {
            lo = fmaf(ws.next(), h[n - 1], lo);
            (void)ws.next();
        }
        return lo + hi;
    };
    float y;
    if constexpr (H3 > 0) {
        float h3[H3];
        dense<H2, H3, true>(ws, h2, h3);
        y = out_layer(h3, H3);
    } else {
        y = out_layer(h2, H2);
    }
    const float out_std = ws.next();
    const float out_mean = ws.next();
    return y * out_std + out_mean;  // unnormalize: X * std + mean
}

#ifndef NPF16_ASM_MLP
#define NPF16_ASM_MLP 1  // 1: hand-scheduled asm class bodies (np_mlp_asm.inc).  0 selected the C++ bodies above (the readable statement of
                         // what the asm computes; A/B reference of the first builds) — since the weights moved from __constant__ to
                         // per-context buffers hipcc no longer selects scalar loads for them ("illegal VGPR to SGPR copy")
#endif
#include "np_mlp_asm.inc"
#include "np_mlp_asm_dual.inc"

// All nets of one class (np_nets.h): same shape, same inputs, KBLOB records back to back.  ONE
// compact loop body per class keeps the instruction footprint of a full aero evaluation at a few
// KB (it stays in the instruction cache) where straight-line code for 42 nets would be ~100 KB.
// Each net's output goes to this lane's column of the LDS scratch: out[slot*LD] (ds_write_b32,
// consecutive lanes -> consecutive banks), from where the coefficient build-up reads it back.
// Single-input nets as table lookups (spec option aero_1d_tables): per net a 6-step binary search over 63 sorted
// breakpoints, then one fma on the segment's line.  The searches of all nets of the class are interleaved so
// that their dependent gathers (per-lane addresses into __constant__ tables, served by the vector L1) overlap.
// a float of a context-owned table at (wave-uniform base) + (per-lane 32-bit byte offset): selects the SGPR-base + VGPR-offset
// addressing mode (one global_load_dword, no 64-bit address arithmetic per lane)
__device__ __forceinline__ float ld_table(const float *base, unsigned byte_off) {
    typedef const float __attribute__((address_space(1))) GT;
    return *reinterpret_cast<GT *>(reinterpret_cast<uintptr_t>(base) + byte_off);
}

template <int CL, int COUNT, int LD, int FIRST>
__device__ __forceinline__ void eval_class_pwl(const AeroWeights &wt, float x, float *__restrict__ out) {
    constexpr NetClass c = CLASSES[CL];
    const float *pwl = wt.pwl;
    const cf32_ptr pwl_unnorm = NPF16_CONST(wt.pwl_unnorm);
    unsigned idx4[COUNT > 0 ? COUNT : 1];  // segment index x 4 (a byte offset)
#pragma unroll
    for (int m = 0; m < COUNT; m++) idx4[m] = 0;
#pragma unroll
    for (int h = PWL_SEG / 2; h >= 1; h >>= 1) {
#pragma unroll
        for (int m = 0; m < COUNT; m++) {
            const unsigned tb4 = (unsigned)(pwl_index(c.nets[FIRST + m]) * PWL_TABLE_FLOATS) * 4u;
            idx4[m] += (x >= ld_table(pwl, tb4 + idx4[m] + 4u * (unsigned)(h - 1))) ? 4u * (unsigned)h : 0u;
        }
    }
#pragma unroll
    for (int m = 0; m < COUNT; m++) {
        const int ti = pwl_index(c.nets[FIRST + m]);
        const unsigned tb4 = (unsigned)(ti * PWL_TABLE_FLOATS) * 4u;
        const float yn = fmaf(ld_table(pwl, tb4 + 4u * PWL_SEG + idx4[m]), x - ld_table(pwl, tb4 + 8u * PWL_SEG + idx4[m]), ld_table(pwl, tb4 + 12u * PWL_SEG + idx4[m]));
        out[(class_slot(CL) + FIRST + m) * LD] = yn * pwl_unnorm[2 * ti] + pwl_unnorm[2 * ti + 1];
    }
}

template <int CL, int COUNT, int LD, int FIRST = 0>  // nets [FIRST, FIRST+COUNT) of class CL
__device__ __forceinline__ void eval_class(const AeroWeights &wt, const float (&xn)[NUM_NORM_GROUPS], float *__restrict__ out, bool tables) {
    constexpr NetClass c = CLASSES[CL];
    constexpr int n = COUNT;
    static_assert(COUNT >= 0 && FIRST >= 0 && FIRST + COUNT <= c.count, "class range");
    if constexpr (c.n_in == 1 && n > 0) {
        if (tables) {  // wave-uniform
            eval_class_pwl<CL, COUNT, LD, FIRST>(wt, xn[c.grp[0]], out);
            return;
        }
    }
#if defined(NPF16_EXP) && (NPF16_EXP & 4)  // timing experiment only: no MLP evaluation
    if constexpr (n > 0) {
#pragma unroll
        for (int i = 0; i < n; i++) out[(class_slot(CL) + FIRST + i) * LD] = xn[c.grp[0]] * 1e-3f;
    }
#elif NPF16_ASM_MLP
    if constexpr (n > 0) {
        // one asm statement for the whole class; `out` points into LDS: the low 32 bits of the flat
        // address are the LDS byte offset ds_write_b32 wants
        const unsigned lds_addr = (unsigned)(unsigned long long)(out + (class_slot(CL) + FIRST) * LD);
        const float x0 = xn[c.grp[0]];
        const float x1 = c.n_in > 1 ? xn[c.grp[c.n_in > 1 ? 1 : 0]] : 0.0f;
        const float x2 = c.n_in > 2 ? xn[c.grp[c.n_in > 2 ? 2 : 0]] : 0.0f;
        mlp_class_asm<c.n_in, c.h1, c.h2, c.h3, n, (int)(LD * sizeof(float))>(wt.kblob + class_base(CL) + FIRST * class_stride(CL), lds_addr, x0, x1, x2);
    }
#else
    if constexpr (n > 0) {
        float x[c.n_in];
#pragma unroll
        for (int i = 0; i < c.n_in; i++) x[i] = xn[c.grp[i]];
        int w = class_base(CL) + FIRST * class_stride(CL);
        float *__restrict__ o = out + (class_slot(CL) + FIRST) * LD;
#pragma nounroll
        for (int i = 0; i < n; i++) {
            *o = mlp_body<c.n_in, c.h1, c.h2, c.h3>(wt.kblob, w, x);
            w += class_stride(CL);
            o += LD;
        }
    }
#endif
}

// the 36 nets that depend on (alpha, beta) only -> slots 0..35.  Within every class the nets that feed
// xdot[6..8] (force side, 14 in total) come first.
// AB_EL / AB_ABALL (round 4, the persistent PlanningEnv kernel's pipelined schedule only): the integrator evaluation with ALL 36 alpha/beta-only
// coefficients already in the columns (only the six el-dependent nets are evaluated: they are the only ones that see the step's action) and
// the force-side evaluation extended to all 36 (what the NEXT integrator evaluation will find there) — the 22 moment-side nets move from
// the front of an inner step, which the controller's next call waits for, to its back, which runs beside that call.
enum AbPart : int { AB_ALL = 0, AB_FORCE = 1, AB_REST = 2, AB_EL = 3, AB_ABALL = 4, AB_GRU = 5 };
template <int LD, int PART>
__device__ __forceinline__ void eval_ab(const AeroWeights &wt, const float (&xn)[NUM_NORM_GROUPS], float *__restrict__ out, bool tables) {
#define NPF16_CLS(cl)                                                                                      \
    eval_class<cl, (PART == AB_ALL ? CLASSES[cl].count : PART == AB_FORCE ? CLASSES[cl].n_force : CLASSES[cl].count - CLASSES[cl].n_force), \
               LD, (PART == AB_REST ? CLASSES[cl].n_force : 0)>(wt, xn, out, tables)
    NPF16_CLS(CL_DAMP);
    NPF16_CLS(CL_DLEF);
    NPF16_CLS(CL_D_RUD);
    NPF16_CLS(CL_D_LEF);
    NPF16_CLS(CL_E_LEF);
    NPF16_CLS(CL_E_RUD);
    NPF16_CLS(CL_F);
    NPF16_CLS(CL_YPLEF);
    NPF16_CLS(CL_YA20);
#undef NPF16_CLS
}
// the el-dependent nets: the first N_C of (Cx Cz Cm Cn Cl) and eta_el -> slots 36..41
template <int N_C, int N_ETA, int LD>
__device__ __forceinline__ void eval_el(const AeroWeights &wt, const float (&xn)[NUM_NORM_GROUPS], float *__restrict__ out, bool tables) {
    eval_class<CL_C, N_C, LD>(wt, xn, out, tables);
    eval_class<CL_ETA, N_ETA, LD>(wt, xn, out, tables);
}

// ---- latency variant: the nets of one evaluation split over the 4 waves of a workgroup that share ONE tile of 64
// aircraft (small batches: a lone wave needs ~44 us per env.step, most of it the serial chain of 44 net evaluations).
// Every wave evaluates whole classes (or net ranges of a class) into the shared coefficient columns; plans are balanced
// by FLOPs (1-in 20-10: 460, 2-in 20-10: 500, 2-in 20-10-5: 610, 2-in 20-20-10: 1300, 3-in 20-10: 540 per net).
struct SplitItem {
    int cl, first, cnt;
};
// WPT codes of the latency family (several waves share ONE tile of 64 aircraft): 4 and 8 waves, and WPT_LAT2 = two waves per tile
// (mid-size batches, 32 K - 98 K aircraft: twice the waves of the pair variant for the same aircraft, so every SIMD holds two or
// three waves where the pair variant leaves it one; the two waves split the nets by the pair plans below, single-set bodies)
constexpr int WPT_LAT2 = 16;
// WPT_DUAL8 (SingleCombat, batches up to one workgroup per CU): eight waves per tile of 128 aircraft — waves 0..3 hold rows 0..63, waves
// 4..7 rows 64..127 — and wave w evaluates slice w of the EIGHT-wave plans for BOTH halves with the two-set class bodies (the other half's
// normalised inputs through LDS, as in the pair variant): an eighth of the weight stream per wave where four waves per 64 aircraft stream
// a quarter, on the same number of waves per aircraft
constexpr int WPT_DUAL8 = 32;
// WPT_DUAL4: the same with FOUR waves per tile of 128 aircraft (waves 0, 1 hold rows 0..63, waves 2, 3 rows 64..127; the four-wave plans):
// two tiles per CU at two waves per SIMD, so batches of up to 256 aircraft per CU stay in one generation
constexpr int WPT_DUAL4 = 33;
constexpr int dual_waves(int wpt) { return wpt == WPT_DUAL8 ? 8 : wpt == WPT_DUAL4 ? 4 : 0; }  // waves per 128-aircraft tile of the dual family
constexpr int lat_waves(int wpt) { return wpt == WPT_LAT2 ? 2 : dual_waves(wpt) ? dual_waves(wpt) / 2 : (wpt >= 4 ? wpt : 1); }
constexpr int SPLIT_WAVES = 8, SPLIT_MAX = 4;  // rows 4..7 stay empty in the four-wave plans
struct SplitPlan {
    SplitItem it[SPLIT_WAVES][SPLIT_MAX];
};
constexpr SplitItem NO_ITEM = {-1, 0, 0};
#define NPF16_NO_WAVE {NO_ITEM, NO_ITEM, NO_ITEM, NO_ITEM}
// AB_REST + C[0:5] + ETA (cached integrator evaluation, 28 nets)
constexpr SplitPlan PLAN_REST = {{{{CL_DAMP, 4, 8}, NO_ITEM, NO_ITEM, NO_ITEM},
                                  {{CL_DLEF, 2, 5}, {CL_D_RUD, 1, 1}, {CL_D_LEF, 1, 1}, {CL_ETA, 0, 1}},
                                  {{CL_E_LEF, 2, 2}, {CL_F, 1, 2}, NO_ITEM, NO_ITEM},
                                  {{CL_E_RUD, 1, 3}, {CL_C, 0, 5}, NO_ITEM, NO_ITEM},
                                  NPF16_NO_WAVE, NPF16_NO_WAVE, NPF16_NO_WAVE, NPF16_NO_WAVE}};
// AB_ALL + C[0:5] + ETA (un-cached evaluation, 42 nets)
constexpr SplitPlan PLAN_ALL = {{{{CL_DAMP, 0, 12}, {CL_ETA, 0, 1}, NO_ITEM, NO_ITEM},
                                 {{CL_DLEF, 0, 7}, {CL_C, 0, 5}, NO_ITEM, NO_ITEM},
                                 {{CL_F, 0, 3}, {CL_D_RUD, 0, 2}, {CL_D_LEF, 0, 2}, NO_ITEM},
                                 {{CL_E_LEF, 0, 4}, {CL_E_RUD, 0, 4}, {CL_YPLEF, 0, 1}, {CL_YA20, 0, 1}},
                                 NPF16_NO_WAVE, NPF16_NO_WAVE, NPF16_NO_WAVE, NPF16_NO_WAVE}};
// AB_FORCE + C[0:2] (Overload re-evaluation, 16 nets)
constexpr SplitPlan PLAN_FORCE2 = {{{{CL_DAMP, 0, 4}, {CL_D_RUD, 0, 1}, NO_ITEM, NO_ITEM},
                                    {{CL_DLEF, 0, 2}, {CL_E_LEF, 0, 2}, NO_ITEM, NO_ITEM},
                                    {{CL_F, 0, 1}, {CL_C, 0, 2}, NO_ITEM, NO_ITEM},
                                    {{CL_D_LEF, 0, 1}, {CL_E_RUD, 0, 1}, {CL_YPLEF, 0, 1}, {CL_YA20, 0, 1}},
                                    NPF16_NO_WAVE, NPF16_NO_WAVE, NPF16_NO_WAVE, NPF16_NO_WAVE}};
// ---- eight waves per tile (batches up to one workgroup per CU): two waves per SIMD, so one wave's scalar-load latency is the
// other's FMA time, and half as many nets on each wave's serial path.  Balanced by VALU instructions per net (1-in 20-10: 138,
// 2-in: 148, 3-in: 158, 2-in 20-10-5: 178, 2-in 20-20-10: 345, 1-in 20-10-5: 168, 2-in 20-10-10: 213).
constexpr SplitPlan PLAN8_REST = {{{{CL_DAMP, 4, 4}, NO_ITEM, NO_ITEM, NO_ITEM},
                                   {{CL_DAMP, 8, 4}, NO_ITEM, NO_ITEM, NO_ITEM},
                                   {{CL_DLEF, 2, 4}, NO_ITEM, NO_ITEM, NO_ITEM},
                                   {{CL_DLEF, 6, 1}, {CL_D_RUD, 1, 1}, {CL_D_LEF, 1, 1}, {CL_ETA, 0, 1}},
                                   {{CL_E_LEF, 2, 2}, {CL_E_RUD, 1, 1}, NO_ITEM, NO_ITEM},
                                   {{CL_E_RUD, 2, 2}, {CL_C, 0, 2}, NO_ITEM, NO_ITEM},
                                   {{CL_F, 1, 1}, {CL_C, 2, 1}, NO_ITEM, NO_ITEM},
                                   {{CL_F, 2, 1}, {CL_C, 3, 2}, NO_ITEM, NO_ITEM}}};
constexpr SplitPlan PLAN8_ALL = {{{{CL_DAMP, 0, 6}, NO_ITEM, NO_ITEM, NO_ITEM},
                                  {{CL_DAMP, 6, 6}, NO_ITEM, NO_ITEM, NO_ITEM},
                                  {{CL_DLEF, 0, 6}, NO_ITEM, NO_ITEM, NO_ITEM},
                                  {{CL_DLEF, 6, 1}, {CL_D_RUD, 0, 2}, {CL_D_LEF, 0, 2}, {CL_ETA, 0, 1}},
                                  {{CL_E_LEF, 0, 4}, {CL_YPLEF, 0, 1}, NO_ITEM, NO_ITEM},
                                  {{CL_E_RUD, 0, 4}, {CL_YA20, 0, 1}, NO_ITEM, NO_ITEM},
                                  {{CL_F, 0, 2}, {CL_C, 0, 1}, NO_ITEM, NO_ITEM},
                                  {{CL_F, 2, 1}, {CL_C, 1, 4}, NO_ITEM, NO_ITEM}}};
constexpr SplitPlan PLAN8_FORCE2 = {{{{CL_DAMP, 0, 3}, NO_ITEM, NO_ITEM, NO_ITEM},
                                     {{CL_DAMP, 3, 1}, {CL_DLEF, 0, 2}, NO_ITEM, NO_ITEM},
                                     {{CL_D_RUD, 0, 1}, {CL_D_LEF, 0, 1}, {CL_C, 0, 1}, NO_ITEM},
                                     {{CL_E_LEF, 0, 2}, NO_ITEM, NO_ITEM, NO_ITEM},
                                     {{CL_F, 0, 1}, NO_ITEM, NO_ITEM, NO_ITEM},
                                     {{CL_E_RUD, 0, 1}, {CL_YA20, 0, 1}, NO_ITEM, NO_ITEM},
                                     {{CL_YPLEF, 0, 1}, {CL_C, 1, 1}, NO_ITEM, NO_ITEM},
                                     NPF16_NO_WAVE}};
// AB_EL (FULL, eight waves): C[0:5] + ETA, one net per wave
constexpr SplitPlan PLAN8_EL = {{{{CL_C, 0, 1}, NO_ITEM, NO_ITEM, NO_ITEM},
                                 {{CL_C, 1, 1}, NO_ITEM, NO_ITEM, NO_ITEM},
                                 {{CL_C, 2, 1}, NO_ITEM, NO_ITEM, NO_ITEM},
                                 {{CL_C, 3, 1}, NO_ITEM, NO_ITEM, NO_ITEM},
                                 {{CL_C, 4, 1}, NO_ITEM, NO_ITEM, NO_ITEM},
                                 {{CL_ETA, 0, 1}, NO_ITEM, NO_ITEM, NO_ITEM},
                                 NPF16_NO_WAVE, NPF16_NO_WAVE}};
// AB_ABALL (force-side build-up, four waves): all 36 alpha/beta-only nets + C[0:2] (an imported tile's state; the pipelined back evaluates
// the plans below instead); balanced by VALU instructions per net as above (1 656 / 1 558 / 1 564 / 1 592)
constexpr SplitPlan PLAN_ABALL2 = {{{{CL_DAMP, 0, 12}, NO_ITEM, NO_ITEM, NO_ITEM},
                                    {{CL_DLEF, 0, 7}, {CL_D_RUD, 0, 2}, {CL_D_LEF, 0, 2}, NO_ITEM},
                                    {{CL_F, 0, 3}, {CL_C, 0, 2}, {CL_YA20, 0, 1}, NO_ITEM},
                                    {{CL_E_LEF, 0, 4}, {CL_E_RUD, 0, 4}, {CL_YPLEF, 0, 1}, NO_ITEM},
                                    NPF16_NO_WAVE, NPF16_NO_WAVE, NPF16_NO_WAVE, NPF16_NO_WAVE}};
// The pipelined back spread over the controller call's barrier-free windows (np_planning.hip): the 16 force-side nets + 10 moment-side ones
// between the GRU layers' barriers (AB_GRU: the Overload evaluation needs the force side there), and one cheap single-input net per helper
// wave in each of the three short windows — the L2, A1 and A2 dense layers (5 K cycles each; a net is ~4 K on a wave that waits for its
// scalar weight stream).  Together: all 36 alpha/beta-only nets + C[0:2].
constexpr SplitPlan PLAN_ABGRU = {{{{CL_DAMP, 0, 4}, {CL_DLEF, 0, 2}, {CL_DLEF, 6, 1}, {CL_YA20, 0, 1}},
                                   {{CL_F, 0, 3}, {CL_D_RUD, 0, 1}, NO_ITEM, NO_ITEM},
                                   {{CL_E_LEF, 0, 4}, {CL_C, 0, 2}, {CL_D_RUD, 1, 1}, NO_ITEM},
                                   {{CL_E_RUD, 0, 4}, {CL_YPLEF, 0, 1}, {CL_D_LEF, 0, 2}, NO_ITEM},
                                   NPF16_NO_WAVE, NPF16_NO_WAVE, NPF16_NO_WAVE, NPF16_NO_WAVE}};
constexpr SplitPlan PLAN_WIN_L2 = {{{{CL_DAMP, 4, 1}, NO_ITEM, NO_ITEM, NO_ITEM}, {{CL_DAMP, 5, 1}, NO_ITEM, NO_ITEM, NO_ITEM},
                                    {{CL_DAMP, 6, 1}, NO_ITEM, NO_ITEM, NO_ITEM}, {{CL_DAMP, 7, 1}, NO_ITEM, NO_ITEM, NO_ITEM},
                                    NPF16_NO_WAVE, NPF16_NO_WAVE, NPF16_NO_WAVE, NPF16_NO_WAVE}};
constexpr SplitPlan PLAN_WIN_A1 = {{{{CL_DAMP, 8, 1}, NO_ITEM, NO_ITEM, NO_ITEM}, {{CL_DAMP, 9, 1}, NO_ITEM, NO_ITEM, NO_ITEM},
                                    {{CL_DAMP, 10, 1}, NO_ITEM, NO_ITEM, NO_ITEM}, {{CL_DAMP, 11, 1}, NO_ITEM, NO_ITEM, NO_ITEM},
                                    NPF16_NO_WAVE, NPF16_NO_WAVE, NPF16_NO_WAVE, NPF16_NO_WAVE}};
constexpr SplitPlan PLAN_WIN_A2 = {{{{CL_DLEF, 2, 1}, NO_ITEM, NO_ITEM, NO_ITEM}, {{CL_DLEF, 3, 1}, NO_ITEM, NO_ITEM, NO_ITEM},
                                    {{CL_DLEF, 4, 1}, NO_ITEM, NO_ITEM, NO_ITEM}, {{CL_DLEF, 5, 1}, NO_ITEM, NO_ITEM, NO_ITEM},
                                    NPF16_NO_WAVE, NPF16_NO_WAVE, NPF16_NO_WAVE, NPF16_NO_WAVE}};
// the four together cover AB_ABALL
constexpr bool plans_cover_aball() {
    for (int cl = 0; cl < NUM_CLASSES; cl++) {
        const int hi = cl < NUM_AB_CLASSES ? CLASSES[cl].count : cl == CL_C ? 2 : 0;
        for (int net = 0; net < CLASSES[cl].count; net++) {
            int hits = 0;
            const SplitPlan *ps[4] = {&PLAN_ABGRU, &PLAN_WIN_L2, &PLAN_WIN_A1, &PLAN_WIN_A2};
            for (int q = 0; q < 4; q++)
                for (int w = 0; w < SPLIT_WAVES; w++)
                    for (int k = 0; k < SPLIT_MAX; k++)
                        if (ps[q]->it[w][k].cl == cl && net >= ps[q]->it[w][k].first && net < ps[q]->it[w][k].first + ps[q]->it[w][k].cnt) hits++;
            if (hits != (net < hi ? 1 : 0)) return false;
        }
    }
    return true;
}
static_assert(plans_cover_aball(), "the window plans of the pipelined back must cover every alpha/beta-only net and C[0:2] exactly once");
// a plan must cover exactly the nets eval_ab<PART> + eval_el<N_C, N_ETA> evaluate, each once
constexpr bool plan_covers(const SplitPlan &p, int part, int n_c, int n_eta) {
    for (int cl = 0; cl < NUM_CLASSES; cl++) {
        int lo = 0, hi = 0;
        if (cl < NUM_AB_CLASSES) {
            lo = part == AB_REST ? CLASSES[cl].n_force : 0;
            hi = part == AB_FORCE ? CLASSES[cl].n_force : part == AB_EL ? 0 : CLASSES[cl].count;
        } else {
            hi = cl == CL_C ? n_c : n_eta;
        }
        for (int net = 0; net < CLASSES[cl].count; net++) {
            int hits = 0;
            for (int w = 0; w < SPLIT_WAVES; w++)
                for (int k = 0; k < SPLIT_MAX; k++)
                    if (p.it[w][k].cl == cl && net >= p.it[w][k].first && net < p.it[w][k].first + p.it[w][k].cnt) hits++;
            if (hits != ((net >= lo && net < hi) ? 1 : 0)) return false;
        }
    }
    return true;
}
static_assert(plan_covers(PLAN_REST, AB_REST, 5, 1) && plan_covers(PLAN_ALL, AB_ALL, 5, 1) && plan_covers(PLAN_FORCE2, AB_FORCE, 2, 0),
              "split plans must cover each net of their phase exactly once");
static_assert(plan_covers(PLAN8_REST, AB_REST, 5, 1) && plan_covers(PLAN8_ALL, AB_ALL, 5, 1) && plan_covers(PLAN8_FORCE2, AB_FORCE, 2, 0),
              "eight-wave split plans must cover each net of their phase exactly once");
static_assert(plan_covers(PLAN8_EL, AB_EL, 5, 1) && plan_covers(PLAN_ABALL2, AB_ABALL, 2, 0), "the pipelined schedule's plans must cover each net of their phase exactly once");

template <const SplitPlan &P, int W, int LD>
__device__ __forceinline__ void eval_plan_wave(const AeroWeights &wt, const float (&xn)[NUM_NORM_GROUPS], float *__restrict__ out, bool tables) {
#define NPF16_ITEM(K)                                                                                         \
    if constexpr (P.it[W][K].cnt > 0) eval_class<P.it[W][K].cl, P.it[W][K].cnt, LD, P.it[W][K].first>(wt, xn, out, tables)
    NPF16_ITEM(0);
    NPF16_ITEM(1);
    NPF16_ITEM(2);
    NPF16_ITEM(3);
#undef NPF16_ITEM
}

// ---- pair variant (WPT == 2): the two waves of a 128-aircraft workgroup split the nets of an evaluation between them, and each
// evaluates its half for BOTH waves' aircraft (dual class bodies, np_mlp_asm_dual.inc: set A = the lane's own aircraft, set
// B = the same lane of the other wave; since round 2 in the neuron-major layout — one neuron of both sets per packed register,
// biases fused into the first FMA, records from wt.kblob_dual, np_nets.h::dual_record_len).  Every scalar weight load then feeds two accumulator sets: half the scalar-cache traffic
// per aircraft, twice the arithmetic behind every wait.  The normalised inputs of the partner come through LDS columns
// NUM_LIVE_NETS.. of the same matrix.  Plans balanced by FLOPs as above.
constexpr int PAIR_MAX = 7;
struct PairPlan {
    SplitItem it[2][PAIR_MAX];
};
constexpr PairPlan PAIR_REST = {{{{CL_DAMP, 4, 8}, {CL_DLEF, 2, 5}, {CL_E_RUD, 1, 3}, NO_ITEM, NO_ITEM, NO_ITEM, NO_ITEM},
                                 {{CL_D_RUD, 1, 1}, {CL_D_LEF, 1, 1}, {CL_E_LEF, 2, 2}, {CL_F, 1, 2}, {CL_C, 0, 5}, {CL_ETA, 0, 1}, NO_ITEM}}};
constexpr PairPlan PAIR_ALL = {{{{CL_DAMP, 0, 12}, {CL_DLEF, 0, 7}, {CL_E_LEF, 0, 4}, {CL_YPLEF, 0, 1}, NO_ITEM, NO_ITEM, NO_ITEM},
                                {{CL_D_RUD, 0, 2}, {CL_D_LEF, 0, 2}, {CL_E_RUD, 0, 4}, {CL_F, 0, 3}, {CL_YA20, 0, 1}, {CL_C, 0, 5}, {CL_ETA, 0, 1}}}};
constexpr PairPlan PAIR_FORCE2 = {{{{CL_DAMP, 0, 4}, {CL_DLEF, 0, 2}, {CL_E_LEF, 0, 2}, {CL_E_RUD, 0, 1}, NO_ITEM, NO_ITEM, NO_ITEM},
                                   {{CL_D_RUD, 0, 1}, {CL_D_LEF, 0, 1}, {CL_F, 0, 1}, {CL_YPLEF, 0, 1}, {CL_YA20, 0, 1}, {CL_C, 0, 2}, NO_ITEM}}};
// aero_1d_tables mode: the single-input nets are table lookups (eval_class_pwl, each wave for its own aircraft); only the multi-input
// nets go through the two-set bodies
constexpr PairPlan PAIR_T_REST = {{{{CL_D_RUD, 1, 1}, {CL_D_LEF, 1, 1}, {CL_E_LEF, 2, 2}, {CL_F, 1, 1}, {CL_C, 0, 2}, NO_ITEM, NO_ITEM},
                                   {{CL_E_RUD, 1, 3}, {CL_F, 2, 1}, {CL_C, 2, 3}, NO_ITEM, NO_ITEM, NO_ITEM, NO_ITEM}}};
constexpr PairPlan PAIR_T_ALL = {{{{CL_D_LEF, 0, 1}, {CL_E_LEF, 0, 4}, {CL_E_RUD, 0, 4}, {CL_YA20, 0, 1}, {CL_C, 0, 2}, NO_ITEM, NO_ITEM},
                                  {{CL_D_RUD, 0, 2}, {CL_D_LEF, 1, 1}, {CL_F, 0, 3}, {CL_C, 2, 3}, NO_ITEM, NO_ITEM, NO_ITEM}}};
constexpr PairPlan PAIR_T_FORCE2 = {{{{CL_D_LEF, 0, 1}, {CL_E_RUD, 0, 1}, {CL_F, 0, 1}, {CL_C, 0, 1}, NO_ITEM, NO_ITEM, NO_ITEM},
                                     {{CL_D_RUD, 0, 1}, {CL_E_LEF, 0, 2}, {CL_YA20, 0, 1}, {CL_C, 1, 1}, NO_ITEM, NO_ITEM, NO_ITEM}}};
constexpr bool pair_plan_covers(const PairPlan &p, int part, int n_c, int n_eta, bool multi_input_only = false) {
    for (int cl = 0; cl < NUM_CLASSES; cl++) {
        int lo = 0, hi = 0;
        if (cl < NUM_AB_CLASSES) {
            lo = part == AB_REST ? CLASSES[cl].n_force : 0;
            hi = part == AB_FORCE ? CLASSES[cl].n_force : CLASSES[cl].count;
        } else {
            hi = cl == CL_C ? n_c : n_eta;
        }
        if (multi_input_only && CLASSES[cl].n_in == 1) hi = lo = 0;
        for (int net = 0; net < CLASSES[cl].count; net++) {
            int hits = 0;
            for (int w = 0; w < 2; w++)
                for (int k = 0; k < PAIR_MAX; k++)
                    if (p.it[w][k].cl == cl && net >= p.it[w][k].first && net < p.it[w][k].first + p.it[w][k].cnt) hits++;
            if (hits != ((net >= lo && net < hi) ? 1 : 0)) return false;
        }
    }
    return true;
}
static_assert(pair_plan_covers(PAIR_REST, AB_REST, 5, 1) && pair_plan_covers(PAIR_ALL, AB_ALL, 5, 1) && pair_plan_covers(PAIR_FORCE2, AB_FORCE, 2, 0),
              "pair plans must cover each net of their phase exactly once");
static_assert(pair_plan_covers(PAIR_T_REST, AB_REST, 5, 1, true) && pair_plan_covers(PAIR_T_ALL, AB_ALL, 5, 1, true) &&
                  pair_plan_covers(PAIR_T_FORCE2, AB_FORCE, 2, 0, true),
              "table-mode pair plans must cover each multi-input net of their phase exactly once");
static_assert(NPF16_PAIR_PLAN_CHECK_T_REST_0 && NPF16_PAIR_PLAN_CHECK_T_REST_1 && NPF16_PAIR_PLAN_CHECK_T_ALL_0 && NPF16_PAIR_PLAN_CHECK_T_ALL_1 &&
                  NPF16_PAIR_PLAN_CHECK_T_FORCE2_0 && NPF16_PAIR_PLAN_CHECK_T_FORCE2_1,
              "the table-mode dual phase statements were generated from other plans (tools/gen_mlp_asm.py::PAIR_PLANS)");
static_assert(NPF16_PAIR_PLAN_CHECK_REST_0 && NPF16_PAIR_PLAN_CHECK_REST_1 && NPF16_PAIR_PLAN_CHECK_ALL_0 && NPF16_PAIR_PLAN_CHECK_ALL_1 &&
                  NPF16_PAIR_PLAN_CHECK_FORCE2_0 && NPF16_PAIR_PLAN_CHECK_FORCE2_1,
              "the dual phase statements were generated from other plans (tools/gen_mlp_asm.py::PAIR_PLANS)");

template <int CL, int N, int LD, int FIRST>
__device__ __forceinline__ void eval_class_dual(const AeroWeights &wt, const float (&xa)[NUM_NORM_GROUPS], const float (&xb)[NUM_NORM_GROUPS],
                                                float *__restrict__ out_a, float *__restrict__ out_b) {
    constexpr NetClass c = CLASSES[CL];
    const unsigned addr_a = (unsigned)(unsigned long long)(out_a + (class_slot(CL) + FIRST) * LD);
    const unsigned addr_b = (unsigned)(unsigned long long)(out_b + (class_slot(CL) + FIRST) * LD);
    constexpr int g0 = c.grp[0], g1 = c.grp[c.n_in > 1 ? 1 : 0], g2 = c.grp[c.n_in > 2 ? 2 : 0];
    mlp_class_asm_dual<c.n_in, c.h1, c.h2, c.h3, N, (int)(LD * sizeof(float))>(wt.kblob_dual + dual_class_base(CL) + FIRST * dual_class_stride(CL), addr_a, addr_b, xa[g0],
                                                                              c.n_in > 1 ? xa[g1] : 0.0f, c.n_in > 2 ? xa[g2] : 0.0f, xb[g0],
                                                                              c.n_in > 1 ? xb[g1] : 0.0f, c.n_in > 2 ? xb[g2] : 0.0f);
}

template <const PairPlan &P, int W, int LD>
__device__ __forceinline__ void eval_pair_wave(const AeroWeights &wt, const float (&xa)[NUM_NORM_GROUPS], const float (&xb)[NUM_NORM_GROUPS],
                                               float *__restrict__ out_a, float *__restrict__ out_b) {
#define NPF16_ITEM(K) \
    if constexpr (P.it[W][K].cnt > 0) eval_class_dual<P.it[W][K].cl, P.it[W][K].cnt, LD, P.it[W][K].first>(wt, xa, xb, out_a, out_b)
    NPF16_ITEM(0);
    NPF16_ITEM(1);
    NPF16_ITEM(2);
    NPF16_ITEM(3);
    NPF16_ITEM(4);
    NPF16_ITEM(5);
    NPF16_ITEM(6);
#undef NPF16_ITEM
}

// a wave's slice of a SPLIT plan with the two-set class bodies (WPT_DUAL8)
template <const SplitPlan &P, int W, int LD>
__device__ __forceinline__ void eval_plan_wave_dual(const AeroWeights &wt, const float (&xa)[NUM_NORM_GROUPS], const float (&xb)[NUM_NORM_GROUPS],
                                                    float *__restrict__ out_a, float *__restrict__ out_b) {
#define NPF16_ITEM(K) \
    if constexpr (P.it[W][K].cnt > 0) eval_class_dual<P.it[W][K].cl, P.it[W][K].cnt, LD, P.it[W][K].first>(wt, xa, xb, out_a, out_b)
    NPF16_ITEM(0);
    NPF16_ITEM(1);
    NPF16_ITEM(2);
    NPF16_ITEM(3);
#undef NPF16_ITEM
}

// the same plans with ONE accumulator set (WPT_LAT2: the two waves hold the same 64 aircraft)
template <const PairPlan &P, int W, int LD>
__device__ __forceinline__ void eval_pairplan_single(const AeroWeights &wt, const float (&xn)[NUM_NORM_GROUPS], float *__restrict__ out, bool tables) {
#define NPF16_ITEM(K) \
    if constexpr (P.it[W][K].cnt > 0) eval_class<P.it[W][K].cl, P.it[W][K].cnt, LD, P.it[W][K].first>(wt, xn, out, tables)
    NPF16_ITEM(0);
    NPF16_ITEM(1);
    NPF16_ITEM(2);
    NPF16_ITEM(3);
    NPF16_ITEM(4);
    NPF16_ITEM(5);
    NPF16_ITEM(6);
#undef NPF16_ITEM
}

// All nets of one nlplant evaluation.  With the asm bodies and the default numerics the whole sequence of classes is ONE
// asm statement (tools/gen_mlp_asm.py, "phase functions"): the weight stream keeps running across class boundaries.
// WPT = 4: the latency variant above; `part` is the wave's index within its workgroup (wave-uniform).
#ifndef NPF16_PHASE_ASM
#define NPF16_PHASE_ASM 1
#endif
template <int LD, int PART, bool FULL, int WPT = 1>
__device__ __forceinline__ void eval_nets(const AeroWeights &wt, const float (&xn)[NUM_NORM_GROUPS], float *__restrict__ out, bool tables, int part = 0) {
    constexpr bool has_phase = (FULL && (PART == AB_ALL || PART == AB_REST)) || (!FULL && PART == AB_FORCE);
    static_assert((PART != AB_EL && PART != AB_ABALL && PART != AB_GRU) || (PART == AB_EL && FULL && WPT == 8) || ((PART == AB_ABALL || PART == AB_GRU) && !FULL && WPT == 4),
                  "AB_EL / AB_ABALL / AB_GRU exist as the eight-wave integrator plan and the four-wave force-side plans only");
    if constexpr (WPT == 2) {  // pair variant: `part` = this wave's index in its 128-aircraft workgroup
        static_assert(has_phase, "no pair plan for this evaluation");
        float *out_b = out + 64 - 128 * part;  // the same lane's column in the other wave's half of the matrix
#pragma unroll
        for (int g = 0; g < NUM_NORM_GROUPS; g++) out[(NUM_LIVE_NETS + g) * LD] = xn[g];
        __syncthreads();  // inputs are visible; both waves have finished reading the coefficients of the previous evaluation
#if NPF16_PHASE_ASM  // one asm statement per wave and phase: the weight stream runs across the class boundaries of the wave's plan;
                     // the statements read the normalised inputs of both sets from these LDS columns themselves (no VGPR operands)
        const unsigned base_a = (unsigned)(unsigned long long)out, base_b = (unsigned)(unsigned long long)out_b;
        constexpr int STEP_BYTES = (int)(LD * sizeof(float));
        static_assert((NUM_LIVE_NETS + NUM_NORM_GROUPS) * STEP_BYTES < 65536, "ds_read offsets of the input columns");
#define NPF16_WAVE(W)                                                                                                                  \
    if constexpr (FULL && PART == AB_ALL) mlp_phase_asm_dual_ALL_##W<STEP_BYTES>(wt.kblob_dual + MLP_PAIR_ALL_##W##_START, base_a, base_b);          \
    else if constexpr (FULL && PART == AB_REST) mlp_phase_asm_dual_REST_##W<STEP_BYTES>(wt.kblob_dual + MLP_PAIR_REST_##W##_START, base_a, base_b);  \
    else mlp_phase_asm_dual_FORCE2_##W<STEP_BYTES>(wt.kblob_dual + MLP_PAIR_FORCE2_##W##_START, base_a, base_b)
#define NPF16_WAVE_T(W)                                                                                                                    \
    if constexpr (FULL && PART == AB_ALL) mlp_phase_asm_dual_T_ALL_##W<STEP_BYTES>(wt.kblob_dual + MLP_PAIR_T_ALL_##W##_START, base_a, base_b);          \
    else if constexpr (FULL && PART == AB_REST) mlp_phase_asm_dual_T_REST_##W<STEP_BYTES>(wt.kblob_dual + MLP_PAIR_T_REST_##W##_START, base_a, base_b);  \
    else mlp_phase_asm_dual_T_FORCE2_##W<STEP_BYTES>(wt.kblob_dual + MLP_PAIR_T_FORCE2_##W##_START, base_a, base_b)
        if (tables) {  // wave-uniform.  aero_1d_tables: the single-input nets of this phase are table lookups — every wave for its OWN
                       // aircraft, straight into its own columns — and the pair splits only the multi-input nets
            constexpr int f_damp = PART == AB_REST ? CLASSES[CL_DAMP].n_force : 0, n_damp = (PART == AB_FORCE ? CLASSES[CL_DAMP].n_force : CLASSES[CL_DAMP].count) - f_damp;
            constexpr int f_dlef = PART == AB_REST ? CLASSES[CL_DLEF].n_force : 0, n_dlef = (PART == AB_FORCE ? CLASSES[CL_DLEF].n_force : CLASSES[CL_DLEF].count) - f_dlef;
            constexpr int f_yp = PART == AB_REST ? CLASSES[CL_YPLEF].n_force : 0, n_yp = (PART == AB_FORCE ? CLASSES[CL_YPLEF].n_force : CLASSES[CL_YPLEF].count) - f_yp;
            if constexpr (n_damp > 0) eval_class_pwl<CL_DAMP, n_damp, LD, f_damp>(wt, xn[CLASSES[CL_DAMP].grp[0]], out);
            if constexpr (n_dlef > 0) eval_class_pwl<CL_DLEF, n_dlef, LD, f_dlef>(wt, xn[CLASSES[CL_DLEF].grp[0]], out);
            if constexpr (n_yp > 0) eval_class_pwl<CL_YPLEF, n_yp, LD, f_yp>(wt, xn[CLASSES[CL_YPLEF].grp[0]], out);
            if constexpr (FULL) eval_class_pwl<CL_ETA, 1, LD, 0>(wt, xn[CLASSES[CL_ETA].grp[0]], out);
            if (part == 0) { NPF16_WAVE_T(0); }
            else { NPF16_WAVE_T(1); }
            __syncthreads();
            return;
        }
#undef NPF16_WAVE_T
#else
        float xb[NUM_NORM_GROUPS];
#pragma unroll
        for (int g = 0; g < NUM_NORM_GROUPS; g++) xb[g] = out_b[(NUM_LIVE_NETS + g) * LD];
#define NPF16_WAVE(W)                                                                                    \
    if constexpr (FULL && PART == AB_ALL) eval_pair_wave<PAIR_ALL, W, LD>(wt, xn, xb, out, out_b);           \
    else if constexpr (FULL && PART == AB_REST) eval_pair_wave<PAIR_REST, W, LD>(wt, xn, xb, out, out_b);    \
    else eval_pair_wave<PAIR_FORCE2, W, LD>(wt, xn, xb, out, out_b)
#endif
        if (part == 0) { NPF16_WAVE(0); }
        else { NPF16_WAVE(1); }
#undef NPF16_WAVE
        __syncthreads();  // all coefficient columns of both waves are complete
        return;
    } else if constexpr (WPT == WPT_LAT2) {  // two waves per 64-aircraft tile: the pair plans, evaluated with the single-set class bodies
        static_assert(has_phase, "no split plan for this evaluation");
        __syncthreads();  // both waves have finished reading the coefficients of the previous evaluation
#define NPF16_WAVE(W)                                                                                        \
    if constexpr (FULL && PART == AB_ALL) eval_pairplan_single<PAIR_ALL, W, LD>(wt, xn, out, tables);        \
    else if constexpr (FULL && PART == AB_REST) eval_pairplan_single<PAIR_REST, W, LD>(wt, xn, out, tables); \
    else eval_pairplan_single<PAIR_FORCE2, W, LD>(wt, xn, out, tables)
        if (part == 0) { NPF16_WAVE(0); }
        else { NPF16_WAVE(1); }
#undef NPF16_WAVE
        __syncthreads();  // all coefficient columns are complete
        return;
    } else if constexpr (dual_waves(WPT) != 0) {  // `part` = the wave's index in its 128-aircraft workgroup (0..7 / 0..3); no table mode (the caller's choice of variant)
        static_assert(has_phase, "no split plan for this evaluation");
        constexpr int HW = dual_waves(WPT) / 2;  // waves per half
        float *out_b = out + 64 - 128 * (part / HW);  // the same lane's column in the other half of the matrix
        if (part % HW == 0) {  // the waves of a half hold the same inputs: one of them publishes
#pragma unroll
            for (int g = 0; g < NUM_NORM_GROUPS; g++) out[(NUM_LIVE_NETS + g) * LD] = xn[g];
        }
        __syncthreads();  // inputs are visible; every wave has finished reading the coefficients of the previous evaluation
        float xb[NUM_NORM_GROUPS];
#pragma unroll
        for (int g = 0; g < NUM_NORM_GROUPS; g++) xb[g] = out_b[(NUM_LIVE_NETS + g) * LD];
#define NPF16_WAVE(W)                                                                                                       \
    if constexpr (WPT == WPT_DUAL8) {                                                                                       \
        if constexpr (FULL && PART == AB_ALL) eval_plan_wave_dual<PLAN8_ALL, W, LD>(wt, xn, xb, out, out_b);                \
        else if constexpr (FULL && PART == AB_REST) eval_plan_wave_dual<PLAN8_REST, W, LD>(wt, xn, xb, out, out_b);         \
        else eval_plan_wave_dual<PLAN8_FORCE2, W, LD>(wt, xn, xb, out, out_b);                                              \
    } else {                                                                                                                \
        if constexpr (FULL && PART == AB_ALL) eval_plan_wave_dual<PLAN_ALL, W, LD>(wt, xn, xb, out, out_b);                 \
        else if constexpr (FULL && PART == AB_REST) eval_plan_wave_dual<PLAN_REST, W, LD>(wt, xn, xb, out, out_b);          \
        else eval_plan_wave_dual<PLAN_FORCE2, W, LD>(wt, xn, xb, out, out_b);                                               \
    }
        if (part == 0) { NPF16_WAVE(0); }
        else if (part == 1) { NPF16_WAVE(1); }
        else if (part == 2) { NPF16_WAVE(2); }
        else if (part == 3) { NPF16_WAVE(3); }
        else if constexpr (WPT == WPT_DUAL8) {
            if (part == 4) { NPF16_WAVE(4); }
            else if (part == 5) { NPF16_WAVE(5); }
            else if (part == 6) { NPF16_WAVE(6); }
            else { NPF16_WAVE(7); }
        }
#undef NPF16_WAVE
        __syncthreads();  // all coefficient columns of both halves are complete
        return;
    } else if constexpr (WPT == 4 || WPT == 8) {
        static_assert(has_phase || PART == AB_EL || PART == AB_ABALL || PART == AB_GRU, "no split plan for this evaluation");
        __syncthreads();  // every wave has finished reading the coefficients of the previous evaluation
#define NPF16_WAVE(W)                                                                                                    \
    if constexpr (PART == AB_EL) {                                                                                       \
        eval_plan_wave<PLAN8_EL, W, LD>(wt, xn, out, tables);                                                            \
    } else if constexpr (PART == AB_ABALL) {                                                                             \
        eval_plan_wave<PLAN_ABALL2, W, LD>(wt, xn, out, tables);                                                         \
    } else if constexpr (PART == AB_GRU) {                                                                               \
        eval_plan_wave<PLAN_ABGRU, W, LD>(wt, xn, out, tables);                                                          \
    } else if constexpr (WPT == 8) {                                                                                     \
        if constexpr (FULL && PART == AB_ALL) eval_plan_wave<PLAN8_ALL, W, LD>(wt, xn, out, tables);                     \
        else if constexpr (FULL && PART == AB_REST) eval_plan_wave<PLAN8_REST, W, LD>(wt, xn, out, tables);              \
        else eval_plan_wave<PLAN8_FORCE2, W, LD>(wt, xn, out, tables);                                                   \
    } else {                                                                                                             \
        if constexpr (FULL && PART == AB_ALL) eval_plan_wave<PLAN_ALL, W, LD>(wt, xn, out, tables);                      \
        else if constexpr (FULL && PART == AB_REST) eval_plan_wave<PLAN_REST, W, LD>(wt, xn, out, tables);               \
        else eval_plan_wave<PLAN_FORCE2, W, LD>(wt, xn, out, tables);                                                    \
    }
        if (part == 0) { NPF16_WAVE(0); }
        else if (part == 1) { NPF16_WAVE(1); }
        else if (part == 2) { NPF16_WAVE(2); }
        else if (part == 3) { NPF16_WAVE(3); }
        else if constexpr (WPT == 8) {
            if (part == 4) { NPF16_WAVE(4); }
            else if (part == 5) { NPF16_WAVE(5); }
            else if (part == 6) { NPF16_WAVE(6); }
            else { NPF16_WAVE(7); }
        }
#undef NPF16_WAVE
        __syncthreads();  // all 42 / 16 coefficient columns are complete
        return;
    } else {
#if NPF16_ASM_MLP && NPF16_PHASE_ASM && !defined(NPF16_EXP)
        if constexpr (has_phase) {
            if (!tables) {  // wave-uniform
                const unsigned lds_base = (unsigned)(unsigned long long)out;
#if NPF16_PHASE_X_IN_LDS  // the 9 normalised inputs go to this lane's LDS slots 42..50; each class of the statement reads its own
#pragma unroll
                for (int g = 0; g < NUM_NORM_GROUPS; g++) out[(NUM_LIVE_NETS + g) * LD] = xn[g];
#define NPF16_PHASE_ARGS lds_base
#else
#define NPF16_PHASE_ARGS lds_base, xn
#endif
                if constexpr (FULL && PART == AB_ALL) mlp_phase_asm_ALL<(int)(LD * sizeof(float))>(wt.kblob + MLP_PHASE_ALL_START, NPF16_PHASE_ARGS);
                else if constexpr (FULL && PART == AB_REST) mlp_phase_asm_REST<(int)(LD * sizeof(float))>(wt.kblob + MLP_PHASE_REST_START, NPF16_PHASE_ARGS);
                else mlp_phase_asm_FORCE2<(int)(LD * sizeof(float))>(wt.kblob + MLP_PHASE_FORCE2_START, NPF16_PHASE_ARGS);
#undef NPF16_PHASE_ARGS
                return;
            }
        }
#endif
        eval_ab<LD, PART>(wt, xn, out, tables);
        eval_el<(FULL ? 5 : 2), (FULL ? 1 : 0), LD>(wt, xn, out, tables);
    }
}

// The 9 distinct input normalisations (X - mean) / std of mean_std.csv.
__device__ __forceinline__ void normalise_inputs(const AeroWeights &wt, float alpha_deg, float beta_deg, float el, float (&xn)[NUM_NORM_GROUPS]) {
#pragma unroll
    for (int g = 0; g < NUM_NORM_GROUPS; g++) {
        const float v = (g <= G_A_RUD) ? alpha_deg : (g <= G_B_O ? beta_deg : el);
        const cf32_ptr h = NPF16_CONST(wt.kblob) + KBLOB_NORM_STRIDE * g;  // (mean, sigma, RN(1 / sigma))
        xn[g] = np_divc(v - h[0], h[1], h[2]);
    }
}

// ---------------------------------------------------------------------------------------------
// F16Dynamics.nlplant — envs/models/F16/F16_dynamics.py:37-228 (atmos :22-35)
// ---------------------------------------------------------------------------------------------
template <class A>  // Airframe, or Airframe in the constant address space
__device__ __forceinline__ float atmos_pow(const A &af, float alt) {   // (1 - 0.703e-5 alt)^4.14 (F16_dynamics.py:22-35)
    return np_pow_posexp(1.0f - af.atm_lapse * alt, af.atm_exp);
}
template <class A>
__device__ __forceinline__ float eas2tas_of(const A &af, float alt) {
    const float e = (1.0f / atmos_pow(af, alt)) * 1.0f;
    return sqrtf(e);
}

struct Trig {  // sines/cosines of the attitude and flow angles of one state
    float sa, ca, sb, cb, st, ct, sphi, cphi;
};

// Aerodynamic force/moment coefficients -> xdot[6..11] (+ xdot[0..5] when FULL).
// FULL=false builds only xdot[6..8] (the Overload check, overload.py:37-42 -> F16_model.py:132-148)
// from the 16 force-side coefficients: identical arithmetic for those three outputs.
// PART: which alpha/beta-only nets are evaluated here — AB_ALL, AB_FORCE (FULL=false), or AB_REST
// when the 14 force-side slots of `coef` already hold the values of THIS state (carried over from
// the Overload evaluation of the previous step).
// SHARE (latency variant, WPT == 4, Euler): the serial fp64 chains of the state — sin / cos of alpha, beta, theta, phi (+ tan theta,
// sin / cos psi when FULL) and tfac^4.14 — are evaluated ONCE per tile instead of once per wave: wave w computes its share
// before the barrier that opens the net evaluation and publishes it in LDS columns NUM_LDS_SLOTS + 12 * SET .. (SET alternates
// between the two evaluations of a step, so a fast wave never overwrites what a slow one still reads); after the evaluation every
// wave reads all twelve values back.  `sc` is then an OUTPUT (the caller's observation / Overload code uses it).  Same operations on
// the same inputs: bit-identical to every wave computing everything (12 of a lone wave's ~23 us were these chains).
struct StateScalars {
    Trig tr;
    float tt, spsi, cpsi, powv;  // tan(theta), sin / cos(psi), (1 - 0.703e-5 alt)^4.14
};
constexpr int NUM_SHARED_SCALARS = 12;
// STAGE (SHARE only; the persistent PlanningEnv kernel splits an evaluation in time): 0 = everything; 1 = this wave's share of the state's
// serial chains -> LDS and nothing else (no barrier: the caller publishes them); 2 = everything BUT the share computation (the set is
// already in LDS and published).
// AF: how the airframe constants are reached — AirframeVia<AP> (step kernels: re-read from the kernel arguments after the MLP phase) or Airframe
template <bool FULL, int PART, int LD, int WPT = 1, bool SHARE = false, int SET = 0, bool HAVE_POW = false, int STAGE = 0, class AF>
__device__ __forceinline__ void nlplant(const AeroWeights &wt, const AF &afv, const float (&s)[12], const float (&u)[4], StateScalars &sc,
                                        float *__restrict__ coef, bool tables, float (&xd)[12], int part = 0) {
    static_assert(!SHARE || WPT == 4 || WPT == 8 || WPT == WPT_LAT2, "shared state scalars belong to the latency variants");
    static_assert(STAGE == 0 || SHARE, "stages split the shared-scalar evaluation");
    const float r2d = (float)(180.0 / 3.141592653589793);

    const float alt = s[2];
    float vt = s[6];
    const float alpha = s[7] * r2d, beta = s[8] * r2d;
    const float P = s[9], Q = s[10], R = s[11];
    const float T = u[0], el = u[1], ail = u[2], rud = u[3];

    if constexpr (SHARE && STAGE != 2) {  // this wave's share of the state's serial chains -> LDS (published by the barrier inside eval_nets)
        float *shr = coef + (NUM_LDS_SLOTS + NUM_SHARED_SCALARS * SET) * LD;
        // (before the MLP phase only the atmosphere's two constants are needed, by the wave that computes tfac^4.14)
        const auto &a0 = afv.get();
        const float tfac = 1.0f - a0.atm_lapse * alt;
        const double atm_exp = a0.atm_exp;
        if constexpr (WPT == WPT_LAT2) {  // two waves: wave 0 alpha, theta (+ tan), psi; wave 1 beta, phi, tfac^4.14
            float a_, b_, c_ = 0.0f;
            if (part == 0) {
                np_sincos(s[7], a_, b_);
                shr[0 * LD] = a_;
                shr[1 * LD] = b_;
                if (FULL || NUM_CACHED_TRIG > 0) np_sincostan(s[4], a_, b_, c_);   // tan(theta) of the new state travels in the cross-step cache
                else np_sincos(s[4], a_, b_);
                shr[4 * LD] = a_;
                shr[5 * LD] = b_;
                shr[8 * LD] = c_;
                if (FULL) {
                    np_sincos(s[5], a_, b_);
                    shr[9 * LD] = a_;
                    shr[10 * LD] = b_;
                }
            } else {
                np_sincos(s[8], a_, b_);
                shr[2 * LD] = a_;
                shr[3 * LD] = b_;
                np_sincos(s[3], a_, b_);
                shr[6 * LD] = a_;
                shr[7 * LD] = b_;
                shr[11 * LD] = np_pow_posexp(tfac, atm_exp);
            }
        } else if (part == 0 || (WPT == 8 && part == 4)) {  // four waves: wave 0 takes alpha and psi; eight: wave 4 takes psi
            float a_, b_;
            if (part == 0) {
                np_sincos(s[7], a_, b_);
                shr[0 * LD] = a_;
                shr[1 * LD] = b_;
            }
            if (FULL && (WPT == 4 || part == 4)) {
                np_sincos(s[5], a_, b_);
                shr[9 * LD] = a_;
                shr[10 * LD] = b_;
            }
        } else if (part == 1) {
            float a_, b_;
            np_sincos(s[8], a_, b_);
            shr[2 * LD] = a_;
            shr[3 * LD] = b_;
            shr[11 * LD] = np_pow_posexp(tfac, atm_exp);
        } else if (part == 2) {
            float a_, b_, c_ = 0.0f;
            if (FULL || NUM_CACHED_TRIG > 0) np_sincostan(s[4], a_, b_, c_);   // tan(theta) of the new state travels in the cross-step cache
            else np_sincos(s[4], a_, b_);
            shr[4 * LD] = a_;
            shr[5 * LD] = b_;
            shr[8 * LD] = c_;
        } else if (part == 3) {
            float a_, b_;
            np_sincos(s[3], a_, b_);
            shr[6 * LD] = a_;
            shr[7 * LD] = b_;
        }
    }

    if constexpr (STAGE == 1) return;
    float xn[NUM_NORM_GROUPS];
    normalise_inputs(wt, alpha, beta, el, xn);
    // non-finite inputs poison every coefficient (numerics spec, "non-finite inputs")
    const float chk = ((alpha - alpha) + (beta - beta)) + (el - el);
    const bool ok = (chk == chk);
    const float qnan = __builtin_nanf("");
    eval_nets<LD, PART, FULL, WPT>(wt, xn, coef, tables, part);
    // "non-finite inputs poison every coefficient": every coefficient enters xdot through one of the six totals below, and a NaN
    // coefficient makes its total NaN — so the rule is applied to the six totals (six selects) instead of to each of the 42 / 16
    // coefficient reads; same values, same NaN-ness of every output (the payload of a NaN is not part of the spec)
#define NPF16_NET(id) (coef[slot_of(id) * LD])
#define NPF16_POISON(v) (ok ? (v) : qnan)

    const auto &af = afv.get();   // the airframe constants: scalar loads issued here, behind the MLP phase
    if constexpr (SHARE) {
        const float *shr = coef + (NUM_LDS_SLOTS + NUM_SHARED_SCALARS * SET) * LD;
        sc.tr.sa = shr[0 * LD]; sc.tr.ca = shr[1 * LD]; sc.tr.sb = shr[2 * LD]; sc.tr.cb = shr[3 * LD];
        sc.tr.st = shr[4 * LD]; sc.tr.ct = shr[5 * LD]; sc.tr.sphi = shr[6 * LD]; sc.tr.cphi = shr[7 * LD];
        sc.tt = shr[8 * LD];
        sc.spsi = FULL ? shr[9 * LD] : 0.0f;
        sc.cpsi = FULL ? shr[10 * LD] : 0.0f;
        sc.powv = shr[11 * LD];
    } else if constexpr (!HAVE_POW) {  // HAVE_POW: the caller brought it (the cross-step cache: the previous step computed it for this state)
        sc.powv = atmos_pow(af, alt);
    }
    const float g = af.g, mass = af.mass, B = af.B, S = af.S, cbar = af.cbar, Heng = af.Heng;
    const float Jy = af.Jy, Jxz = af.Jxz, Jz = af.Jz, Jx = af.Jx;
    const float xc = af.xc, cbar_over_B = af.cbar_over_B, c1 = af.c1, c2 = af.c2, c3 = af.c3, c4 = af.c4, denom = af.denom;
    const Trig &tr = sc.tr;
    const float tt = sc.tt, spsi = sc.spsi, cpsi = sc.cpsi;
    const float sa = tr.sa, ca = tr.ca, sb = tr.sb, cb = tr.cb, st = tr.st, ct = tr.ct, sphi = tr.sphi, cphi = tr.cphi;

    vt = (vt <= 0.01f ? 1.0f : 0.0f) * 0.01f + (vt > 0.01f ? 1.0f : 0.0f) * vt;  // :104

    const float dail = np_divc(ail, af.ail_ref, af.r_ail_ref), drud = np_divc(rud, af.rud_ref, af.r_rud_ref);  // lef == 0 -> dlef == 1 (exact)

    const float rho = af.rho0 * sc.powv;
    const float qbar = (0.5f * rho) * (vt * vt);

    const float U = (vt * ca) * cb, V = vt * sb, W = (vt * sa) * cb;

    if (FULL) {
        xd[0] = (U * (ct * cpsi) + V * ((sphi * cpsi) * st - cphi * spsi)) + W * ((cphi * st) * cpsi + sphi * spsi);
        xd[1] = (U * (ct * spsi) + V * ((sphi * spsi) * st + cphi * cpsi)) + W * ((cphi * st) * spsi - sphi * cpsi);
        xd[2] = (U * st - V * (sphi * ct)) - W * (cphi * ct);
        xd[3] = P + tt * (Q * sphi + R * cphi);
        xd[4] = Q * cphi - R * sphi;
        xd[5] = (Q * sphi + R * cphi) / ct;
    }

    const float inv2vt = 1.0f / (2.0f * vt);
    const float c2v = inv2vt * cbar;
    const float b2v = inv2vt * B;

    // --- X / Z force ---------------------------------------------------------------------
    const float dCx_lef = NPF16_NET(N_dCx_lef);
    const float dXdQ = c2v * (NPF16_NET(N_Cxq) + NPF16_NET(N_dCxq_lef));
    const float Cx_tot = NPF16_POISON((NPF16_NET(N_Cx) + dCx_lef) + dXdQ * Q);
    const float dCz_lef = NPF16_NET(N_dCz_lef);
    const float dZdQ = c2v * (NPF16_NET(N_Czq) + dCz_lef);  // reference uses delta_Cz_lef here (:199)
    const float Cz_tot = NPF16_POISON((NPF16_NET(N_Cz) + dCz_lef) + dZdQ * Q);
    // --- Y force ---------------------------------------------------------------------------
    const float dYdail = NPF16_NET(N_dCy_a20) + NPF16_NET(N_dCy_a20_lef);
    const float dYdR = b2v * (NPF16_NET(N_Cyr) + NPF16_NET(N_dCyr_lef));
    const float dYdP = b2v * (NPF16_NET(N_Cyp) + NPF16_NET(N_dCyp_lef));
    const float Cy_tot =
        NPF16_POISON(((((NPF16_NET(N_Cy) + NPF16_NET(N_dCy_lef)) + dYdail * dail) + NPF16_NET(N_dCy_r30) * drud) + dYdR * R) + dYdP * P);

    const float r_mass = af.r_mass;
    const float Udot = (((R * V - Q * W) - g * st) + np_divc((qbar * S) * Cx_tot, mass, r_mass)) + np_divc(T, mass, r_mass);
    const float Vdot = ((P * W - R * U) + (g * ct) * sphi) + np_divc((qbar * S) * Cy_tot, mass, r_mass);
    const float Wdot = ((Q * U - P * V) + (g * ct) * cphi) + np_divc((qbar * S) * Cz_tot, mass, r_mass);
    xd[6] = ((U * Udot + V * Vdot) + W * Wdot) / vt;
    xd[7] = (U * Wdot - W * Udot) / (U * U + W * W);
    xd[8] = (Vdot * vt - V * xd[6]) / ((vt * vt) * cb);

    if (FULL) {
        // --- pitching moment ----------------------------------------------------------------
        const float dMdQ = c2v * (NPF16_NET(N_Cmq) + NPF16_NET(N_dCmq_lef));
        const float Cm_tot =
            NPF16_POISON(((((NPF16_NET(N_Cm) * NPF16_NET(N_eta_el) + Cz_tot * xc) + NPF16_NET(N_dCm_lef)) + dMdQ * Q) + NPF16_NET(N_dCm)) +
                         0.0f);
        // --- yawing moment ------------------------------------------------------------------
        const float dNdail = NPF16_NET(N_dCn_a20) + NPF16_NET(N_dCn_a20_lef);
        const float dNdR = b2v * (NPF16_NET(N_Cnr) + NPF16_NET(N_dCnr_lef));
        const float dNdP = b2v * (NPF16_NET(N_Cnp) + NPF16_NET(N_dCnp_lef));
        const float Cn_tot = NPF16_POISON(((((((NPF16_NET(N_Cn) + NPF16_NET(N_dCn_lef)) - (Cy_tot * xc) * cbar_over_B) + dNdail * dail) +
                                             NPF16_NET(N_dCn_r30) * drud) + dNdR * R) + dNdP * P) + NPF16_NET(N_dCnbeta) * beta);
        // --- rolling moment -----------------------------------------------------------------
        const float dLdail = NPF16_NET(N_dCl_a20) + NPF16_NET(N_dCl_a20_lef);
        const float dLdR = b2v * (NPF16_NET(N_Clr) + NPF16_NET(N_dClr_lef));
        const float dLdP = b2v * (NPF16_NET(N_Clp) + NPF16_NET(N_dClp_lef));
        const float Cl_tot = NPF16_POISON((((((NPF16_NET(N_Cl) + NPF16_NET(N_dCl_lef)) + dLdail * dail) + NPF16_NET(N_dCl_r30) * drud) +
                                            dLdR * R) + dLdP * P) + NPF16_NET(N_dClbeta) * beta);

        const float L_tot = ((Cl_tot * qbar) * S) * B;
        const float M_tot = ((Cm_tot * qbar) * S) * cbar;
        const float N_tot = ((Cn_tot * qbar) * S) * B;
        xd[9] = np_divc((((Jz * L_tot + Jxz * N_tot) - (c1 * Q) * R) + (c2 * P) * Q) + (Jxz * Q) * Heng, denom, af.r_denom);
        xd[10] = np_divc(((M_tot + (c3 * P) * R) - Jxz * (P * P - R * R)) - R * Heng, Jy, af.r_Jy);
        xd[11] = np_divc((((Jx * N_tot + Jxz * L_tot) + (c4 * P) * Q) - (c2 * Q) * R) + (Jx * Q) * Heng, denom, af.r_denom);
    }
#undef NPF16_NET
#undef NPF16_POISON
}

// the state's trigonometry computed by the caller (every variant but the latency one)
template <bool FULL, int PART, int LD, int WPT = 1, class AF>
__device__ __forceinline__ void nlplant(const AeroWeights &wt, const AF &afv, const float (&s)[12], const float (&u)[4], const Trig &tr, float tt, float spsi,
                                        float cpsi, float *__restrict__ coef, bool tables, float (&xd)[12], int part = 0) {
    StateScalars sc;
    sc.tr = tr;
    sc.tt = tt;
    sc.spsi = spsi;
    sc.cpsi = cpsi;
    nlplant<FULL, PART, LD, WPT, false, 0>(wt, afv, s, u, sc, coef, tables, xd, part);
}

// F16Model.update's first-order control lag (F16_model.py:52-62), left to right exactly as written:
//   T' = 0.9 T + 0.1 a0 * 0.225 * 76300 / 0.3048;  el' = 0.9 el + 0.1 a1 * 45  (ail, rud alike)
template <class A>  // Airframe, or Airframe in the constant address space
__device__ __forceinline__ void control_lag(const A &af, const float (&act)[4], float (&u)[4]) {
    u[0] = af.lag_keep * u[0] + np_divc(((af.lag_new * act[0]) * af.thrust_frac) * af.thrust_max, af.thrust_unit, af.r_thrust_unit);
    u[1] = af.lag_keep * u[1] + (af.lag_new * act[1]) * af.surf_max[0];
    u[2] = af.lag_keep * u[2] + (af.lag_new * act[2]) * af.surf_max[1];
    u[3] = af.lag_keep * u[3] + (af.lag_new * act[3]) * af.surf_max[2];
}

__device__ __forceinline__ void trig_of(const float (&s)[12], Trig &tr, float &tt) {
    np_sincos(s[7], tr.sa, tr.ca);
    np_sincos(s[8], tr.sb, tr.cb);
    np_sincostan(s[4], tr.st, tr.ct, tt);
    np_sincos(s[3], tr.sphi, tr.cphi);
}

// full derivative at (s,u) including the heading terms
template <int PART, int LD, int WPT = 1, class AF>
__device__ __forceinline__ void xdot_full(const AeroWeights &wt, const AF &afv, const float (&s)[12], const float (&u)[4], float *__restrict__ coef, bool tables,
                                          float (&xd)[12], int part = 0) {
    Trig tr;
    float tt, spsi, cpsi;
    trig_of(s, tr, tt);
    np_sincos(s[5], spsi, cpsi);
    nlplant<true, PART, LD, WPT>(wt, afv, s, u, tr, tt, spsi, cpsi, coef, tables, xd, part);
}
// F16Model.get_acceleration — F16_model.py:132-148, from xdot[6..8] at (s,u)
__device__ __forceinline__ void body_acceleration(const float (&s)[12], const Trig &tr, const float (&xd)[12], float (&a)[3]) {
    const float sina = tr.sa, cosa = tr.ca, sinb = tr.sb, cosb = tr.cb;
    const float vt = s[6];
    const float vel_u = (vt * cosb) * cosa, vel_v = vt * sinb, vel_w = (vt * cosb) * sina;
    const float u_dot = ((cosb * cosa) * xd[6] - ((vt * sinb) * cosa) * xd[8]) - ((vt * cosb) * sina) * xd[7];
    const float v_dot = sinb * xd[6] + (vt * cosb) * xd[8];
    const float w_dot = ((cosb * sina) * xd[6] - ((vt * sinb) * sina) * xd[8]) + ((vt * cosb) * cosa) * xd[7];
    a[0] = (u_dot + s[10] * vel_w) - s[11] * vel_v;
    a[1] = (v_dot + s[11] * vel_u) - s[9] * vel_w;
    a[2] = (w_dot + s[9] * vel_v) - s[10] * vel_u;
}


// ---------------------------------------------------------------------------------------------
// reset of one flagged aircraft — F16_model.py:33-45 + task.reset + env_base.py:92
// ---------------------------------------------------------------------------------------------
template <int TASK, class CFG>  // CFG: DevCfg, or DevCfg in the constant address space (scalar loads)
__device__ __forceinline__ void reset_row(const CFG &cfg, const float (&ru)[5], float (&s)[12], float (&u)[4],
                                          float (&tgt)[3], long long &step_count) {
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = 0.0f;
    u[1] = u[2] = u[3] = 0.0f;
    s[2] = ru[0] * cfg.alt_span + cfg.min_altitude;
    s[6] = ru[1] * cfg.vt_span + cfg.min_vt;
    u[0] = cfg.init_T;
    if (TASK == 0) {  // heading_task.py:49-69
        tgt[0] = s[2] + 1000.0f;
        tgt[1] = np_wrap_pi(s[5] + (float)(2.0 * 3.141592653589793 / 3.0));
        tgt[2] = s[6] + 0.0f;
    } else if (TASK == 1) {  // control_task.py:49-68
        const float dp = (2.0f * (ru[2] - 0.5f)) * cfg.max_pitch_increment;
        const float dh = (2.0f * (ru[3] - 0.5f)) * cfg.max_heading_increment;
        const float dv = (2.0f * (ru[4] - 0.5f)) * cfg.max_velocities_u_increment;
        tgt[0] = np_wrap_pi(s[4] + dp);
        tgt[1] = np_wrap_pi(s[5] + dh);
        tgt[2] = s[6] + dv;
    } else {  // tracking_task.py:48-71
        const float PI_F = 3.14159265358979323846f;
        const float dist = ru[2] * cfg.dist_span + cfg.min_distance;
        const float th1 = (ru[3] * PI_F) / 3.0f - (float)(3.141592653589793 / 6.0);
        const float th2 = (ru[4] * PI_F) / 3.0f - (float)(3.141592653589793 / 6.0);
        float s1, c1, s2, c2;
        np_sincos(th1, s1, c1);
        np_sincos(th2, s2, c2);
        tgt[0] = s[0] + (dist * c1) * c2;
        tgt[1] = s[1] + (dist * c1) * s2;
        tgt[2] = s[2] + dist * s1;
    }
    step_count = 0;
}

// ---------------------------------------------------------------------------------------------
// observation (before noise) — heading_task.py:71-152 / control_task.py:70-152 / tracking_task.py:73-155
// ---------------------------------------------------------------------------------------------
template <int TASK, bool HAVE_POW = false, class CFG>  // CFG: DevCfg, or DevCfg in the constant address space (scalar loads)
__device__ __forceinline__ void observe(const CFG &cfg, const float (&s)[12], const float (&u)[4], const float (&tgt)[3],
                                        const Trig &tr, float (&o)[22], float powv = 0.0f) {
    const float alt = s[2], pitch = s[4], heading = s[5], vt = s[6];
    // F16Model.get_EAS2TAS (F16_model.py:156-162); powv = (1 - 0.703e-5 alt)^4.14 when the caller already has it (latency variant)
    const float eas2tas = HAVE_POW ? sqrtf((1.0f / powv) * 1.0f) : eas2tas_of(cfg.af, alt);
    const float TAS = vt + cfg.airspeed * 1.0f;
    const float EAS = TAS / eas2tas;
    if (TASK == 0) {
        o[0] = NP_DIVC((alt - tgt[0]) * 0.3048f, 1000.0f);
        o[1] = np_wrap_pi(heading - tgt[1]);
        o[2] = NP_DIVC((vt - tgt[2]) * 0.3048f, 340.0f);
    } else if (TASK == 1) {
        o[0] = np_wrap_pi(pitch - tgt[0]);
        o[1] = np_wrap_pi(heading - tgt[1]);
        o[2] = NP_DIVC((vt - tgt[2]) * 0.3048f, 340.0f);
    } else {
        o[0] = NP_DIVC((s[0] - tgt[0]) * 0.3048f, 1000.0f);
        o[1] = NP_DIVC((s[1] - tgt[1]) * 0.3048f, 1000.0f);
        o[2] = NP_DIVC((alt - tgt[2]) * 0.3048f, 1000.0f);
    }
    o[3] = NP_DIVC(alt * 0.3048f, 5000.0f);
    o[4] = tr.sphi;
    o[5] = tr.cphi;
    o[6] = tr.st;
    o[7] = tr.ct;
    o[8] = NP_DIVC(EAS * 0.3048f, 340.0f);
    o[9] = tr.sa;
    o[10] = tr.ca;
    o[11] = tr.sb;
    o[12] = tr.cb;
    o[13] = s[9];
    o[14] = s[10];
    o[15] = s[11];
    o[16] = NP_DIVC(NP_DIVC(u[0], 0.225f), 76300.0f) * 0.3048f;
    o[17] = NP_DIVC(u[1], 45.0f);
    o[18] = NP_DIVC(u[2], 45.0f);
    o[19] = NP_DIVC(u[3], 45.0f);
    o[20] = 0.0f / 45.0f;  // lef
    o[21] = eas2tas;
}

// Observation noise (numerics spec): 22 normals of (seed, call_idx, global row) from FOUR Philox4x32-10 blocks
// (counter blocks 2..5) by Box-Muller on 11 pairs.  Every pair takes a 21-bit radius index K1 and a 21-bit direction index K2:
// pairs 0..7 the top 21 bits of two words, pairs 8..10 are assembled from the low 11 bits of the 16 words.
//   u = (K1 + 0.5) * 2^-21;  o[2i], o[2i+1] += (sqrt(-2 ln u) * scale) * (cos, sin)(direction K2), one fma each
// one Box-Muller pair from its two 21-bit indices: radius x scale, and the direction
__device__ __forceinline__ void noise_pair(uint32_t k1, uint32_t k2, float scale, float &rs, float &cs, float &sn) {
    const float u = fmaf((float)k1, 4.76837158203125e-07f, 2.384185791015625e-07f);  // (K1 + 0.5) * 2^-21, exact
    rs = sqrt_spec(neg2ln_spec(u)) * scale;
    unit_vector_spec(k2, cs, sn);
}
// the index pairs one Philox block contributes: pairs 2b and 2b+1 from the top 21 bits of its words and (b < 3) pair 8+b from
// their low 11 bits
__device__ __forceinline__ void noise_block_indices(const uint32_t (&w)[4], uint32_t (&k1)[3], uint32_t (&k2)[3]) {
    k1[0] = w[0] >> 11;
    k2[0] = w[1] >> 11;
    k1[1] = w[2] >> 11;
    k2[1] = w[3] >> 11;
    k1[2] = ((w[0] & 0x7FFu) << 10) | ((w[1] >> 1) & 0x3FFu);
    k2[2] = ((w[2] & 0x7FFu) << 10) | ((w[3] >> 1) & 0x3FFu);
}
// two pairs at once (np_math.h: the packed forms of the same sequences): o[2 pa], o[2 pa + 1], o[2 pb], o[2 pb + 1]
__device__ __forceinline__ void add_noise_pairs2(uint32_t k1a, uint32_t k2a, int pa, uint32_t k1b, uint32_t k2b, int pb, float scale,
                                                 float (&o)[22]) {
    np_f32x2 kf = {(float)k1a, (float)k1b};
    const np_f32x2 u = np_fma2(kf, 4.76837158203125e-07f, 2.384185791015625e-07f);
    const np_f32x2 rs = sqrt_spec2(neg2ln_spec2(u)) * (np_f32x2)(scale);
    const uint32_t k2[2] = {k2a, k2b};
    np_f32x2 cs, sn;
    unit_vector_spec2(k2, cs, sn);
    np_f32x2 oc = {o[2 * pa], o[2 * pb]}, os = {o[2 * pa + 1], o[2 * pb + 1]};
    oc = np_fma2(rs, cs, oc);
    os = np_fma2(rs, sn, os);
    o[2 * pa] = oc[0];
    o[2 * pb] = oc[1];
    o[2 * pa + 1] = os[0];
    o[2 * pb + 1] = os[1];
}
__device__ __forceinline__ void add_rng_noise(uint64_t seed, uint64_t call_idx, int64_t row, float scale, float (&o)[22]) {
    uint32_t k1[4][3], k2[4][3];
#pragma unroll
    for (uint32_t b = 0; b < 4; b++) {
        uint32_t blk[4];
        rng_block(seed, call_idx, row, 2 + b, blk);
        noise_block_indices(blk, k1[b], k2[b]);
        add_noise_pairs2(k1[b][0], k2[b][0], 2 * (int)b, k1[b][1], k2[b][1], 2 * (int)b + 1, scale, o);   // pairs 2b, 2b + 1
    }
    add_noise_pairs2(k1[0][2], k2[0][2], 8, k1[1][2], k2[1][2], 9, scale, o);                               // pairs 8, 9
    float rs, cs, sn;                                                                                       // pair 10 (the 4th block's low bits are unused)
    noise_pair(k1[2][2], k2[2][2], scale, rs, cs, sn);
    o[20] = fmaf(rs, cs, o[20]);
    o[21] = fmaf(rs, sn, o[21]);
}

// ---------------------------------------------------------------------------------------------
// terminations + reward at the new state — termination_conditions/*.py, reward_functions/*.py,
// task_base.py:60-96, env_base.py:70-75
// ---------------------------------------------------------------------------------------------
template <int TASK, class CFG>  // CFG: DevCfg, or DevCfg in the constant address space (scalar loads)
__device__ __forceinline__ void done_and_reward(const CFG &cfg, const float (&s)[12], const float (&tgt)[3],
                                                const float (&acc3)[3], long long step_count, bool done_prev, bool bad_prev,
                                                bool &done, bool &bad, float &reward, unsigned &reasons, float &reward_task) {
    // `reasons`: which condition fired at THIS state (NP_TERM_* bits) — what the reference prints per condition
    const float PI_F = 3.14159265358979323846f;
    const float acc = sqrtf((acc3[0] * acc3[0] + acc3[1] * acc3[1]) + acc3[2] * acc3[2]);
    const bool r_over = (acc - cfg.acceleration_limit) > 0.0f;        // overload.py:37-42
    const bool r_low = (s[2] - cfg.altitude_limit) < 0.0f;             // low_altitude.py:29-30
    const float TAS = s[6] + cfg.airspeed * 1.0f;
    const float vel = NP_DIVC(TAS * 0.3048f, 340.0f);
    const bool r_fast = (vel - cfg.max_velocity) >= 0.0f;              // high_speed.py:29-30
    const bool r_slow = (vel - cfg.min_velocity) <= 0.0f;              // low_speed.py:29-30
    const float alpha = NP_DIVC(s[7] * 180.0f, PI_F), beta = NP_DIVC(s[8] * 180.0f, PI_F);
    const bool r_ext = ((alpha < cfg.min_alpha) | (alpha > cfg.max_alpha)) | ((beta < cfg.min_beta) | (beta > cfg.max_beta));  // extreme_state.py:32-36
    bool b = (((r_over | r_low) | r_fast) | r_slow) | r_ext;
    const float pi36 = (float)(3.141592653589793 / 36.0);
    const bool m1 = step_count >= cfg.max_check_interval;
    bool m2 = true, m3, m4, m5;
    float rew;
    if (TASK == 0) {  // unreach_heading.py:38-53, heading_reward.py:26-36
        m2 = step_count >= cfg.min_check_interval;
        const float dpsi = np_wrap_pi(s[5] - tgt[1]);
        m3 = fabsf(dpsi) >= pi36;
        m4 = fabsf(s[2] - tgt[0]) >= 100.0f;
        m5 = fabsf(s[6] - tgt[2]) >= 20.0f;
        const float da = NP_DIVC((s[2] - tgt[0]) * 0.3048f, 1000.0f);
        const float dh = NP_DIVC(dpsi, PI_F);
        const float dv = NP_DIVC((s[6] - tgt[2]) * 0.3048f, 340.0f);
        rew = (-(da * da) + -(dh * dh)) + -(dv * dv);
    } else if (TASK == 1) {  // unreach_posture.py:40-55, posture_reward.py:26-35
        const float dpsi = np_wrap_pi(s[5] - tgt[1]);
        m3 = fabsf(dpsi) >= pi36;
        m4 = fabsf(s[4] - tgt[0]) >= pi36;
        m5 = fabsf(s[6] - tgt[2]) >= 20.0f;
        const float dp = NP_DIVC(np_wrap_pi(s[4] - tgt[0]), PI_F);
        const float dh = NP_DIVC(dpsi, PI_F);
        const float dv = NP_DIVC((s[6] - tgt[2]) * 0.3048f, 340.0f);
        rew = (-(dp * dp) + -(dh * dh)) + -(dv * dv);
    } else {  // unreach_target.py:38-47, position_reward.py:26-34
        m3 = fabsf(s[0] - tgt[0]) >= 100.0f;
        m4 = fabsf(s[1] - tgt[1]) >= 100.0f;
        m5 = fabsf(s[2] - tgt[2]) >= 100.0f;
        const float dn = NP_DIVC((s[0] - tgt[0]) * 0.3048f, 1000.0f);
        const float de = NP_DIVC((s[1] - tgt[1]) * 0.3048f, 1000.0f);
        const float da = NP_DIVC((s[2] - tgt[2]) * 0.3048f, 1000.0f);
        rew = 0.1f * ((-(dn * dn) + -(de * de)) + -(da * da));
    }
    const bool off = (m3 | m4) | m5;
    const bool r_unreach = m1 & off, r_reach = ((!off) & (!m1)) & m2;
    reasons = (r_over ? 1u : 0u) | (r_low ? 2u : 0u) | (r_fast ? 4u : 0u) | (r_slow ? 8u : 0u) | (r_ext ? 16u : 0u) |
              (r_unreach ? 32u : 0u) | (r_reach ? 64u : 0u);
    b |= r_unreach;
    b |= bad_prev;                                               // env_base.py:72-74: flags accumulate until the next reset()
    const bool d = r_reach | done_prev;
    rew = 0.0f + rew;                                            // task_base.py:70-72
    reward_task = rew;                                           // the task's own reward function, before the event term
    rew = rew + (float)(-200 * (int)b + 200 * (int)d);           // event_driven_reward.py:28
    done = d;
    bad = b;
    reward = rew;
}

}  // namespace npf16
