// np_actor_i8.h — PlanningEnv's frozen low-level controller, SECOND numerics spec: block fixed point on the gfx950 i8 matrix pipe.
//
// Reference: PPOActor.forward(obs, rnn_states, masks, deterministic=True) (algorithms/ppo/ppo_actor.py:38-64, configuration
// envs/planning_env.py:18-29) — the network of np_actor.h.  What changes: the six Linear layers with N >= 128 (22 -> 128, 128 -> 128 x 3,
// the GRU's two 128 -> 384) run as v_mfma_i32_32x32x32_i8 on integer limbs instead of 1 184 dependent K = 1 fp32 MFMA steps, and every
// activation stays in that instruction's ACCUMULATOR layout from the first layer to the head: D = W . X^T, wave w of a 32-aircraft tile owns
// output features 32 w .. 32 w + 31, lane (a = lane & 31, h = lane >> 5) holds aircraft a and, in accumulator register r = 4 g + t, feature
//         f = 32 w + 8 g + 4 h + t.
// A layer's output, quantised in registers, IS the B-operand fragment of the next layer's k-step w (the k-slots of a 32-wide step are
// assigned in exactly that order; the weights are packed to match at load time), so the only data that moves between layers are 3 KB of
// limb bytes per wave through LDS and the LayerNorm partial sums.
//
// Numerics spec = the CPU restatement f16_actor_i8.inc, statement by statement (integer class sums are exact; the conversions, the fused multiply-adds
// that combine them, the LayerNorm summation order, the row exponent and the quantiser are spelled out there); the tests hold the kernels to
// it bit for bit, and hold the spec to the reference's recordings (actions 5e-6, closed loop 4e-5 — as the fp32 spec).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "np_actor.h"

#ifndef NPACT8_EXP
#define NPACT8_EXP 0   // timing-only experiment switches (tools/microbench/i8_actor_phases.hip: wrong results); 0 in every shipped build
#endif

namespace npact8 {
using namespace npact;

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

// ---- the packed weight buffer (floats): [0, TOTAL) the fp32 layout of np_actor.h, then a block of TABLES in the order the kernel wants
// them (staged into LDS once per workgroup: reading them from global memory inside the call would queue behind the weight stream — every
// s_waitcnt vmcnt for an epilogue constant drains the prefetched fragments of the NEXT M-block, measured 6 us of a 25 us call), then the limb
// fragments -------------------------------------------------------------------------------------------------------------------------------
enum : int {
    TAB0 = TOTAL,
    T_SW = 0,                        // per-output scales 2^(ew - 18): L1 128 | L2 128 | GI 384 | GH 384 | A1 128 | A2 128
    T_BIAS = 1280,                   // the biases in the same order
    T_LN = 2560,                     // LayerNorm 1..5: gamma 128 | beta 128 each
    T_HEAD = 3840,                   // mu_net weights [feature][4], then its 4 biases
    T_LNMAX = 4356,                  // [6][2]: max |gamma|, max |beta| of LayerNorm 0..5
    TAB_FLOATS = 4368,
    O_L1 = 0, O_L2 = 128, O_GI = 256, O_GH = 640, O_A1 = 1024, O_A2 = 1152,   // offsets inside T_SW / T_BIAS
    LNMAX = TAB0 + T_LNMAX,
    FRAG = TAB0 + TAB_FLOATS,        // byte fragments from here (16-byte aligned)
    FRAG_BYTES = 1024,               // one A-operand fragment: 64 lanes x 16 bytes
    MB_BYTES_K4 = 4 * 4 * FRAG_BYTES,   // an M-block of a K = 128 layer: [k-step][limb] fragments
    FR_L1 = 0, FR_L2 = FR_L1 + 4 * 4 * FRAG_BYTES, FR_GI = FR_L2 + 4 * MB_BYTES_K4, FR_GH = FR_GI + 12 * MB_BYTES_K4, FR_A1 = FR_GH + 12 * MB_BYTES_K4,
    FR_A2 = FR_A1 + 4 * MB_BYTES_K4, FR_END = FR_A2 + 4 * MB_BYTES_K4,
    TOTAL_I8 = FRAG + FR_END / 4
};
static_assert(TOTAL % 4 == 0 && TAB_FLOATS % 4 == 0 && FR_END == 592 * 1024 && TOTAL_I8 == 309312, "packed i8 actor layout (np_actor_pack_i8, neuralplane_amd/actor.py)");

constexpr unsigned MAGIC_BITS = 0x4B400000u;   // 1.5 * 2^23
constexpr int XBITS = 22, WBITS = 29;

// LDS (floats): two sets of B-operand fragments [k-step][limb][lane][16 B] (x and the recurrent state), the LayerNorm / head exchange
constexpr int XF_FLOATS = 4 * 3 * FRAG_BYTES / 4;          // 3 072 floats = 12 KB
constexpr int PART_FLOATS = 32 * 8;                        // [aircraft][lane block]
constexpr int LDS8_XF = 0, LDS8_HF = XF_FLOATS, LDS8_PS = 2 * XF_FLOATS, LDS8_PQ = LDS8_PS + PART_FLOATS, LDS8_PM = LDS8_PQ + PART_FLOATS,
              LDS8_PH = LDS8_PM + PART_FLOATS, LDS8_HEAD = LDS8_PH + PART_FLOATS, ACTOR8_LDS_FLOATS = LDS8_HEAD + 32 * 8 * 4;
// a second LDS region, wherever the caller has room: the masked recurrent state and the z gate wait here while the GRU's matrix phases run (the
// registers they would hold — with 64 of weight fragments, 64 of class sums and 12 of operands in flight — are what a 256-register wave lacks)
constexpr int ACTOR8_PARK_FLOATS = 2 * 16 * 256;   // [2][4 float4][256 threads]: 32 KB
constexpr int ACTOR8_BARRIERS = 15;   // __syncthreads() executed by actor8_body (straight-line code)

__device__ __forceinline__ int exponent_of(float b) {   // e with |b| < 2^e, clamped below
    const int e = (int)((__float_as_uint(b) >> 23) & 255u) - 126;
    return e < -100 ? -100 : e;
}
__device__ __forceinline__ float pow2f(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }

// the 16 values of this lane -> limb bytes: P = (q + 0x808080) ^ 0x808080 with q = round-half-even(x * 2^(22 - ex)); the three limb planes
// as B-operand dwords (dword g = the limb of registers 4 g .. 4 g + 3) -> LDS fragment `ks` of `frags`
__device__ __forceinline__ void quantise_store(const float (&y)[16], int ex, float *frags, int ks, int lane) {
    const float scale = pow2f(XBITS - ex), magic = __uint_as_float(MAGIC_BITS);
    unsigned p[16];
#pragma unroll
    for (int r = 0; r < 16; r++) p[r] = (__float_as_uint(fmaf(y[r], scale, magic)) + (0x808080u - MAGIC_BITS)) ^ 0x808080u;
    i32x4 l0, l1, l2;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const unsigned a_lo = __builtin_amdgcn_perm(p[4 * g + 1], p[4 * g], 0x05010400u), a_hi = __builtin_amdgcn_perm(p[4 * g + 1], p[4 * g], 0x07030602u);
        const unsigned b_lo = __builtin_amdgcn_perm(p[4 * g + 3], p[4 * g + 2], 0x05010400u), b_hi = __builtin_amdgcn_perm(p[4 * g + 3], p[4 * g + 2], 0x07030602u);
        l0[g] = (int)__builtin_amdgcn_perm(b_lo, a_lo, 0x05040100u);
        l1[g] = (int)__builtin_amdgcn_perm(b_lo, a_lo, 0x07060302u);
        l2[g] = (int)__builtin_amdgcn_perm(b_hi, a_hi, 0x05040100u);
    }
    i32x4 *dst = reinterpret_cast<i32x4 *>(frags) + (ks * 3) * 64 + lane;
    dst[0] = l0;
    dst[64] = l1;
    dst[128] = l2;
}

// LayerNorm over the 128 features of an aircraft held as 8 lane blocks of 16 (accumulator layout), through two LDS exchanges; optionally a
// third quantity rides in the first exchange (the recurrent state's |max|: `extra_in` -> `extra_out` = the row maximum).
// Returns y and the row exponent of y (the CPU restatement f16_actor_i8.inc::ai8_layernorm).
template <bool EXTRA>
__device__ __forceinline__ int layernorm_acc(const float (&v)[16], const float *gp, const float *bp, float gmax, float bmax, float *lds, int blk, int a,
                                             int fbase, float (&y)[16], float extra_in, float &extra_out) {
    float *part_s = lds + LDS8_PS, *part_q = lds + LDS8_PQ, *part_m = lds + LDS8_PM, *part_h = lds + LDS8_PH;
    float gg[16], bb[16];
#pragma unroll
    for (int g = 0; g < 4; g++) {   // this lane's features 8 g + 4 h + t of its wave's 32: four runs of four
        const float4 qg = *reinterpret_cast<const float4 *>(gp + fbase + 8 * g), qb = *reinterpret_cast<const float4 *>(bp + fbase + 8 * g);
        gg[4 * g] = qg.x; gg[4 * g + 1] = qg.y; gg[4 * g + 2] = qg.z; gg[4 * g + 3] = qg.w;
        bb[4 * g] = qb.x; bb[4 * g + 1] = qb.y; bb[4 * g + 2] = qb.z; bb[4 * g + 3] = qb.w;
    }
    float s = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; r++) s = s + v[r];
    part_s[a * 8 + blk] = s;
    if constexpr (EXTRA) part_h[a * 8 + blk] = extra_in;
    __syncthreads();
    float total = 0.0f;
    {
        const float4 p0 = *reinterpret_cast<const float4 *>(part_s + a * 8), p1 = *reinterpret_cast<const float4 *>(part_s + a * 8 + 4);
        total = total + p0.x; total = total + p0.y; total = total + p0.z; total = total + p0.w;
        total = total + p1.x; total = total + p1.y; total = total + p1.z; total = total + p1.w;
    }
    if constexpr (EXTRA) {
        const float4 p0 = *reinterpret_cast<const float4 *>(part_h + a * 8), p1 = *reinterpret_cast<const float4 *>(part_h + a * 8 + 4);
        extra_out = fmaxf(fmaxf(fmaxf(p0.x, p0.y), fmaxf(p0.z, p0.w)), fmaxf(fmaxf(p1.x, p1.y), fmaxf(p1.z, p1.w)));
    }
    const float mean = total * (1.0f / 128.0f);
    float d[16], q = 0.0f, m = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        d[r] = v[r] - mean;
        q = fmaf(d[r], d[r], q);
        m = fmaxf(m, fabsf(d[r]));
    }
    part_q[a * 8 + blk] = q;
    part_m[a * 8 + blk] = m;
    __syncthreads();
    float qt = 0.0f, mm;
    {
        const float4 p0 = *reinterpret_cast<const float4 *>(part_q + a * 8), p1 = *reinterpret_cast<const float4 *>(part_q + a * 8 + 4);
        qt = qt + p0.x; qt = qt + p0.y; qt = qt + p0.z; qt = qt + p0.w;
        qt = qt + p1.x; qt = qt + p1.y; qt = qt + p1.z; qt = qt + p1.w;
        const float4 m0 = *reinterpret_cast<const float4 *>(part_m + a * 8), m1 = *reinterpret_cast<const float4 *>(part_m + a * 8 + 4);
        mm = fmaxf(fmaxf(fmaxf(m0.x, m0.y), fmaxf(m0.z, m0.w)), fmaxf(fmaxf(m1.x, m1.y), fmaxf(m1.z, m1.w)));
    }
    const float rstd = 1.0f / sqrtf(qt * (1.0f / 128.0f) + 1e-5f);
#pragma unroll
    for (int r = 0; r < 16; r++) y[r] = fmaf(d[r] * rstd, gg[r], bb[r]);
    return exponent_of(fmaf(mm * rstd, gmax, bmax) * 1.000001f);
}

// The A-operand (weight) fragments of the M-block in flight, [k-step][limb], 64 VGPRs.  They ROTATE: as soon as the nine products of k-step ks
// have been issued, the same registers receive k-step ks of the NEXT M-block — every fragment is requested a whole M-block (>= 1 150 cycles
// of matrix pipe + its epilogue) before it is used, which is what the L2 round trip needs, at the register cost of one M-block.
struct WFrags {
    i32x4 w[4][4];
};
__device__ __forceinline__ void wfrag_load(WFrags &f, const unsigned char *mb_frags, int ks, int slot_ks, int lane) {
#if NPACT8_EXP & 2   // timing only: no weight stream
    (void)mb_frags; (void)ks; (void)lane;
#pragma unroll
    for (int l = 0; l < 4; l++) f.w[slot_ks][l] = i32x4{1, 2, 3, 4};
#else
    const i32x4 *wl = reinterpret_cast<const i32x4 *>(mb_frags) + lane;
#pragma unroll
    for (int l = 0; l < 4; l++) f.w[slot_ks][l] = wl[(ks * 4 + l) * 64];
#endif
}

// One M-block (32 output features x 32 aircraft) of a quantised Linear layer: KS k-steps of nine limb products into four class sums,
// then the epilogue y = fmaf(fmaf-chain(c0..c3) * 2^(ex - 17), 2^(ew - 18), bias) per accumulator register.
// wf: this M-block's A fragments (already requested); next: the NEXT M-block's fragments [k-step][limb][lane][16 B] in global memory, of
// which k-steps [0, NEXT_KS) are requested into the registers this M-block frees (a KS = 1 block frees slot 0 only: the caller requested
// the follower's k-steps 1..3 up front); xfrag: the B fragments [k-step][limb][lane][16 B] (LDS), or `xreg` (KS == 1: the first layer's
// operand comes from registers)
template <int KS, int NEXT_KS>
__device__ __forceinline__ void mblock(WFrags &wf, const unsigned char *next, const float *xfrag, const i32x4 (&xreg)[3], const float *swp, const float *biasp,
                                       int fbase, int ex, int lane, float (&y)[16]) {
    i32x16 c0, c1, c2, c3;
#pragma unroll
    for (int r = 0; r < 16; r++) { c0[r] = 0; c1[r] = 0; c2[r] = 0; c3[r] = 0; }
    const i32x4 *xl = reinterpret_cast<const i32x4 *>(xfrag) + lane;
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
        i32x4 x0, x1, x2;
        if constexpr (KS == 1) { x0 = xreg[0]; x1 = xreg[1]; x2 = xreg[2]; }
        else { x0 = xl[(ks * 3 + 0) * 64]; x1 = xl[(ks * 3 + 1) * 64]; x2 = xl[(ks * 3 + 2) * 64]; }
        const i32x4 w0 = wf.w[ks][0], w1 = wf.w[ks][1], w2 = wf.w[ks][2], w3 = wf.w[ks][3];
#if NPACT8_EXP & 1   // timing only: no matrix instructions (the operands are kept alive)
        asm volatile("" :: "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(x0), "v"(x1), "v"(x2));
#else
        // consecutive instructions never touch the same accumulator
        c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(w0, x2, c3, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1, x2, c2, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(w2, x2, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(w3, x2, c0, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1, x1, c3, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(w2, x1, c2, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(w3, x1, c1, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(w2, x0, c3, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(w3, x0, c2, 0, 0, 0);
#endif
        __builtin_amdgcn_sched_barrier(0x486);   // vector ALU, scalar ALU, LDS and transcendental instructions may cross; the refill below (VMEM) stays behind the products (MFMA) that read the registers
        if (ks < NEXT_KS) wfrag_load(wf, next, ks, ks, lane);
        __builtin_amdgcn_sched_barrier(0x486);
    }
    const float sa = pow2f(ex - 17);
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const float4 qs = *reinterpret_cast<const float4 *>(swp + fbase + 8 * g), qb = *reinterpret_cast<const float4 *>(biasp + fbase + 8 * g);
        const float sw[4] = {qs.x, qs.y, qs.z, qs.w}, bi[4] = {qb.x, qb.y, qb.z, qb.w};
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int r = 4 * g + t;
#if NPACT8_EXP & 16   // timing only: one conversion instead of the epilogue
            y[r] = (float)(c0[r] + c1[r] + c2[r] + c3[r]) * sa + sw[t] + bi[t];
#else
            float u = fmaf((float)c0[r], 256.0f, (float)c1[r]);
            u = fmaf(u, 256.0f, (float)c2[r]);
            u = fmaf(u, 256.0f, (float)c3[r]);
            y[r] = fmaf(u * sa, sw[t], bi[t]);
#endif
        }
    }
    __builtin_amdgcn_sched_barrier(0x486);
}

__device__ __forceinline__ void park_store(float *park, const float (&v)[16], unsigned tid) {
#pragma unroll
    for (int g = 0; g < 4; g++) reinterpret_cast<float4 *>(park)[g * 256 + (tid & 255u)] = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
}
__device__ __forceinline__ void park_load(const float *park, float (&v)[16], unsigned tid) {
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const float4 q = reinterpret_cast<const float4 *>(park)[g * 256 + (tid & 255u)];
        v[4 * g] = q.x; v[4 * g + 1] = q.y; v[4 * g + 2] = q.z; v[4 * g + 3] = q.w;
    }
}

__device__ __forceinline__ void relu_acc(float (&v)[16]) {
#pragma unroll
    for (int r = 0; r < 16; r++) v[r] = v[r] > 0.0f ? v[r] : 0.0f;
}

// One 32-aircraft tile, the calling workgroup's waves 0..3 (tid < 256).  xr = the 22 raw observations of this lane's aircraft; hm = the
// MASKED recurrent state (gru.py:26) of this lane's 16 features (accumulator layout); returns hn (same layout) and `action` = tanh(mu) of
// (aircraft a, output w) in the lanes with h == 0.
__device__ __forceinline__ void actor8_body(float *lds, float *park, const float *tab, const float *weights, const float (&xr)[OBS], const float (&hm)[16], float (&hn)[16], float &action,
                                            unsigned tid) {
    const cw_ptr W = (cw_ptr)(unsigned long long)weights;   // wave-uniform reads: scalar loads
    const unsigned char *frag = reinterpret_cast<const unsigned char *>(weights + FRAG);
    const int lane = (int)(tid & 63u), a = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const int blk = 2 * w + h, fbase = 32 * w + 4 * h;   // this lane's features: fbase + 8 g + t
    float *xf = lds + LDS8_XF, *hf = lds + LDS8_HF;
    const i32x4 none[3] = {};
    float v[16], y[16], dummy = 0.0f;
    int ex;
    // the weight stream starts before anything else: the first layer's single k-step and k-steps 1..3 of the second layer's M-block
    const unsigned char *f_l1 = frag + FR_L1 + w * 4 * FRAG_BYTES, *f_l2 = frag + FR_L2 + w * MB_BYTES_K4, *f_a1 = frag + FR_A1 + w * MB_BYTES_K4,
                        *f_a2 = frag + FR_A2 + w * MB_BYTES_K4, *f_gi = frag + FR_GI + w * MB_BYTES_K4, *f_gh = frag + FR_GH + w * MB_BYTES_K4;
    WFrags wf;
    wfrag_load(wf, f_l1, 0, 0, lane);
#pragma unroll
    for (int ks = 1; ks < 4; ks++) wfrag_load(wf, f_l2, ks, ks, lane);
    __builtin_amdgcn_sched_barrier(0);

    // base.feature_norm over the 22 observations: every lane computes its aircraft's (no exchange), quantises, and picks the 16 k-slots of
    // its half: slot e <-> feature 8 (e >> 2) + 4 h + (e & 3); slots of features >= 22 are zero (as the packed weights are)
    NPACT_STAMP(0);
    i32x4 x1reg[3];
    {
        float total = 0.0f;
#pragma unroll
        for (int j = 0; j < OBS; j++) total = total + xr[j];
        const float mean = total * (1.0f / (float)OBS);
        float d[OBS], q = 0.0f, m = 0.0f;
#pragma unroll
        for (int j = 0; j < OBS; j++) {
            d[j] = xr[j] - mean;
            q = fmaf(d[j], d[j], q);
            m = fmaxf(m, fabsf(d[j]));
        }
        const float rstd = 1.0f / sqrtf(q * (1.0f / (float)OBS) + 1e-5f);
        ex = exponent_of(fmaf(m * rstd, W[LNMAX + 0], W[LNMAX + 1]) * 1.000001f);
        const float scale = pow2f(XBITS - ex), magic = __uint_as_float(MAGIC_BITS);
        unsigned p[32];
#pragma unroll
        for (int j = 0; j < 32; j++) {
            if (j < OBS) {
                const float yy = fmaf(d[j] * rstd, W[LN0_G + j], W[LN0_B + j]);
                p[j] = (__float_as_uint(fmaf(yy, scale, magic)) + (0x808080u - MAGIC_BITS)) ^ 0x808080u;
            } else {
                p[j] = 0u;
            }
        }
#pragma unroll
        for (int g = 0; g < 4; g++) {
            unsigned s4[4];
#pragma unroll
            for (int t = 0; t < 4; t++) s4[t] = h ? p[8 * g + 4 + t] : p[8 * g + t];
            const unsigned a_lo = __builtin_amdgcn_perm(s4[1], s4[0], 0x05010400u), a_hi = __builtin_amdgcn_perm(s4[1], s4[0], 0x07030602u);
            const unsigned b_lo = __builtin_amdgcn_perm(s4[3], s4[2], 0x05010400u), b_hi = __builtin_amdgcn_perm(s4[3], s4[2], 0x07030602u);
            x1reg[0][g] = (int)__builtin_amdgcn_perm(b_lo, a_lo, 0x05040100u);
            x1reg[1][g] = (int)__builtin_amdgcn_perm(b_lo, a_lo, 0x07060302u);
            x1reg[2][g] = (int)__builtin_amdgcn_perm(b_hi, a_hi, 0x05040100u);
        }
    }
    NPACT_STAMP(1);
    // base.mlp: Linear(22, 128) + ReLU + LayerNorm
    mblock<1, 1>(wf, f_l2, nullptr, x1reg, tab + T_SW + O_L1, tab + T_BIAS + O_L1, fbase, ex, lane, v);
    relu_acc(v);
    NPACT_STAMP(2);
    ex = layernorm_acc<false>(v, tab + T_LN + 0, tab + T_LN + 128, W[LNMAX + 2], W[LNMAX + 3], lds, blk, a, fbase, y, 0.0f, dummy);
    quantise_store(y, ex, xf, w, lane);
    __syncthreads();
    NPACT_STAMP(3);
    // Linear(128, 128) + ReLU + LayerNorm; the recurrent state's row maximum rides in the LayerNorm's first exchange
    mblock<4, 4>(wf, f_gi + 4 * MB_BYTES_K4, xf, none, tab + T_SW + O_L2, tab + T_BIAS + O_L2, fbase, ex, lane, v);
    relu_acc(v);
    NPACT_STAMP(4);
    float hmax_l = 0.0f, hmax;
#pragma unroll
    for (int r = 0; r < 16; r++) hmax_l = fmaxf(hmax_l, fabsf(hm[r]));
    ex = layernorm_acc<true>(v, tab + T_LN + 256, tab + T_LN + 384, W[LNMAX + 4], W[LNMAX + 5], lds, blk, a, fbase, y, hmax_l, hmax);
    const int eh = exponent_of(hmax);
    quantise_store(y, ex, xf, w, lane);
    quantise_store(hm, eh, hf, w, lane);
    park_store(park, hm, tid);   // (own data: no barrier of its own)
    __syncthreads();
    NPACT_STAMP(5);
    // rnn: GRU cell.  Gate order of the ARITHMETIC as in torch (r, z, n); evaluated z, r, n so that only one gate vector is in registers beside
    // a matrix phase: z waits in LDS, r is folded into r * gh_n before the last M-block
    {
        float yi[16], yh[16], t[16];
        mblock<4, 4>(wf, f_gh + 4 * MB_BYTES_K4, xf, none, tab + T_SW + O_GI + HID, tab + T_BIAS + O_GI + HID, fbase, ex, lane, yi);          // gi_z
        mblock<4, 4>(wf, f_gi, hf, none, tab + T_SW + O_GH + HID, tab + T_BIAS + O_GH + HID, fbase, eh, lane, yh);                           // gh_z
#pragma unroll
        for (int r = 0; r < 16; r++) t[r] = act_sigmoid(yi[r] + yh[r]);
        park_store(park + 16 * 256, t, tid);
        mblock<4, 4>(wf, f_gh, xf, none, tab + T_SW + O_GI, tab + T_BIAS + O_GI, fbase, ex, lane, yi);                                        // gi_r
        mblock<4, 4>(wf, f_gh + 8 * MB_BYTES_K4, hf, none, tab + T_SW + O_GH, tab + T_BIAS + O_GH, fbase, eh, lane, yh);                       // gh_r
#pragma unroll
        for (int r = 0; r < 16; r++) t[r] = act_sigmoid(yi[r] + yh[r]);
        mblock<4, 4>(wf, f_gi + 8 * MB_BYTES_K4, hf, none, tab + T_SW + O_GH + 2 * HID, tab + T_BIAS + O_GH + 2 * HID, fbase, eh, lane, yh);   // gh_n
#pragma unroll
        for (int r = 0; r < 16; r++) t[r] = t[r] * yh[r];
        mblock<4, 4>(wf, f_a1, xf, none, tab + T_SW + O_GI + 2 * HID, tab + T_BIAS + O_GI + 2 * HID, fbase, ex, lane, yi);                    // gi_n
        float zz[16], hq[16];
        park_load(park + 16 * 256, zz, tid);
        park_load(park, hq, tid);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float nn = act_tanh(yi[r] + t[r]);
            hn[r] = (hq[r] - nn) * zz[r] + nn;
        }
    }
    NPACT_STAMP(12);
    // rnn.norm (its two barriers also separate the GRU's fragment reads from the next writes)
    ex = layernorm_acc<false>(hn, tab + T_LN + 512, tab + T_LN + 640, W[LNMAX + 6], W[LNMAX + 7], lds, blk, a, fbase, y, 0.0f, dummy);
    quantise_store(y, ex, xf, w, lane);
    __syncthreads();
    NPACT_STAMP(13);
    // act.mlp
    mblock<4, 4>(wf, f_a2, xf, none, tab + T_SW + O_A1, tab + T_BIAS + O_A1, fbase, ex, lane, v);
    relu_acc(v);
    NPACT_STAMP(14);
    ex = layernorm_acc<false>(v, tab + T_LN + 768, tab + T_LN + 896, W[LNMAX + 8], W[LNMAX + 9], lds, blk, a, fbase, y, 0.0f, dummy);
    quantise_store(y, ex, xf, w, lane);
    __syncthreads();
    NPACT_STAMP(15);
    mblock<4, 0>(wf, f_a2, xf, none, tab + T_SW + O_A2, tab + T_BIAS + O_A2, fbase, ex, lane, v);
    relu_acc(v);
    NPACT_STAMP(16);
    (void)layernorm_acc<false>(v, tab + T_LN + 1024, tab + T_LN + 1152, W[LNMAX + 10], W[LNMAX + 11], lds, blk, a, fbase, y, 0.0f, dummy);
    NPACT_STAMP(17);
    // mu_net: Linear(128, 4) + tanh — per lane block a sequential chain over its 16 features for the four outputs; wave o finishes output o
    {
        float p4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const float4 hw = *reinterpret_cast<const float4 *>(tab + T_HEAD + (fbase + 8 * g + t) * 4);
                p4[0] = fmaf(hw.x, y[4 * g + t], p4[0]);
                p4[1] = fmaf(hw.y, y[4 * g + t], p4[1]);
                p4[2] = fmaf(hw.z, y[4 * g + t], p4[2]);
                p4[3] = fmaf(hw.w, y[4 * g + t], p4[3]);
            }
        float *hd = lds + LDS8_HEAD;   // [aircraft][output][lane block]
#pragma unroll
        for (int o = 0; o < 4; o++) hd[(a * 4 + o) * 8 + blk] = p4[o];
        __syncthreads();
        const float4 q0 = *reinterpret_cast<const float4 *>(hd + (a * 4 + w) * 8), q1 = *reinterpret_cast<const float4 *>(hd + (a * 4 + w) * 8 + 4);
        float tot = W[HD_B + w];
        tot = tot + q0.x; tot = tot + q0.y; tot = tot + q0.z; tot = tot + q0.w;
        tot = tot + q1.x; tot = tot + q1.y; tot = tot + q1.z; tot = tot + q1.w;
        action = act_tanh(tot);
    }
    NPACT_STAMP(18);
}

// the tables -> LDS (TAB_FLOATS floats at `tab`): once per workgroup, by `threads` threads; the caller's barrier makes them visible
__device__ __forceinline__ void actor8_stage_tables(float *tab, const float *weights, unsigned tid, unsigned threads) {
    const float4 *src = reinterpret_cast<const float4 *>(weights + TAB0);
    for (unsigned i = tid; i < TAB_FLOATS / 4; i += threads) reinterpret_cast<float4 *>(tab)[i] = src[i];
}

// tile `tile` = aircraft [32 tile, 32 tile + 32) through global memory
__device__ __forceinline__ void actor8_tile(float *lds, const float *weights, long long n, const float *obs, const float *h_in, const float *mask, float *act,
                                            float *h_out, long long tile, unsigned tid) {
    const int lane = (int)(tid & 63u), a = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const long long i = tile * 32 + a;
    const bool valid = i < n;
    const long long ic = valid ? i : n - 1;
    const float mk = mask[ic];
    float hm[16], xr[OBS], hn[16], action;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const float4 q = *reinterpret_cast<const float4 *>(h_in + ic * HID + 32 * w + 4 * h + 8 * g);
        hm[4 * g] = q.x * mk; hm[4 * g + 1] = q.y * mk; hm[4 * g + 2] = q.z * mk; hm[4 * g + 3] = q.w * mk;
    }
#pragma unroll
    for (int j = 0; j < OBS; j++) xr[j] = obs[ic * OBS + j];
    actor8_stage_tables(lds + ACTOR8_LDS_FLOATS + ACTOR8_PARK_FLOATS, weights, tid, 256u);
    __syncthreads();
    actor8_body(lds, lds + ACTOR8_LDS_FLOATS, lds + ACTOR8_LDS_FLOATS + ACTOR8_PARK_FLOATS, weights, xr, hm, hn, action, tid);
    if (valid && h == 0) act[i * 4 + w] = action;
    if (valid) {
#pragma unroll
        for (int g = 0; g < 4; g++)
            *reinterpret_cast<float4 *>(h_out + i * HID + 32 * w + 4 * h + 8 * g) = make_float4(hn[4 * g], hn[4 * g + 1], hn[4 * g + 2], hn[4 * g + 3]);
    }
}

}  // namespace npact8
