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
#define NPACT8_EXP 0   // timing-only experiment switches (tools/microbench/i8_actor_phases.hip: wrong results: 1 no matrix instructions, 2 no weight stream,
                       // 4 no LayerNorm exchange, 8 no quantiser arithmetic, 16 one-instruction epilogue); 0 in every shipped build
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
              LDS8_PH = LDS8_PM + PART_FLOATS, ACTOR8_LDS_FLOATS = LDS8_PH + PART_FLOATS,
              LDS8_HEAD = LDS8_XF;   // the head's exchange [aircraft][output][lane block] re-uses the x fragments' space (dead after the last layer's k-loop; two barriers in between)
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
#if NPACT8_EXP & 8   // timing only: no quantiser arithmetic (the raw bits go to the fragment slots: same LDS traffic)
    {
        i32x4 *dst = reinterpret_cast<i32x4 *>(frags) + (ks * 3) * 64 + lane;
        (void)ex;
        dst[0] = i32x4{(int)__float_as_uint(y[0]), (int)__float_as_uint(y[1]), (int)__float_as_uint(y[2]), (int)__float_as_uint(y[3])};
        dst[64] = i32x4{(int)__float_as_uint(y[4]), (int)__float_as_uint(y[5]), (int)__float_as_uint(y[6]), (int)__float_as_uint(y[7])};
        dst[128] = i32x4{(int)__float_as_uint(y[8]), (int)__float_as_uint(y[9]), (int)__float_as_uint(y[10]), (int)__float_as_uint(y[11])};
        return;
    }
#endif
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

// Everything below is written for T tiles of 32 aircraft per workgroup that share ONE weight stream: a fragment in registers multiplies the
// B operands of tile 0, then of tile 1, ... (T x the class sums, 1 / T of the L2 traffic per aircraft — the stand-alone kernel at large batches
// would be bound by exactly that traffic: 592 KB per tile and call).  T = 1 is what is instantiated: T = 2 was built, is bit-identical, and was NOT
// faster (n = 262 144: 664 us per call against 633 — a workgroup's time is the latency chain of its phases, not the weight stream; what helps is a
// SECOND workgroup per CU, which needs the LDS block below 80 KB: profiles/r05_actor_i8_tiles.log).  Tile t's LDS block is lds + t * ACTOR8_LDS_FLOATS, its parking area park + t * ACTOR8_PARK_FLOATS.

// LayerNorm over the 128 features of an aircraft held as 8 lane blocks of 16 (accumulator layout), through two LDS exchanges (two barriers for
// all T tiles); optionally a third quantity rides in the first exchange (the recurrent state's |max|: extra_in -> extra_out = the row maximum).
// y and the row exponent of y as the CPU restatement f16_actor_i8.inc::ai8_layernorm computes them.
template <bool EXTRA, int T>
__device__ __forceinline__ void layernorm_acc(const float (&v)[T][16], const float *gp, const float *bp, float gmax, float bmax, float *lds, int blk, int a,
                                              int fbase, float (&y)[T][16], int (&ex)[T], const float (&extra_in)[T], float (&extra_out)[T]) {
#if NPACT8_EXP & 4   // timing only: no LayerNorm exchange (no LDS round trips, no barriers: the lane's own 16 values stand in for the row's 128)
#pragma unroll
    for (int t = 0; t < T; t++) {
        float s = 0.0f, q = 0.0f, m = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r++) s = s + v[t][r];
        const float mean = s * (1.0f / 16.0f);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float d = v[t][r] - mean;
            q = fmaf(d, d, q);
            m = fmaxf(m, fabsf(d));
            y[t][r] = d;
        }
        const float rstd = 1.0f / sqrtf(q * (1.0f / 16.0f) + 1e-5f);
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const float4 qg = *reinterpret_cast<const float4 *>(gp + fbase + 8 * g), qb = *reinterpret_cast<const float4 *>(bp + fbase + 8 * g);
            y[t][4 * g] = fmaf(y[t][4 * g] * rstd, qg.x, qb.x); y[t][4 * g + 1] = fmaf(y[t][4 * g + 1] * rstd, qg.y, qb.y);
            y[t][4 * g + 2] = fmaf(y[t][4 * g + 2] * rstd, qg.z, qb.z); y[t][4 * g + 3] = fmaf(y[t][4 * g + 3] * rstd, qg.w, qb.w);
        }
        ex[t] = exponent_of(fmaf(m * rstd, gmax, bmax) * 1.000001f);
        if constexpr (EXTRA) extra_out[t] = extra_in[t];
    }
    (void)lds; (void)blk; (void)a;
    return;
#endif
#pragma unroll
    for (int t = 0; t < T; t++) {
        float s = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r++) s = s + v[t][r];
        lds[t * ACTOR8_LDS_FLOATS + LDS8_PS + a * 8 + blk] = s;
        if constexpr (EXTRA) lds[t * ACTOR8_LDS_FLOATS + LDS8_PH + a * 8 + blk] = extra_in[t];
    }
    __syncthreads();
    float d[T][16];
#pragma unroll
    for (int t = 0; t < T; t++) {
        const float *part_s = lds + t * ACTOR8_LDS_FLOATS + LDS8_PS, *part_h = lds + t * ACTOR8_LDS_FLOATS + LDS8_PH;
        float total = 0.0f;
        {
            const float4 p0 = *reinterpret_cast<const float4 *>(part_s + a * 8), p1 = *reinterpret_cast<const float4 *>(part_s + a * 8 + 4);
            total = total + p0.x; total = total + p0.y; total = total + p0.z; total = total + p0.w;
            total = total + p1.x; total = total + p1.y; total = total + p1.z; total = total + p1.w;
        }
        if constexpr (EXTRA) {
            const float4 p0 = *reinterpret_cast<const float4 *>(part_h + a * 8), p1 = *reinterpret_cast<const float4 *>(part_h + a * 8 + 4);
            extra_out[t] = fmaxf(fmaxf(fmaxf(p0.x, p0.y), fmaxf(p0.z, p0.w)), fmaxf(fmaxf(p1.x, p1.y), fmaxf(p1.z, p1.w)));
        }
        const float mean = total * (1.0f / 128.0f);
        float q = 0.0f, m = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            d[t][r] = v[t][r] - mean;
            q = fmaf(d[t][r], d[t][r], q);
            m = fmaxf(m, fabsf(d[t][r]));
        }
        lds[t * ACTOR8_LDS_FLOATS + LDS8_PQ + a * 8 + blk] = q;
        lds[t * ACTOR8_LDS_FLOATS + LDS8_PM + a * 8 + blk] = m;
    }
    __syncthreads();
    float gg[16], bb[16];
#pragma unroll
    for (int g = 0; g < 4; g++) {   // this lane's features 8 g + 4 h + t of its wave's 32: four runs of four (tables in LDS)
        const float4 qg = *reinterpret_cast<const float4 *>(gp + fbase + 8 * g), qb = *reinterpret_cast<const float4 *>(bp + fbase + 8 * g);
        gg[4 * g] = qg.x; gg[4 * g + 1] = qg.y; gg[4 * g + 2] = qg.z; gg[4 * g + 3] = qg.w;
        bb[4 * g] = qb.x; bb[4 * g + 1] = qb.y; bb[4 * g + 2] = qb.z; bb[4 * g + 3] = qb.w;
    }
#pragma unroll
    for (int t = 0; t < T; t++) {
        const float *part_q = lds + t * ACTOR8_LDS_FLOATS + LDS8_PQ, *part_m = lds + t * ACTOR8_LDS_FLOATS + LDS8_PM;
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
        for (int r = 0; r < 16; r++) y[t][r] = fmaf(d[t][r] * rstd, gg[r], bb[r]);
        ex[t] = exponent_of(fmaf(mm * rstd, gmax, bmax) * 1.000001f);
    }
}

// The A-operand (weight) fragments of the M-block in flight, [k-step][limb], 64 VGPRs.  They ROTATE: as soon as the products of k-step ks
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

// One M-block (32 output features x 32 aircraft, T tiles) of a quantised Linear layer: KS k-steps of nine limb products into four class
// sums per tile, then the epilogue y = fmaf(fmaf-chain(c0..c3) * 2^(ex - 17), 2^(ew - 18), bias) per accumulator register.
// wf: this M-block's A fragments (already requested); next: the NEXT M-block's fragments [k-step][limb][lane][16 B] in global memory, of
// which k-steps [0, NEXT_KS) are requested into the registers this M-block frees (a KS = 1 block frees slot 0 only: the caller requested
// the follower's k-steps 1..3 up front); xoff: float offset of the B fragments [k-step][limb][lane][16 B] inside a tile's LDS block, or
// `xreg` (KS == 1: the first layer's operand comes from registers); swp / biasp: the layer's scales and biases (LDS tables)
template <int KS, int NEXT_KS, int T>
__device__ __forceinline__ void mblock(WFrags &wf, const unsigned char *next, const float *lds, int xoff, const i32x4 (&xreg)[T][3], const float *swp,
                                       const float *biasp, int fbase, const int (&ex)[T], int lane, float (&y)[T][16]) {
    i32x16 c[T][4];
#pragma unroll
    for (int t = 0; t < T; t++)
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int r = 0; r < 16; r++) c[t][k][r] = 0;
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
        const i32x4 w0 = wf.w[ks][0], w1 = wf.w[ks][1], w2 = wf.w[ks][2], w3 = wf.w[ks][3];
#pragma unroll
        for (int t = 0; t < T; t++) {
            i32x4 x0, x1, x2;
            if constexpr (KS == 1) { x0 = xreg[t][0]; x1 = xreg[t][1]; x2 = xreg[t][2]; }
            else {
                const i32x4 *xl = reinterpret_cast<const i32x4 *>(lds + t * ACTOR8_LDS_FLOATS + xoff) + lane;
                x0 = xl[(ks * 3 + 0) * 64]; x1 = xl[(ks * 3 + 1) * 64]; x2 = xl[(ks * 3 + 2) * 64];
            }
#if NPACT8_EXP & 1   // timing only: no matrix instructions (the operands are kept alive)
            asm volatile("" :: "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(x0), "v"(x1), "v"(x2));
#else
            // consecutive instructions never touch the same accumulator
            c[t][3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w0, x2, c[t][3], 0, 0, 0);
            c[t][2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1, x2, c[t][2], 0, 0, 0);
            c[t][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w2, x2, c[t][1], 0, 0, 0);
            c[t][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w3, x2, c[t][0], 0, 0, 0);
            c[t][3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1, x1, c[t][3], 0, 0, 0);
            c[t][2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w2, x1, c[t][2], 0, 0, 0);
            c[t][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w3, x1, c[t][1], 0, 0, 0);
            c[t][3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w2, x0, c[t][3], 0, 0, 0);
            c[t][2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w3, x0, c[t][2], 0, 0, 0);
#endif
        }
        // vector ALU, scalar ALU, LDS and transcendental instructions may cross; the refill below (VMEM) stays behind the products (MFMA) that read the registers
        __builtin_amdgcn_sched_barrier(0x486);
        if (ks < NEXT_KS) wfrag_load(wf, next, ks, ks, lane);
        __builtin_amdgcn_sched_barrier(0x486);
    }
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const float4 qs = *reinterpret_cast<const float4 *>(swp + fbase + 8 * g), qb = *reinterpret_cast<const float4 *>(biasp + fbase + 8 * g);
        const float sw[4] = {qs.x, qs.y, qs.z, qs.w}, bi[4] = {qb.x, qb.y, qb.z, qb.w};
#pragma unroll
        for (int t = 0; t < T; t++) {
            const float sa = pow2f(ex[t] - 17);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int r = 4 * g + q;
#if NPACT8_EXP & 16   // timing only: one conversion instead of the epilogue
                y[t][r] = (float)(c[t][0][r] + c[t][1][r] + c[t][2][r] + c[t][3][r]) * sa + sw[q] + bi[q];
#else
                // (float)(c0 * 256 + c1) IS fmaf((float)c0, 256, (float)c1): both round the same exact integer (|c0| < 2^19, |c1| < 2^22) once, to nearest even
                float u = (float)(c[t][0][r] * 256 + c[t][1][r]);
                u = fmaf(u, 256.0f, (float)c[t][2][r]);
                u = fmaf(u, 256.0f, (float)c[t][3][r]);
                y[t][r] = fmaf(u * sa, sw[q], bi[q]);
#endif
            }
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

template <int T>
__device__ __forceinline__ void relu_acc(float (&v)[T][16]) {
#pragma unroll
    for (int t = 0; t < T; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) v[t][r] = v[t][r] > 0.0f ? v[t][r] : 0.0f;
}

// T tiles of 32 aircraft, the calling workgroup's waves 0..3 (tid < 256).  xr[t] = the 22 raw observations of this lane's aircraft of tile t;
// hm[t] = the MASKED recurrent state (gru.py:26) of this lane's 16 features (accumulator layout); returns hn (same layout) and action[t] =
// tanh(mu) of (aircraft a, output w) in the lanes with h == 0.  tab: the staged tables (actor8_stage_tables).
// TANH = false: action = mu itself (the policy step's sampled act layer / value head: policy_act_i8_kernel); NOBS < 22: a network on fewer observations
// in the same packed layout (xr[.][j >= NOBS] is not read; the first layer's k-slots of features >= NOBS are zero on both sides)
template <int T, bool TANH = true, int NOBS = OBS>
__device__ __forceinline__ void actor8_body(float *lds, float *park, const float *tab, const float *weights, const float (&xr)[T][OBS], const float (&hm)[T][16],
                                            float (&hn)[T][16], float (&action)[T], unsigned tid) {
    const cw_ptr W = (cw_ptr)(unsigned long long)weights;   // wave-uniform reads: scalar loads
    const unsigned char *frag = reinterpret_cast<const unsigned char *>(weights + FRAG);
    const int lane = (int)(tid & 63u), a = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const int blk = 2 * w + h, fbase = 32 * w + 4 * h;   // this lane's features: fbase + 8 g + t
    i32x4 none[T][3];
#pragma unroll
    for (int t = 0; t < T; t++)
#pragma unroll
        for (int l = 0; l < 3; l++) none[t][l] = i32x4{0, 0, 0, 0};
    float v[T][16], y[T][16], zero[T], dummy[T];
    int ex[T], eh[T];
#pragma unroll
    for (int t = 0; t < T; t++) zero[t] = 0.0f;
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
    i32x4 x1reg[T][3];
#pragma unroll
    for (int t = 0; t < T; t++) {
        float total = 0.0f;
#pragma unroll
        for (int j = 0; j < NOBS; j++) total = total + xr[t][j];
        const float mean = total * (1.0f / (float)NOBS);
        float d[NOBS], q = 0.0f, m = 0.0f;
#pragma unroll
        for (int j = 0; j < NOBS; j++) {
            d[j] = xr[t][j] - mean;
            q = fmaf(d[j], d[j], q);
            m = fmaxf(m, fabsf(d[j]));
        }
        const float rstd = 1.0f / sqrtf(q * (1.0f / (float)NOBS) + 1e-5f);
        ex[t] = exponent_of(fmaf(m * rstd, W[LNMAX + 0], W[LNMAX + 1]) * 1.000001f);
        const float scale = pow2f(XBITS - ex[t]), magic = __uint_as_float(MAGIC_BITS);
        unsigned p[32];
#pragma unroll
        for (int j = 0; j < 32; j++) {
            if (j < NOBS) {
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
            for (int q4 = 0; q4 < 4; q4++) s4[q4] = h ? p[8 * g + 4 + q4] : p[8 * g + q4];
            const unsigned a_lo = __builtin_amdgcn_perm(s4[1], s4[0], 0x05010400u), a_hi = __builtin_amdgcn_perm(s4[1], s4[0], 0x07030602u);
            const unsigned b_lo = __builtin_amdgcn_perm(s4[3], s4[2], 0x05010400u), b_hi = __builtin_amdgcn_perm(s4[3], s4[2], 0x07030602u);
            x1reg[t][0][g] = (int)__builtin_amdgcn_perm(b_lo, a_lo, 0x05040100u);
            x1reg[t][1][g] = (int)__builtin_amdgcn_perm(b_lo, a_lo, 0x07060302u);
            x1reg[t][2][g] = (int)__builtin_amdgcn_perm(b_hi, a_hi, 0x05040100u);
        }
    }
    // base.mlp: Linear(22, 128) + ReLU + LayerNorm
    mblock<1, 1, T>(wf, f_l2, lds, 0, x1reg, tab + T_SW + O_L1, tab + T_BIAS + O_L1, fbase, ex, lane, v);
    relu_acc<T>(v);
    layernorm_acc<false, T>(v, tab + T_LN + 0, tab + T_LN + 128, W[LNMAX + 2], W[LNMAX + 3], lds, blk, a, fbase, y, ex, zero, dummy);
#pragma unroll
    for (int t = 0; t < T; t++) quantise_store(y[t], ex[t], lds + t * ACTOR8_LDS_FLOATS + LDS8_XF, w, lane);
    __syncthreads();
    // Linear(128, 128) + ReLU + LayerNorm; the recurrent state's row maximum rides in the LayerNorm's first exchange
    mblock<4, 4, T>(wf, f_gi + 4 * MB_BYTES_K4, lds, LDS8_XF, none, tab + T_SW + O_L2, tab + T_BIAS + O_L2, fbase, ex, lane, v);
    relu_acc<T>(v);
    float hmax_l[T], hmax[T];
#pragma unroll
    for (int t = 0; t < T; t++) {
        hmax_l[t] = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r++) hmax_l[t] = fmaxf(hmax_l[t], fabsf(hm[t][r]));
    }
    layernorm_acc<true, T>(v, tab + T_LN + 256, tab + T_LN + 384, W[LNMAX + 4], W[LNMAX + 5], lds, blk, a, fbase, y, ex, hmax_l, hmax);
#pragma unroll
    for (int t = 0; t < T; t++) {
        eh[t] = exponent_of(hmax[t]);
        quantise_store(y[t], ex[t], lds + t * ACTOR8_LDS_FLOATS + LDS8_XF, w, lane);
        quantise_store(hm[t], eh[t], lds + t * ACTOR8_LDS_FLOATS + LDS8_HF, w, lane);
        park_store(park + t * ACTOR8_PARK_FLOATS, hm[t], tid);   // (own data: no barrier of its own)
    }
    __syncthreads();
    // rnn: GRU cell.  Gate order of the ARITHMETIC as in torch (r, z, n); evaluated z, r, n so that only one gate vector is in registers beside
    // a matrix phase: z waits in LDS, r is folded into r * gh_n before the last M-block
    {
        float yi[T][16], yh[T][16], g1[T][16];
        mblock<4, 4, T>(wf, f_gh + 4 * MB_BYTES_K4, lds, LDS8_XF, none, tab + T_SW + O_GI + HID, tab + T_BIAS + O_GI + HID, fbase, ex, lane, yi);          // gi_z
        mblock<4, 4, T>(wf, f_gi, lds, LDS8_HF, none, tab + T_SW + O_GH + HID, tab + T_BIAS + O_GH + HID, fbase, eh, lane, yh);                           // gh_z
#pragma unroll
        for (int t = 0; t < T; t++) {
#pragma unroll
            for (int r = 0; r < 16; r++) g1[t][r] = act_sigmoid(yi[t][r] + yh[t][r]);
            park_store(park + t * ACTOR8_PARK_FLOATS + 16 * 256, g1[t], tid);
        }
        mblock<4, 4, T>(wf, f_gh, lds, LDS8_XF, none, tab + T_SW + O_GI, tab + T_BIAS + O_GI, fbase, ex, lane, yi);                                        // gi_r
        mblock<4, 4, T>(wf, f_gh + 8 * MB_BYTES_K4, lds, LDS8_HF, none, tab + T_SW + O_GH, tab + T_BIAS + O_GH, fbase, eh, lane, yh);                       // gh_r
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) g1[t][r] = act_sigmoid(yi[t][r] + yh[t][r]);
        mblock<4, 4, T>(wf, f_gi + 8 * MB_BYTES_K4, lds, LDS8_HF, none, tab + T_SW + O_GH + 2 * HID, tab + T_BIAS + O_GH + 2 * HID, fbase, eh, lane, yh);   // gh_n
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) g1[t][r] = g1[t][r] * yh[t][r];
        mblock<4, 4, T>(wf, f_a1, lds, LDS8_XF, none, tab + T_SW + O_GI + 2 * HID, tab + T_BIAS + O_GI + 2 * HID, fbase, ex, lane, yi);                    // gi_n
#pragma unroll
        for (int t = 0; t < T; t++) {
            float zz[16], hq[16];
            park_load(park + t * ACTOR8_PARK_FLOATS + 16 * 256, zz, tid);
            park_load(park + t * ACTOR8_PARK_FLOATS, hq, tid);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float nn = act_tanh(yi[t][r] + g1[t][r]);
                hn[t][r] = (hq[r] - nn) * zz[r] + nn;
            }
        }
    }
    // rnn.norm (its two barriers also separate the GRU's fragment reads from the next writes)
    layernorm_acc<false, T>(hn, tab + T_LN + 512, tab + T_LN + 640, W[LNMAX + 6], W[LNMAX + 7], lds, blk, a, fbase, y, ex, zero, dummy);
#pragma unroll
    for (int t = 0; t < T; t++) quantise_store(y[t], ex[t], lds + t * ACTOR8_LDS_FLOATS + LDS8_XF, w, lane);
    __syncthreads();
    // act.mlp
    mblock<4, 4, T>(wf, f_a2, lds, LDS8_XF, none, tab + T_SW + O_A1, tab + T_BIAS + O_A1, fbase, ex, lane, v);
    relu_acc<T>(v);
    layernorm_acc<false, T>(v, tab + T_LN + 768, tab + T_LN + 896, W[LNMAX + 8], W[LNMAX + 9], lds, blk, a, fbase, y, ex, zero, dummy);
#pragma unroll
    for (int t = 0; t < T; t++) quantise_store(y[t], ex[t], lds + t * ACTOR8_LDS_FLOATS + LDS8_XF, w, lane);
    __syncthreads();
    mblock<4, 0, T>(wf, f_a2, lds, LDS8_XF, none, tab + T_SW + O_A2, tab + T_BIAS + O_A2, fbase, ex, lane, v);
    relu_acc<T>(v);
    layernorm_acc<false, T>(v, tab + T_LN + 1024, tab + T_LN + 1152, W[LNMAX + 10], W[LNMAX + 11], lds, blk, a, fbase, y, ex, zero, dummy);
    // mu_net: Linear(128, 4) + tanh — per lane block a sequential chain over its 16 features for the four outputs; wave o finishes output o
#pragma unroll
    for (int t = 0; t < T; t++) {
        float p4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float4 hw = *reinterpret_cast<const float4 *>(tab + T_HEAD + (fbase + 8 * g + q) * 4);
                p4[0] = fmaf(hw.x, y[t][4 * g + q], p4[0]);
                p4[1] = fmaf(hw.y, y[t][4 * g + q], p4[1]);
                p4[2] = fmaf(hw.z, y[t][4 * g + q], p4[2]);
                p4[3] = fmaf(hw.w, y[t][4 * g + q], p4[3]);
            }
        float *hd = lds + t * ACTOR8_LDS_FLOATS + LDS8_HEAD;   // [aircraft][output][lane block]
#pragma unroll
        for (int o = 0; o < 4; o++) hd[(a * 4 + o) * 8 + blk] = p4[o];
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < T; t++) {
        const float *hd = lds + t * ACTOR8_LDS_FLOATS + LDS8_HEAD;
        const float4 q0 = *reinterpret_cast<const float4 *>(hd + (a * 4 + w) * 8), q1 = *reinterpret_cast<const float4 *>(hd + (a * 4 + w) * 8 + 4);
        float tot = W[HD_B + w];
        tot = tot + q0.x; tot = tot + q0.y; tot = tot + q0.z; tot = tot + q0.w;
        tot = tot + q1.x; tot = tot + q1.y; tot = tot + q1.z; tot = tot + q1.w;
        action[t] = TANH ? act_tanh(tot) : tot;
    }
}

// the tables -> LDS (TAB_FLOATS floats at `tab`): once per workgroup, by `threads` threads; the caller's barrier makes them visible
__device__ __forceinline__ void actor8_stage_tables(float *tab, const float *weights, unsigned tid, unsigned threads) {
    const float4 *src = reinterpret_cast<const float4 *>(weights + TAB0);
    for (unsigned i = tid; i < TAB_FLOATS / 4; i += threads) reinterpret_cast<float4 *>(tab)[i] = src[i];
}

// LDS of a stand-alone workgroup that runs T tiles: the tiles' blocks, their parking areas, the tables
template <int T>
constexpr int actor8_tile_lds_floats() { return T * (ACTOR8_LDS_FLOATS + ACTOR8_PARK_FLOATS) + TAB_FLOATS; }

// tiles T * group .. T * group + T - 1 (32 aircraft each) through global memory
template <int T>
__device__ __forceinline__ void actor8_tiles(float *lds, const float *weights, long long n, const float *obs, const float *h_in, const float *mask, float *act,
                                             float *h_out, long long group, unsigned tid) {
    const int lane = (int)(tid & 63u), a = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    float hm[T][16], xr[T][OBS], hn[T][16], action[T];
#pragma unroll
    for (int t = 0; t < T; t++) {
        const long long i = (group * T + t) * 32 + a;
        const long long ic = i < n ? i : n - 1;   // aircraft beyond the batch shadow its last one; nothing of them is stored
        const float mk = mask[ic];
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const float4 q = *reinterpret_cast<const float4 *>(h_in + ic * HID + 32 * w + 4 * h + 8 * g);
            hm[t][4 * g] = q.x * mk; hm[t][4 * g + 1] = q.y * mk; hm[t][4 * g + 2] = q.z * mk; hm[t][4 * g + 3] = q.w * mk;
        }
#pragma unroll
        for (int j = 0; j < OBS; j++) xr[t][j] = obs[ic * OBS + j];
    }
    float *park = lds + T * ACTOR8_LDS_FLOATS, *tab = park + T * ACTOR8_PARK_FLOATS;
    actor8_stage_tables(tab, weights, tid, 256u);
    __syncthreads();
    actor8_body<T>(lds, park, tab, weights, xr, hm, hn, action, tid);
#pragma unroll
    for (int t = 0; t < T; t++) {
        const long long i = (group * T + t) * 32 + a;
        if (i < n) {
            if (h == 0) act[i * 4 + w] = action[t];
#pragma unroll
            for (int g = 0; g < 4; g++)
                *reinterpret_cast<float4 *>(h_out + i * HID + 32 * w + 4 * h + 8 * g) = make_float4(hn[t][4 * g], hn[t][4 * g + 1], hn[t][4 * g + 2], hn[t][4 * g + 3]);
        }
    }
}

}  // namespace npact8
