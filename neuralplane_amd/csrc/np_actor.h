// np_actor.h — PlanningEnv's frozen low-level controller as one fused gfx950 kernel.
//
// Reference: PPOActor.forward(obs, rnn_states, masks, deterministic=True) (algorithms/ppo/ppo_actor.py:38-64) in the configuration
// envs/planning_env.py:18-29 builds: LayerNorm(22) -> [Linear, ReLU, LayerNorm] x 2 (128) -> GRU(128) -> LayerNorm -> [Linear, ReLU,
// LayerNorm] x 2 (128) -> Linear(128, 4) -> tanh  (algorithms/utils/{mlp,gru,act,distributions}.py).  The reference runs it 50
// times per PlanningEnv.step as ~15 small torch kernels; here one launch per call, activations never leave the CU.
//
// Two kernels, the same arithmetic.  The shipped one (actor_forward_mfma_kernel, second half of this file) runs the 128-wide
// layers on the matrix cores: 4 waves per 64-row tile, each a chain of K = 1 fp32 MFMAs per layer.  The first half is the
// vector-FMA formulation it replaced (NPACT_MFMA=0: 8 waves per tile, wave w computes features [16w, 16w+16) with the weights
// travelling through the scalar unit into v_pk_fma_f32 SGPR operands) — kept as the A/B reference.  In both, activations live in
// LDS as [feature][row] matrices, LayerNorm statistics are reduced across the waves through LDS, 151 K fused multiply-adds per
// row and call.
//
// Numerics spec: ordered fmaf chains (bias first, k ascending), LayerNorm sums in blocks of 16 features added
// in order, explicit fp32 exp — the tests hold this kernel bit-exact to a scalar CPU restatement of the same spec, and within
// ~1e-5 of the reference's ATen kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include "np_actor_asm.inc"
#include "np_actor_mfma_asm.inc"
#include "np_actor_mfma16_asm.inc"
#ifndef NPACT_MFMA
#define NPACT_MFMA 1  // 1: matrix-core kernel (4 waves per tile); 0: the vector-FMA kernel with the scalar weight stream (8 waves per tile)
#endif
#ifndef NPACT_PRIO
#define NPACT_PRIO 0  // > 0: wave priority outside the MFMA chains of the 32-row tile (the chains themselves run at 0)
#endif
#ifndef NPACT_EXP
#define NPACT_EXP 0  // timing-only experiment switches (tools/microbench/README.md); 0 in every shipped build
#endif

namespace npact {

enum : int {
    OBS = 22, HID = 128,
    LN0_G = 0, LN0_B = 22,
    L1_B = 44, L1_W = L1_B + 128, LN1_G = L1_W + 22 * 128, LN1_B = LN1_G + 128,
    L2_B = LN1_B + 128, L2_W = L2_B + 128, LN2_G = L2_W + 128 * 128, LN2_B = LN2_G + 128,
    GI_B = LN2_B + 128, GI_W = GI_B + 384, GH_B = GI_W + 128 * 384, GH_W = GH_B + 384,
    LN3_G = GH_W + 128 * 384, LN3_B = LN3_G + 128,
    A1_B = LN3_B + 128, A1_W = A1_B + 128, LN4_G = A1_W + 128 * 128, LN4_B = LN4_G + 128,
    A2_B = LN4_B + 128, A2_W = A2_B + 128, LN5_G = A2_W + 128 * 128, LN5_B = LN5_G + 128,
    HD_B = LN5_B + 128, HD_W = HD_B + 4, TOTAL = HD_W + 128 * 4
};
static_assert(TOTAL == 153392, "packed actor layout (neuralplane_amd/actor.py)");

constexpr int TILE = 64, WAVES = 8, THREADS = TILE * WAVES, SLICE = HID / WAVES;
typedef const float __attribute__((address_space(4))) *cw_ptr;  // weights: read-only, wave-uniform -> scalar loads

__device__ __forceinline__ float act_exp(float x) {
    x = x < -87.0f ? -87.0f : x;
    x = x > 88.0f ? 88.0f : x;
    const float k = rintf(x * 1.44269504f);
    float r = fmaf(k, -0.693145752f, x);
    r = fmaf(k, -1.42860677e-6f, r);
    float p = fmaf(r, 1.38888889e-3f, 8.33333333e-3f);
    p = fmaf(r, p, 4.16666667e-2f);
    p = fmaf(r, p, 1.66666667e-1f);
    p = fmaf(r, p, 0.5f);
    p = fmaf(r, p, 1.0f);
    p = fmaf(r, p, 1.0f);
    return __uint_as_float(__float_as_uint(p) + ((uint32_t)(int32_t)k << 23));
}
#if NPACT_EXP & 8  // timing only (wrong results): free gate nonlinearities
__device__ __forceinline__ float act_sigmoid(float x) { return x; }
__device__ __forceinline__ float act_tanh(float x) { return x; }
#else
__device__ __forceinline__ float act_sigmoid(float x) { return 1.0f / (1.0f + act_exp(-x)); }
__device__ __forceinline__ float act_tanh(float x) { return 1.0f - 2.0f / (act_exp(2.0f * x) + 1.0f); }
#endif

// 16 outputs [j0, j0+16) of a Linear(128, .) layer for this lane's row; x[k] comes from LDS column `xin[k * TILE]`.
// acc = bias; acc = fma(W[j][k], x[k], acc), k ascending — as a generated weight-stream asm loop (np_actor_asm.inc): the 16
// weights of feature k are one s_load_dwordx16 feeding 8 v_pk_fma_f32 as SGPR-pair operands, 3 features per group, double-buffered.
template <int K, int LD_W>
__device__ __forceinline__ void dense16(cw_ptr bias, cw_ptr wt, int j0, const float *__restrict__ xin, float (&acc)[SLICE]) {
    static_assert(K == HID && SLICE == 16 && TILE == 64, "actor_dense16_asm is generated for Linear(128, .) slices of 16");
#if NPACT_EXP & 1  // timing only (wrong results): every wave streams the same weight slice
    j0 = 0;
#endif
    const float *w = (const float *)(wt + j0), *b = (const float *)(bias + j0);
    const unsigned xaddr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)xin;
#if NPACT_EXP & 2  // timing only (wrong results): every feature reads the same 64 B of weights (scalar cache always hits)
    actor_dense16_asm<0>(w, b, xaddr, acc);
#else
    actor_dense16_asm<LD_W * 4>(w, b, xaddr, acc);
#endif
}

// LayerNorm over the 128 features of a row held as 8 slices of 16 in the 8 waves; writes y into the LDS matrix `out`.
__device__ __forceinline__ void layernorm_slices(const float (&v)[SLICE], cw_ptr g, cw_ptr b, int j0, int wave, int lane,
                                                 float *__restrict__ part_s, float *__restrict__ part_q, float *__restrict__ out) {
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < SLICE; j++) s = s + v[j];
    part_s[wave * TILE + lane] = s;
    __syncthreads();
    float total = 0.0f;
#pragma unroll
    for (int w = 0; w < WAVES; w++) total = total + part_s[w * TILE + lane];
    const float mean = total / (float)HID;
    float q = 0.0f;
#pragma unroll
    for (int j = 0; j < SLICE; j++) {
        const float d = v[j] - mean;
        q = fmaf(d, d, q);
    }
    part_q[wave * TILE + lane] = q;
    __syncthreads();
    float qt = 0.0f;
#pragma unroll
    for (int w = 0; w < WAVES; w++) qt = qt + part_q[w * TILE + lane];
    const float rstd = 1.0f / sqrtf(qt / (float)HID + 1e-5f);
#pragma unroll
    for (int j = 0; j < SLICE; j++) out[(j0 + j) * TILE + lane] = fmaf((v[j] - mean) * rstd, g[j0 + j], b[j0 + j]);
    __syncthreads();  // `out` is complete (and part_s / part_q are free again) before anyone goes on
}

__device__ __forceinline__ void relu16(float (&v)[SLICE]) {
#pragma unroll
    for (int j = 0; j < SLICE; j++) v[j] = v[j] > 0.0f ? v[j] : 0.0f;
}

#ifndef NPACT_NO_KERNELS  // np_planning.hip takes the device functions only
__global__ __launch_bounds__(THREADS, 4) void actor_forward_kernel(const float *__restrict__ weights, long long n,
                                                                   const float *__restrict__ obs, const float *__restrict__ h_in,
                                                                   const float *__restrict__ mask, float *__restrict__ act,
                                                                   float *__restrict__ h_out) {
    extern __shared__ float lds[];
    // two activation matrices [feature][lane] + the LayerNorm partials: 68 KB, so that TWO tiles are resident per CU and one
    // tile's barrier / LayerNorm / activation phases overlap the other's weight streams
    float *bufA = lds, *bufB = lds + HID * TILE;
    float *part_s = lds + 2 * HID * TILE, *part_q = part_s + WAVES * TILE;
    const cw_ptr W = (cw_ptr)(unsigned long long)weights;
    const int lane = (int)(threadIdx.x % TILE);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / TILE));
    const int j0 = wave * SLICE;
    const long long i = (long long)blockIdx.x * TILE + lane;
    const bool valid = i < n;
    const long long ic = valid ? i : n - 1;

    // masked recurrent state (this wave's 16 features of its lane's row), gru.py:26 — held in registers until bufA is free
    const float mk = mask[ic];
    float hm[SLICE];
#pragma unroll
    for (int j = 0; j < SLICE; j++) hm[j] = h_in[ic * HID + j0 + j] * mk;

    // base.feature_norm: every wave normalises the 22 observations of its lane's row itself (two blocks: 16 + 6)
    float x0[OBS];
    {
        float xr[OBS];
#pragma unroll
        for (int j = 0; j < OBS; j++) xr[j] = obs[ic * OBS + j];
        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
        for (int j = 0; j < 16; j++) s0 = s0 + xr[j];
#pragma unroll
        for (int j = 16; j < OBS; j++) s1 = s1 + xr[j];
        const float mean = ((0.0f + s0) + s1) / (float)OBS;
        float q0 = 0.0f, q1 = 0.0f;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const float d = xr[j] - mean;
            q0 = fmaf(d, d, q0);
        }
#pragma unroll
        for (int j = 16; j < OBS; j++) {
            const float d = xr[j] - mean;
            q1 = fmaf(d, d, q1);
        }
        const float rstd = 1.0f / sqrtf(((0.0f + q0) + q1) / (float)OBS + 1e-5f);
#pragma unroll
        for (int j = 0; j < OBS; j++) x0[j] = fmaf((xr[j] - mean) * rstd, W[LN0_G + j], W[LN0_B + j]);
    }

    float v[SLICE];
    // base.mlp: Linear(22, 128) + ReLU + LayerNorm -> bufA
#pragma unroll
    for (int j = 0; j < SLICE; j++) v[j] = W[L1_B + j0 + j];
#pragma unroll
    for (int k = 0; k < OBS; k++) {
#pragma unroll
        for (int j = 0; j < SLICE; j++) v[j] = fmaf(W[L1_W + k * HID + j0 + j], x0[k], v[j]);
    }
    relu16(v);
    layernorm_slices(v, W + LN1_G, W + LN1_B, j0, wave, lane, part_s, part_q, bufA);
    // Linear(128, 128) + ReLU + LayerNorm -> bufB
    dense16<HID, HID>(W + L2_B, W + L2_W, j0, bufA + lane, v);
    relu16(v);
    layernorm_slices(v, W + LN2_G, W + LN2_B, j0, wave, lane, part_s, part_q, bufB);

    // rnn: GRU cell (gate order r, z, n as in torch) on x = bufB, h = bufA (dead since every wave passed LayerNorm 2's barriers)
#pragma unroll
    for (int j = 0; j < SLICE; j++) bufA[(j0 + j) * TILE + lane] = hm[j];
    __syncthreads();
    {
        float gi[SLICE], gh[SLICE], r[SLICE], z[SLICE];
        dense16<HID, 3 * HID>(W + GI_B, W + GI_W, j0, bufB + lane, gi);
        dense16<HID, 3 * HID>(W + GH_B, W + GH_W, j0, bufA + lane, gh);
#pragma unroll
        for (int j = 0; j < SLICE; j++) r[j] = act_sigmoid(gi[j] + gh[j]);
        dense16<HID, 3 * HID>(W + GI_B + HID, W + GI_W + HID, j0, bufB + lane, gi);
        dense16<HID, 3 * HID>(W + GH_B + HID, W + GH_W + HID, j0, bufA + lane, gh);
#pragma unroll
        for (int j = 0; j < SLICE; j++) z[j] = act_sigmoid(gi[j] + gh[j]);
        dense16<HID, 3 * HID>(W + GI_B + 2 * HID, W + GI_W + 2 * HID, j0, bufB + lane, gi);
        dense16<HID, 3 * HID>(W + GH_B + 2 * HID, W + GH_W + 2 * HID, j0, bufA + lane, gh);
#pragma unroll
        for (int j = 0; j < SLICE; j++) {
            const float nn = act_tanh(gi[j] + r[j] * gh[j]);
            v[j] = (bufA[(j0 + j) * TILE + lane] - nn) * z[j] + nn;  // h again from LDS: 16 fewer live registers across the gate GEMVs
        }
        if (valid) {
#pragma unroll
            for (int j = 0; j < SLICE; j++) h_out[i * HID + j0 + j] = v[j];
        }
    }
    // rnn.norm -> bufB: `out` is written after the LayerNorm's two barriers, i.e. after every wave finished reading x and h
    layernorm_slices(v, W + LN3_G, W + LN3_B, j0, wave, lane, part_s, part_q, bufB);
    // act.mlp
    dense16<HID, HID>(W + A1_B, W + A1_W, j0, bufB + lane, v);
    relu16(v);
    layernorm_slices(v, W + LN4_G, W + LN4_B, j0, wave, lane, part_s, part_q, bufA);
    dense16<HID, HID>(W + A2_B, W + A2_W, j0, bufA + lane, v);
    relu16(v);
    layernorm_slices(v, W + LN5_G, W + LN5_B, j0, wave, lane, part_s, part_q, bufB);
    // mu_net: Linear(128, 4) + tanh — wave 0
    if (wave == 0) {
        float m[4];
#pragma unroll
        for (int j = 0; j < 4; j++) m[j] = W[HD_B + j];
#pragma unroll 4
        for (int k = 0; k < HID; k++) {
            const float xk = bufB[k * TILE + lane];
#pragma unroll
            for (int j = 0; j < 4; j++) m[j] = fmaf(W[HD_W + k * 4 + j], xk, m[j]);
        }
        if (valid) {
#pragma unroll
            for (int j = 0; j < 4; j++) act[i * 4 + j] = act_tanh(m[j]);
        }
    }
}
#endif

// ------------------------------------------------------------------------------------------------------------------------------
// MFMA variant (the shipped one): the 128-wide layers are GEMMs [64 rows x 128] x [128 x 128], so they run on the matrix cores.
// v_mfma_f32_32x32x1_2b_f32 performs, per output element, ONE IEEE fused multiply-add per instruction (K = 1), so a chain of
// them over k IS acc = fmaf(W[f][k], x[k], acc), k ascending — bit-identical to the vector formulation above and to the CPU
// restatement (tools/microbench/mfma_exact.hip proves this on the hardware, denormals and overflow included).
//
// A workgroup of 4 waves owns a tile of 64 rows; wave w computes features [32w, 32w+32) for all 64 rows:
//   A operand = weights  (lane l -> feature 32w + l%32, the same in both 32-lane halves; coalesced 128 B vector loads from L2),
//   B operand = inputs   (lane l -> row l: block 0 = rows 0..31, block 1 = rows 32..63; one ds_read_b32 from the LDS matrix),
//   D (32 VGPRs): register e, lane l -> row 32*(e/16) + l%32, feature 32w + 8*((e%16)/4) + 4*(l/32) + e%4.
// Results are transposed through the LDS matrix [feature][row] (which the next layer's B operand reads anyway); LayerNorm then
// runs row-per-lane on 32 features per thread with the same block-of-16 summation order as the spec.
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int MW = 4, MTHREADS = TILE * MW, MSLICE = HID / MW;

__device__ __forceinline__ int mf_feat(int e, int half) { return 8 * ((e & 15) >> 2) + 4 * half + (e & 3); }
__device__ __forceinline__ int mf_row(int e, int l32) { return 32 * (e >> 4) + l32; }

// A operands (weights) of input features 0..7 for this lane, for the FIRST 128-wide layer: requested at kernel entry.  Every later
// layer gets them from the layer before it (dense_mfma fetches the next layer's first group behind its own last MFMAs), so no
// chain ever starts by waiting for L2.
template <int LD_W>
__device__ __forceinline__ void prefetch_w(const float *__restrict__ wt, int l32, float (&pa)[8]) {
#pragma unroll
    for (int u = 0; u < 8; u++) pa[u] = wt[u * LD_W + l32];
}

__device__ __forceinline__ void init_bias(const float *__restrict__ bias, int half, f32x32 &acc) {
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const float bv = bias[mf_feat(e, half)];
        acc[e] = bv;
        acc[e + 16] = bv;
    }
}

// acc = bias, then acc = fma(W[f][k], x[k], acc) for k = 0..127 on the matrix cores (generated loop, np_actor_mfma_asm.inc; operands
// fetched 8 features ahead of their use).  wt = &Wt[0][32w] (rows LD_W apart), bias = &bias[32w], xin = &X[0][lane] (LDS);
// pa: in = this layer's first 8 A operands, out = those of the layer at wnext (rows LD_NEXT apart).
template <int LD_W, int LD_NEXT>
__device__ __forceinline__ void dense_mfma(const float *__restrict__ bias, const float *__restrict__ wt, const float *__restrict__ wnext,
                                           const float *__restrict__ xin, int l32, int half, float (&pa)[8], f32x32 &acc) {
    init_bias(bias, half, acc);
    const unsigned xaddr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)xin;
#if NPACT_EXP & 4  // timing only (wrong results): no GEMV at all — what the rest of the kernel costs
    (void)xaddr;
#else
    actor_dense_mfma_asm<LD_W, LD_NEXT>(wt, wnext, 4u * (unsigned)l32, xaddr, pa, acc);
#endif
}

// D registers -> LDS matrix [feature][row]
template <bool RELU>
__device__ __forceinline__ void store_transposed(const f32x32 &acc, float *__restrict__ out, int f0, int l32, int half) {
#pragma unroll
    for (int e = 0; e < 32; e++) {
        float x = acc[e];
        if (RELU) x = x > 0.0f ? x : 0.0f;
        out[(f0 + mf_feat(e, half)) * TILE + mf_row(e, l32)] = x;
    }
}

// in-place LayerNorm of the LDS matrix `buf`: thread (wave, lane) owns row `lane`, features [32 wave, 32 wave + 32) = two of the
// eight blocks of 16 the spec sums in.  Ends with a barrier: `buf` is complete for every reader.
template <bool STORE_H>
__device__ __forceinline__ void layernorm_rows(float *__restrict__ buf, cw_ptr g, cw_ptr b, int wave, int lane, float *__restrict__ part_s,
                                               float *__restrict__ part_q, float *__restrict__ h_out_row) {
    const int f0 = wave * MSLICE;
    float v[MSLICE];
#pragma unroll
    for (int j = 0; j < MSLICE; j++) v[j] = buf[(f0 + j) * TILE + lane];
    if (STORE_H && h_out_row) {
        float4 *hq = reinterpret_cast<float4 *>(h_out_row + f0);
#pragma unroll
        for (int j = 0; j < MSLICE / 4; j++) hq[j] = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
    }
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; j++) s0 = s0 + v[j];
#pragma unroll
    for (int j = 16; j < 32; j++) s1 = s1 + v[j];
    part_s[(2 * wave) * TILE + lane] = s0;
    part_s[(2 * wave + 1) * TILE + lane] = s1;
    __syncthreads();
    float total = 0.0f;
#pragma unroll
    for (int w = 0; w < 8; w++) total = total + part_s[w * TILE + lane];
    const float mean = total / (float)HID;
    float q0 = 0.0f, q1 = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const float d = v[j] - mean;
        q0 = fmaf(d, d, q0);
    }
#pragma unroll
    for (int j = 16; j < 32; j++) {
        const float d = v[j] - mean;
        q1 = fmaf(d, d, q1);
    }
    part_q[(2 * wave) * TILE + lane] = q0;
    part_q[(2 * wave + 1) * TILE + lane] = q1;
    __syncthreads();
    float qt = 0.0f;
#pragma unroll
    for (int w = 0; w < 8; w++) qt = qt + part_q[w * TILE + lane];
    const float rstd = 1.0f / sqrtf(qt / (float)HID + 1e-5f);
#pragma unroll
    for (int j = 0; j < MSLICE; j++) buf[(f0 + j) * TILE + lane] = fmaf((v[j] - mean) * rstd, g[f0 + j], b[f0 + j]);
    __syncthreads();
}

#ifndef NPACT_NO_KERNELS  // np_planning.hip takes the device functions only
__global__ __launch_bounds__(MTHREADS, 2) void actor_forward_mfma_kernel(const float *__restrict__ weights, long long n,
                                                                         const float *__restrict__ obs, const float *__restrict__ h_in,
                                                                         const float *__restrict__ mask, float *__restrict__ act,
                                                                         float *__restrict__ h_out) {
    extern __shared__ float lds[];
    float *bufA = lds, *bufB = lds + HID * TILE;
    float *part_s = lds + 2 * HID * TILE, *part_q = part_s + 8 * TILE;
    const cw_ptr W = (cw_ptr)(unsigned long long)weights;  // wave-uniform reads (LayerNorm gains, the head): scalar loads
    const float *Wv = weights;                             // per-lane reads (MFMA A operands, biases): vector loads
    const int lane = (int)(threadIdx.x % TILE), l32 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / TILE));
    const int f0 = wave * MSLICE;
    const long long i = (long long)blockIdx.x * TILE + lane;
    const bool valid = i < n;
    const long long ic = valid ? i : n - 1;

    // the first layer's A operands (22 weights per lane) and the masked recurrent state (gru.py:26: this thread's 32 features of
    // its row, 128 B as 8 x 16 B) are requested first; both are consumed only after the observation LayerNorm
    float a1[OBS];
#pragma unroll
    for (int k = 0; k < OBS; k++) a1[k] = Wv[L1_W + k * HID + f0 + l32];
    const float mk = mask[ic];
    float hm[MSLICE];
    {
        const float4 *hp = reinterpret_cast<const float4 *>(h_in + ic * HID + f0);
#pragma unroll
        for (int j = 0; j < MSLICE / 4; j++) {
            const float4 q = hp[j];
            hm[4 * j] = q.x * mk;
            hm[4 * j + 1] = q.y * mk;
            hm[4 * j + 2] = q.z * mk;
            hm[4 * j + 3] = q.w * mk;
        }
    }

    // base.feature_norm (two blocks: 16 + 6) -> bufB rows 0..21; every wave computes it, wave 0 stores it
    {
        float xr[OBS];
#pragma unroll
        for (int j = 0; j < OBS; j++) xr[j] = obs[ic * OBS + j];
        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
        for (int j = 0; j < 16; j++) s0 = s0 + xr[j];
#pragma unroll
        for (int j = 16; j < OBS; j++) s1 = s1 + xr[j];
        const float mean = ((0.0f + s0) + s1) / (float)OBS;
        float q0 = 0.0f, q1 = 0.0f;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const float d = xr[j] - mean;
            q0 = fmaf(d, d, q0);
        }
#pragma unroll
        for (int j = 16; j < OBS; j++) {
            const float d = xr[j] - mean;
            q1 = fmaf(d, d, q1);
        }
        const float rstd = 1.0f / sqrtf(((0.0f + q0) + q1) / (float)OBS + 1e-5f);
        if (wave == 0) {
#pragma unroll
            for (int j = 0; j < OBS; j++) bufB[j * TILE + lane] = fmaf((xr[j] - mean) * rstd, W[LN0_G + j], W[LN0_B + j]);
        }
    }
    float pa[8];
    prefetch_w<HID>(Wv + L2_W + f0, l32, pa);
    __syncthreads();

    f32x32 acc;
    // base.mlp: Linear(22, 128) + ReLU + LayerNorm -> bufA
    init_bias(Wv + L1_B + f0, half, acc);
#pragma unroll
    for (int k = 0; k < OBS; k++) acc = __builtin_amdgcn_mfma_f32_32x32x1f32(a1[k], bufB[k * TILE + lane], acc, 0, 0, 0);
    store_transposed<true>(acc, bufA, f0, l32, half);
    __syncthreads();
    layernorm_rows<false>(bufA, W + LN1_G, W + LN1_B, wave, lane, part_s, part_q, nullptr);
    // Linear(128, 128) + ReLU + LayerNorm -> bufB
    dense_mfma<HID, 3 * HID>(Wv + L2_B + f0, Wv + L2_W + f0, Wv + GI_W + f0, bufA + lane, l32, half, pa, acc);
    store_transposed<true>(acc, bufB, f0, l32, half);
    __syncthreads();  // also: every wave is done reading bufA
    layernorm_rows<false>(bufB, W + LN2_G, W + LN2_B, wave, lane, part_s, part_q, nullptr);

    // rnn: GRU cell (gate order r, z, n as in torch) on x = bufB, h = bufA
#pragma unroll
    for (int j = 0; j < MSLICE; j++) bufA[(f0 + j) * TILE + lane] = hm[j];
    __syncthreads();
    {
        f32x32 gi, gh, r, z;
        dense_mfma<3 * HID, 3 * HID>(Wv + GI_B + f0, Wv + GI_W + f0, Wv + GH_W + f0, bufB + lane, l32, half, pa, gi);
        dense_mfma<3 * HID, 3 * HID>(Wv + GH_B + f0, Wv + GH_W + f0, Wv + GI_W + HID + f0, bufA + lane, l32, half, pa, gh);
#pragma unroll
        for (int e = 0; e < 32; e++) r[e] = act_sigmoid(gi[e] + gh[e]);
        dense_mfma<3 * HID, 3 * HID>(Wv + GI_B + HID + f0, Wv + GI_W + HID + f0, Wv + GH_W + HID + f0, bufB + lane, l32, half, pa, gi);
        dense_mfma<3 * HID, 3 * HID>(Wv + GH_B + HID + f0, Wv + GH_W + HID + f0, Wv + GI_W + 2 * HID + f0, bufA + lane, l32, half, pa, gh);
#pragma unroll
        for (int e = 0; e < 32; e++) z[e] = act_sigmoid(gi[e] + gh[e]);
        dense_mfma<3 * HID, 3 * HID>(Wv + GI_B + 2 * HID + f0, Wv + GI_W + 2 * HID + f0, Wv + GH_W + 2 * HID + f0, bufB + lane, l32, half, pa, gi);
        dense_mfma<3 * HID, HID>(Wv + GH_B + 2 * HID + f0, Wv + GH_W + 2 * HID + f0, Wv + A1_W + f0, bufA + lane, l32, half, pa, gh);
#pragma unroll
        for (int e = 0; e < 32; e++) {
            const float nn = act_tanh(gi[e] + r[e] * gh[e]);
            const float hp = bufA[(f0 + mf_feat(e, half)) * TILE + mf_row(e, l32)];
            acc[e] = (hp - nn) * z[e] + nn;
        }
    }
    __syncthreads();  // every wave is done reading x (bufB) and h (bufA)
    store_transposed<false>(acc, bufB, f0, l32, half);
    __syncthreads();
    // new recurrent state out (128 B per thread), then rnn.norm in place
    layernorm_rows<true>(bufB, W + LN3_G, W + LN3_B, wave, lane, part_s, part_q, valid ? h_out + i * HID : nullptr);
    // act.mlp
    dense_mfma<HID, HID>(Wv + A1_B + f0, Wv + A1_W + f0, Wv + A2_W + f0, bufB + lane, l32, half, pa, acc);
    store_transposed<true>(acc, bufA, f0, l32, half);
    __syncthreads();
    layernorm_rows<false>(bufA, W + LN4_G, W + LN4_B, wave, lane, part_s, part_q, nullptr);
    dense_mfma<HID, HID>(Wv + A2_B + f0, Wv + A2_W + f0, Wv + A2_W + f0, bufA + lane, l32, half, pa, acc);  // nothing follows
    store_transposed<true>(acc, bufB, f0, l32, half);
    __syncthreads();
    layernorm_rows<false>(bufB, W + LN5_G, W + LN5_B, wave, lane, part_s, part_q, nullptr);
    // mu_net: Linear(128, 4) + tanh — wave 0
    if (wave == 0) {
        float m[4];
#pragma unroll
        for (int j = 0; j < 4; j++) m[j] = W[HD_B + j];
#pragma unroll 4
        for (int k = 0; k < HID; k++) {
            const float xk = bufB[k * TILE + lane];
#pragma unroll
            for (int j = 0; j < 4; j++) m[j] = fmaf(W[HD_W + k * 4 + j], xk, m[j]);
        }
        if (valid) {
#pragma unroll
            for (int j = 0; j < 4; j++) act[i * 4 + j] = act_tanh(m[j]);
        }
    }
}
#endif

constexpr size_t ACTOR_LDS_BYTES = sizeof(float) * (2 * HID * TILE + 2 * WAVES * TILE);

// ------------------------------------------------------------------------------------------------------------------------------
// Small-batch variant (round 3): tiles of 32 rows.  At n <= 16 K the 64-row kernel occupies n / 64 of the 256 CUs with one wave
// per SIMD, every wave a serial chain of 1 174 sixteen-pass MFMAs (31 us) with the LayerNorm / gate epilogues exposed in between
// (59 us per call, 50 calls per PlanningEnv.step).  Here a wave computes 32 features x 32 rows per v_mfma_f32_16x16x1_4b_f32
// (eight passes; again ONE IEEE fma per element and instruction, tools/microbench/mfma16_exact.hip), so a tile's chain is half as
// long, twice as many CUs carry tiles, and where two tiles share a CU one's epilogues hide behind the other's MFMAs.
//   A operand = weights  (lane l -> feature 32w + 16 (l / 32) + l % 16),
//   B operand = inputs   (lane l -> row l % 32; one ds_read from the LDS matrix [feature][32 rows]),
//   D (16 VGPRs): register r, lane l -> feature 32w + 16 (r / 8) + 4 (l / 16) + r % 4, row 16 ((r / 4) % 2) + l % 16.
// Row-per-lane phases (LayerNorm, state load / store): thread t -> row t % 32, the block of 16 features t / 32 — the blocks the
// numerics spec sums in, so every result is bit-identical to the 64-row kernels.
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int T32 = 32, BLK = 16;

#ifndef NPACT_TRACE
#define NPACT_TRACE 0  // 1: workgroup 0 / wave 0 stamps the shader clock after every phase (tools/microbench/actor_phases.py); never shipped
#endif
#if NPACT_TRACE
__device__ long long npact_trace[64];
#define NPACT_STAMP(k) do { if (blockIdx.x == gridDim.x - 1 && tid == 0) npact_trace[k] = __builtin_readcyclecounter(); } while (0)
#else
#define NPACT_STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ int m16_feat(int r, int g4) { return 16 * (r >> 3) + 4 * g4 + (r & 3); }
__device__ __forceinline__ int m16_row(int r, int l16) { return 16 * ((r >> 2) & 1) + l16; }

template <int LD_W>
__device__ __forceinline__ void prefetch_w16(const float *__restrict__ wt, const float *__restrict__ bias, int fl, float (&pa)[33]) {
#pragma unroll
    for (int u = 0; u < 32; u++) pa[u] = wt[u * LD_W + fl];
    pa[32] = bias[fl];
}

template <int LD_W, int LD_NEXT>
__device__ __forceinline__ void dense_mfma16(const float *__restrict__ wt, const float *__restrict__ wnext, const float *__restrict__ bnext,
                                             const float *__restrict__ xin, int fl, float (&pa)[33], f32x16 &acc) {
    const unsigned xaddr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)xin;
#if NPACT_PRIO   // two tiles on one CU: a dependent MFMA chain otherwise starves the other wave of the SIMD (tools/microbench/mfma_coissue.hip)
    __builtin_amdgcn_s_setprio(0);
#endif
    actor_dense_mfma16_asm<LD_W, LD_NEXT>(wt, wnext, bnext, 4u * (unsigned)fl, xaddr, pa, acc);
#if NPACT_PRIO
    __builtin_amdgcn_s_setprio(NPACT_PRIO);
#endif
}

template <bool RELU>
__device__ __forceinline__ void store_transposed16(const f32x16 &acc, float *__restrict__ out, int f0, int l16, int g4) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
        float x = acc[r];
        if (RELU) x = x > 0.0f ? x : 0.0f;
        out[(f0 + m16_feat(r, g4)) * T32 + m16_row(r, l16)] = x;
    }
}

// in-place LayerNorm of the LDS matrix `buf` [feature][32 rows]: thread -> (row, block of 16 features).  gb = {gains, shifts} of this
// thread's block (per-lane loads: the block differs between the two halves of a wave).  v returns the values before the normalisation
// (the caller stores the new recurrent state from them).  Ends with a barrier.
__device__ __forceinline__ void layernorm_rows16(float *__restrict__ buf, const float *__restrict__ g, const float *__restrict__ b, int blk,
                                                 int row, float *__restrict__ part_s, float *__restrict__ part_q, float (&v)[BLK]) {
    const int f0 = blk * BLK;
    float gg[BLK], bb[BLK];
    {
        const float4 *gp = reinterpret_cast<const float4 *>(g + f0), *bp = reinterpret_cast<const float4 *>(b + f0);
#pragma unroll
        for (int j = 0; j < BLK / 4; j++) {
            const float4 x = gp[j], y = bp[j];
            gg[4 * j] = x.x, gg[4 * j + 1] = x.y, gg[4 * j + 2] = x.z, gg[4 * j + 3] = x.w;
            bb[4 * j] = y.x, bb[4 * j + 1] = y.y, bb[4 * j + 2] = y.z, bb[4 * j + 3] = y.w;
        }
    }
#pragma unroll
    for (int j = 0; j < BLK; j++) v[j] = buf[(f0 + j) * T32 + row];
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < BLK; j++) s = s + v[j];
    part_s[blk * T32 + row] = s;
    __syncthreads();
    float total = 0.0f;
#pragma unroll
    for (int w = 0; w < 8; w++) total = total + part_s[w * T32 + row];
    const float mean = total / (float)HID;
    float q = 0.0f;
#pragma unroll
    for (int j = 0; j < BLK; j++) {
        const float d = v[j] - mean;
        q = fmaf(d, d, q);
    }
    part_q[blk * T32 + row] = q;
    __syncthreads();
    float qt = 0.0f;
#pragma unroll
    for (int w = 0; w < 8; w++) qt = qt + part_q[w * T32 + row];
    const float rstd = 1.0f / sqrtf(qt / (float)HID + 1e-5f);
#pragma unroll
    for (int j = 0; j < BLK; j++) buf[(f0 + j) * T32 + row] = fmaf((v[j] - mean) * rstd, gg[j], bb[j]);
    __syncthreads();
}

constexpr int ACTOR32_LDS_FLOATS = 2 * HID * T32 + 2 * 8 * T32 + 4 * HID;
constexpr int ACTOR32_HEAD_W = 2 * HID * T32 + 2 * 8 * T32;  // float offset of the staged mu_net weights [k][4] inside the tile's LDS
constexpr int ACTOR32_BARRIERS = 23;  // __syncthreads() executed by actor32_body (straight-line code): what a wave that sits the call out has to match

// One 32-row tile of the controller, the calling workgroup's waves 0..3 (tid = threadIdx.x < 256), in three pieces so that the
// persistent PlanningEnv kernel (np_planning.hip) can feed it from registers / LDS instead of global memory:
//   actor32_request_l1   the first layer's A operands of this lane (22 weights + the bias): requested first, consumed after the obs LayerNorm
//   actor32_stage_head   mu_net's weights -> LDS (the head reads them 128 times in a row)
//   actor32_body         observation LayerNorm ... head: xr = the 22 raw observations of this thread's row, hm = the MASKED recurrent
//                        state (gru.py:26) of (row, block of 16 features); returns hn = the new recurrent state of (row, block) and
//                        `action` = tanh(mu) of (row, wave) — the caller stores them (lanes with hi == 0 hold the action)
// actor_tile32 = the three with global loads / stores around them: the body of actor_forward_mfma32_kernel.
struct Actor32Pre {
    float a1[OBS];
    float b1;
};

__device__ __forceinline__ void actor32_request_l1(const float *weights, unsigned tid, Actor32Pre &pre) {
    const int lane = (int)(tid % TILE), l16 = lane & 15, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(tid / TILE));
    const int f0 = wave * MSLICE, fl = 16 * hi + l16;
#pragma unroll
    for (int k = 0; k < OBS; k++) pre.a1[k] = weights[L1_W + k * HID + f0 + fl];
    pre.b1 = weights[L1_B + f0 + fl];
}

__device__ __forceinline__ void actor32_stage_head(float *lds, const float *weights, unsigned tid) {
    float *head_w = lds + ACTOR32_HEAD_W;
    if (tid < HID) reinterpret_cast<float4 *>(head_w)[tid] = reinterpret_cast<const float4 *>(weights + HD_W)[tid];
}

// TANH = false: `action` = mu itself (np_policy.hip: the critic's value head, the sampled actor's epilogue); NOBS < 22: a network on fewer
// observations (the 1v1 combat policy's 15) in the same packed layout — xr[j >= NOBS] and the first layer's rows k >= NOBS are not read
template <bool TANH = true, int NOBS = OBS>
__device__ __forceinline__ void actor32_body(float *lds, const float *weights, const Actor32Pre &pre, const float (&xr)[OBS], const float (&hm)[BLK],
                                             float (&hn)[BLK], float &action, unsigned tid) {
    float *bufA = lds, *bufB = lds + HID * T32;
    float *part_s = lds + 2 * HID * T32, *part_q = part_s + 8 * T32;
    float *head_w = part_q + 8 * T32;
    const cw_ptr W = (cw_ptr)(unsigned long long)weights;  // wave-uniform reads: scalar loads
    const float *Wv = weights;                             // per-lane reads
    const int lane = (int)(tid % TILE), row = lane & 31, l16 = lane & 15, g4 = lane >> 4, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(tid / TILE));
    const int f0 = wave * MSLICE;        // MFMA role: features [f0, f0 + 32); this lane's A operand is feature f0 + fl
    const int fl = 16 * hi + l16;
    const int blk = 2 * wave + hi;       // row role: features [16 blk, 16 blk + 16) of row `row`
    // base.feature_norm (two blocks: 16 + 6) -> bufB rows 0..21; every thread computes its row's, block 0 stores it
    {
        constexpr int N0 = NOBS < 16 ? NOBS : 16;   // the first block of 16 (or fewer) features
        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
        for (int j = 0; j < N0; j++) s0 = s0 + xr[j];
#pragma unroll
        for (int j = 16; j < NOBS; j++) s1 = s1 + xr[j];
        const float mean = ((0.0f + s0) + s1) / (float)NOBS;
        float q0 = 0.0f, q1 = 0.0f;
#pragma unroll
        for (int j = 0; j < N0; j++) {
            const float d = xr[j] - mean;
            q0 = fmaf(d, d, q0);
        }
#pragma unroll
        for (int j = 16; j < NOBS; j++) {
            const float d = xr[j] - mean;
            q1 = fmaf(d, d, q1);
        }
        const float rstd = 1.0f / sqrtf(((0.0f + q0) + q1) / (float)NOBS + 1e-5f);
        if (blk == 0) {
#pragma unroll
            for (int j = 0; j < NOBS; j++) bufB[j * T32 + row] = fmaf((xr[j] - mean) * rstd, W[LN0_G + j], W[LN0_B + j]);
        }
    }
    float pa[33];
    prefetch_w16<HID>(Wv + L2_W + f0, Wv + L2_B + f0, fl, pa);
    __syncthreads();
    NPACT_STAMP(1);

    f32x16 acc;
    float lv[BLK];  // LayerNorm inputs of this thread (only rnn.norm's are used afterwards)
    // base.mlp: Linear(22, 128) + ReLU + LayerNorm -> bufA
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    acc = __builtin_amdgcn_mfma_f32_16x16x1f32(pre.b1, 1.0f, acc, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < NOBS; k++) acc = __builtin_amdgcn_mfma_f32_16x16x1f32(pre.a1[k], bufB[k * T32 + row], acc, 0, 0, 0);
    store_transposed16<true>(acc, bufA, f0, l16, g4);
    __syncthreads();
    NPACT_STAMP(2);
    layernorm_rows16(bufA, Wv + LN1_G, Wv + LN1_B, blk, row, part_s, part_q, lv);
    NPACT_STAMP(3);
    // Linear(128, 128) + ReLU + LayerNorm -> bufB
    dense_mfma16<HID, 3 * HID>(Wv + L2_W + f0, Wv + GI_W + f0, Wv + GI_B + f0, bufA + row, fl, pa, acc);
    NPACT_STAMP(4);
    store_transposed16<true>(acc, bufB, f0, l16, g4);
    __syncthreads();  // also: every wave is done reading bufA
    layernorm_rows16(bufB, Wv + LN2_G, Wv + LN2_B, blk, row, part_s, part_q, lv);
    NPACT_STAMP(5);

    // rnn: GRU cell (gate order r, z, n as in torch) on x = bufB, h = bufA
#pragma unroll
    for (int j = 0; j < BLK; j++) bufA[(blk * BLK + j) * T32 + row] = hm[j];
    __syncthreads();
    {
        f32x16 gi, gh, rr, z;
        dense_mfma16<3 * HID, 3 * HID>(Wv + GI_W + f0, Wv + GH_W + f0, Wv + GH_B + f0, bufB + row, fl, pa, gi);
        NPACT_STAMP(6);
        dense_mfma16<3 * HID, 3 * HID>(Wv + GH_W + f0, Wv + GI_W + HID + f0, Wv + GI_B + HID + f0, bufA + row, fl, pa, gh);
NPACT_STAMP(7);
#pragma unroll
        for (int e = 0; e < 16; e++) rr[e] = act_sigmoid(gi[e] + gh[e]);
        NPACT_STAMP(8);
        dense_mfma16<3 * HID, 3 * HID>(Wv + GI_W + HID + f0, Wv + GH_W + HID + f0, Wv + GH_B + HID + f0, bufB + row, fl, pa, gi);
        dense_mfma16<3 * HID, 3 * HID>(Wv + GH_W + HID + f0, Wv + GI_W + 2 * HID + f0, Wv + GI_B + 2 * HID + f0, bufA + row, fl, pa, gh);
NPACT_STAMP(9);
#pragma unroll
        for (int e = 0; e < 16; e++) z[e] = act_sigmoid(gi[e] + gh[e]);
        NPACT_STAMP(10);
        dense_mfma16<3 * HID, 3 * HID>(Wv + GI_W + 2 * HID + f0, Wv + GH_W + 2 * HID + f0, Wv + GH_B + 2 * HID + f0, bufB + row, fl, pa, gi);
        dense_mfma16<3 * HID, HID>(Wv + GH_W + 2 * HID + f0, Wv + A1_W + f0, Wv + A1_B + f0, bufA + row, fl, pa, gh);
NPACT_STAMP(11);
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const float nn = act_tanh(gi[e] + rr[e] * gh[e]);
            const float hp = bufA[(f0 + m16_feat(e, g4)) * T32 + m16_row(e, l16)];
            acc[e] = (hp - nn) * z[e] + nn;
        }
    }
    __syncthreads();  // every wave is done reading x (bufB) and h (bufA)
    NPACT_STAMP(12);
    store_transposed16<false>(acc, bufB, f0, l16, g4);
    __syncthreads();
    // rnn.norm in place; hn = the new recurrent state (this thread's 16 features), stored by the caller at the very end: a store issued
    // here would be the oldest entry of the vector-memory queue that the next layer's operand waits (s_waitcnt vmcnt) have to drain
    layernorm_rows16(bufB, Wv + LN3_G, Wv + LN3_B, blk, row, part_s, part_q, hn);
    NPACT_STAMP(13);
    // act.mlp
    dense_mfma16<HID, HID>(Wv + A1_W + f0, Wv + A2_W + f0, Wv + A2_B + f0, bufB + row, fl, pa, acc);
    NPACT_STAMP(14);
    store_transposed16<true>(acc, bufA, f0, l16, g4);
    __syncthreads();
    layernorm_rows16(bufA, Wv + LN4_G, Wv + LN4_B, blk, row, part_s, part_q, lv);
    NPACT_STAMP(15);
    dense_mfma16<HID, HID>(Wv + A2_W + f0, Wv + A2_W + f0, Wv + A2_B + f0, bufA + row, fl, pa, acc);  // nothing follows
    NPACT_STAMP(16);
    store_transposed16<true>(acc, bufB, f0, l16, g4);
    __syncthreads();
    layernorm_rows16(bufB, Wv + LN5_G, Wv + LN5_B, blk, row, part_s, part_q, lv);
    NPACT_STAMP(17);
    // mu_net: Linear(128, 4) + tanh — wave j computes action j of the 32 rows
    {
        float m = W[HD_B + wave];
#pragma unroll
        for (int k = 0; k < HID; k++) m = fmaf(head_w[k * 4 + wave], bufB[k * T32 + row], m);
        action = TANH ? act_tanh(m) : m;
    }
}

// tile `tile` = rows [32 tile, 32 tile + 32) through global memory.  Plain (aliasing) pointers: inside the persistent kernel the buffers
// read here are written by the same workgroup between calls.  tid = threadIdx.x (the persistent kernel hands it over through an opaque
// copy per iteration, so that per-thread addresses are recomputed, not kept in registers across its loop).
__device__ __forceinline__ void actor_tile32(float *lds, const float *weights, long long n, const float *obs, const float *h_in,
                                             const float *mask, float *act, float *h_out, long long tile, unsigned tid) {
    const int lane = (int)(tid % TILE), row = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(tid / TILE));
    const int blk = 2 * wave + hi;
    const long long i = tile * T32 + row;
    const bool valid = i < n;
    const long long ic = valid ? i : n - 1;
    NPACT_STAMP(0);
    // requested first, consumed after the observation LayerNorm: the first layer's A operands and the masked recurrent state
    // (gru.py:26: this thread's 16 features of its row, 64 B)
    Actor32Pre pre;
    actor32_request_l1(weights, tid, pre);
    const float mk = mask[ic];
    float hm[BLK];
    {
        const float4 *hp = reinterpret_cast<const float4 *>(h_in + ic * HID + blk * BLK);
#pragma unroll
        for (int j = 0; j < BLK / 4; j++) {
            const float4 q = hp[j];
            hm[4 * j] = q.x * mk;
            hm[4 * j + 1] = q.y * mk;
            hm[4 * j + 2] = q.z * mk;
            hm[4 * j + 3] = q.w * mk;
        }
    }
    actor32_stage_head(lds, weights, tid);
    float xr[OBS];
#pragma unroll
    for (int j = 0; j < OBS; j++) xr[j] = obs[ic * OBS + j];
    float hn[BLK], action;
    actor32_body(lds, weights, pre, xr, hm, hn, action, tid);
    if (valid && hi == 0) act[i * 4 + wave] = action;
    if (valid) {
        float4 *hq = reinterpret_cast<float4 *>(h_out + i * HID + blk * BLK);
#pragma unroll
        for (int j = 0; j < BLK / 4; j++) hq[j] = make_float4(hn[4 * j], hn[4 * j + 1], hn[4 * j + 2], hn[4 * j + 3]);
    }
    NPACT_STAMP(18);
}

#ifndef NPACT_NO_KERNELS  // np_planning.hip takes the device functions only
__global__ __launch_bounds__(MTHREADS, 2) void actor_forward_mfma32_kernel(const float *__restrict__ weights, long long n,
                                                                           const float *__restrict__ obs, const float *__restrict__ h_in,
                                                                           const float *__restrict__ mask, float *__restrict__ act,
                                                                           float *__restrict__ h_out) {
    __shared__ float lds[ACTOR32_LDS_FLOATS];
    actor_tile32(lds, weights, n, obs, h_in, mask, act, h_out, (long long)blockIdx.x, threadIdx.x);
}
#endif

}  // namespace npact
