// np_actor_i8.hip — the controller's block-fixed-point kernel (np_actor_i8.h) as a stand-alone launch, its launcher, and the load-time packer
// of the quantised weights (np_actor_pack_i8, C ABI).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>

#include "../../include/neuralplane_amd.h"
#define NPACT_NO_KERNELS 1
#include "np_actor_i8.h"
#include "np_policy.h"

namespace npact8 {

// One 32-aircraft tile per four-wave workgroup, 77 KB of LDS (the call's block, its GRU parking area, the staged tables): two workgroups per CU,
// whose phases (matrix / vector / exchange) overlap each other.
__global__ __launch_bounds__(256, 2) void actor_forward_i8_kernel(const float *__restrict__ weights, long long n, const float *__restrict__ obs,
                                                                  const float *__restrict__ h_in, const float *__restrict__ mask, float *__restrict__ act,
                                                                  float *__restrict__ h_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // dynamic: above the 64 KB a kernel may use without asking
    actor8_tiles<1>(lds, weights, n, obs, h_in, mask, act, h_out, (long long)blockIdx.x, threadIdx.x);
}

hipError_t launch_actor_i8(const float *weights, long long n, const float *obs, const float *h_in, const float *masks, float *actions, float *h_out,
                           hipStream_t stream) {
    constexpr size_t bytes = sizeof(float) * actor8_tile_lds_floats<1>();
    static_assert(2 * bytes <= 160 * 1024, "two workgroups per CU");
    static bool set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipGetLastError();
    if (dev < 64 && !set[dev]) {
        const hipError_t e = hipFuncSetAttribute((const void *)actor_forward_i8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
        set[dev] = true;
    }
    hipLaunchKernelGGL(actor_forward_i8_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), bytes, stream, weights, n, obs, h_in, masks, actions, h_out);
    return hipGetLastError();
}

// The rollout policy's inference step (np_policy.hip states the act layer / value head) with both networks in these numerics: one workgroup per
// (32-row tile, network), grid.y 0 = actor, 1 = critic.
template <int NOBS>
__global__ __launch_bounds__(256, 2) void policy_act_i8_kernel(const nppol::ActArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned tid = threadIdx.x;
    const int net = __builtin_amdgcn_readfirstlane(a.first_net + (int)blockIdx.y);
    const float *weights = a.w[net];
    const float *h_in = a.h_in[net];
    float *h_out = a.h_out[net];
    const long long n = a.n;
    const int lane = (int)(tid & 63u), row = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const long long i = (long long)blockIdx.x * 32 + row;
    const bool valid = i < n;
    const long long ic = valid ? i : n - 1;
    float hm[1][16], xr[1][OBS], hn[1][16], mu[1];
    {
        float mk;   // masks[i], or the previous env step's flags turned into the runner's insert rule (np_policy.hip)
        bool ended = false;
        if (a.prev) {
            const bool d = a.prev[ic] != 0, b = a.prev[n + ic] != 0;
            ended = d || b || a.prev[2 * n + ic] != 0;
            mk = 1.0f;
            if (valid && net == a.first_net && w == 0 && h == 0) {
                a.masks_out[i] = d ? 0.0f : 1.0f;
                a.bad_masks_out[i] = b ? 0.0f : 1.0f;
            }
        } else {
            mk = a.mask[ic];
        }
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const float4 *hp = reinterpret_cast<const float4 *>(h_in + ic * HID + 32 * w + 4 * h + 8 * g);
            const float4 q = *hp;
            hm[0][4 * g] = ended ? 0.0f : q.x * mk; hm[0][4 * g + 1] = ended ? 0.0f : q.y * mk;
            hm[0][4 * g + 2] = ended ? 0.0f : q.z * mk; hm[0][4 * g + 3] = ended ? 0.0f : q.w * mk;
            if (ended && valid) *const_cast<float4 *>(hp) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
#pragma unroll
        for (int j = 0; j < NOBS; j++) xr[0][j] = a.obs[ic * NOBS + j];
    }
    float *park = lds + ACTOR8_LDS_FLOATS, *tab = park + ACTOR8_PARK_FLOATS;
    actor8_stage_tables(tab, weights, tid, 256u);
    __syncthreads();
    actor8_body<1, false, NOBS>(lds, park, tab, weights, xr, hm, hn, mu, tid);   // mu of (row, head column = wave) in the lanes with h == 0
    if (net == 0) {
        float *lp = lds + LDS8_PS;   // the LayerNorm exchange: every wave is past its last read (the head's barrier)
        const int A = a.act_dim;
        if (h == 0 && w < A) {
            const float mean = act_tanh(mu[0]);
            float act = mean;
            if (!(a.flags & NP_POLICY_DETERMINISTIC)) {
                const float e = a.noise[ic * A + w] * a.std[w];
                act = e + mean;
            }
            const float d = act - mean;
            const float q = -(d * d);
            const float var = a.std[w] * a.std[w];
            float t = q / (2.0f * var);
            t = t - a.log_std[w];
            t = t - 0.9189385f;
            lp[w * 32 + row] = t;
            if (valid) a.actions[i * A + w] = act;
        }
        __syncthreads();
        if (w == 0 && h == 0 && valid) {
            float s = lp[row];
            for (int j = 1; j < A; j++) s = s + lp[j * 32 + row];
            a.log_probs[i] = s;
        }
    } else if (w == 0 && h == 0 && valid) {
        a.values[i] = mu[0];
    }
    if (valid) {
#pragma unroll
        for (int g = 0; g < 4; g++)
            *reinterpret_cast<float4 *>(h_out + i * HID + 32 * w + 4 * h + 8 * g) = make_float4(hn[0][4 * g], hn[0][4 * g + 1], hn[0][4 * g + 2], hn[0][4 * g + 3]);
    }
}

hipError_t launch_policy_act_i8(const nppol::ActArgs &a, hipStream_t stream) {
    constexpr size_t bytes = sizeof(float) * actor8_tile_lds_floats<1>();
    static bool set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipGetLastError();
    if (dev < 64 && !set[dev]) {
        hipError_t e = hipFuncSetAttribute((const void *)policy_act_i8_kernel<OBS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)policy_act_i8_kernel<15>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
        set[dev] = true;
    }
    const int nets = ((a.flags & NP_POLICY_ACTOR) ? 1 : 0) + ((a.flags & NP_POLICY_CRITIC) ? 1 : 0);
    const dim3 grid((unsigned)((a.n + 31) / 32), (unsigned)nets);
    if (a.obs_dim == 15)
        hipLaunchKernelGGL(policy_act_i8_kernel<15>, grid, dim3(256), bytes, stream, a);
    else
        hipLaunchKernelGGL(policy_act_i8_kernel<OBS>, grid, dim3(256), bytes, stream, a);
    return hipGetLastError();
}

}  // namespace npact8

// ---- load-time packer (host) -----------------------------------------------------------------------------------------------------------
namespace {
int exponent_of_host(float b) {
    uint32_t bits;
    std::memcpy(&bits, &b, 4);
    const int e = (int)((bits >> 23) & 255u) - 126;
    return e < -100 ? -100 : e;
}
float pow2_host(int e) {
    const uint32_t bits = (uint32_t)(e + 127) << 23;
    float x;
    std::memcpy(&x, &bits, 4);
    return x;
}
// one Linear layer: wt = W^T, k-major [n_in][ld]; outputs [j0, j0 + 32 * blocks) as M-blocks of 32 -> fragments + scales
void pack_layer(const float *wt, int n_in, int ld, int j0, int blocks, int ks_count, unsigned char *frag, float *sw) {
    using namespace npact8;
    for (int mb = 0; mb < blocks; mb++) {
        for (int m = 0; m < 32; m++) {
            const int j = j0 + 32 * mb + m;
            float mx = 0.0f;
            for (int k = 0; k < n_in; k++) mx = std::fmax(mx, std::fabs(wt[k * ld + j]));
            const int ew = exponent_of_host(mx);
            sw[32 * mb + m] = pow2_host(ew - 18);
            for (int ks = 0; ks < ks_count; ks++)
                for (int hh = 0; hh < 2; hh++)
                    for (int e = 0; e < 16; e++) {
                        const int k = 32 * ks + 8 * (e >> 2) + 4 * hh + (e & 3);
                        uint32_t p = 0u;
                        if (k < n_in) {
                            const int32_t q = (int32_t)std::nearbyint(std::ldexp((double)wt[k * ld + j], WBITS - ew));   // exact product, round-half-even
                            p = ((uint32_t)q + 0x80808080u) ^ 0x80808080u;
                        }
                        for (int limb = 0; limb < 4; limb++)
                            frag[(size_t)((mb * ks_count + ks) * 4 + limb) * FRAG_BYTES + (size_t)(m + 32 * hh) * 16 + e] = (unsigned char)(p >> (8 * limb));
                    }
        }
    }
}
}  // namespace

extern "C" int np_actor_pack_i8(const float *packed_fp32, float *out) {
    using namespace npact8;
    if (!packed_fp32 || !out) return np_internal_fail("np_actor_pack_i8: null argument");
    for (int k = 0; k < TOTAL; k++)   // a NaN / infinite weight has no fixed-point image ((int32_t)nearbyint(...) of it is undefined behaviour)
        if (!std::isfinite(packed_fp32[k])) return np_internal_fail("np_actor_pack_i8: non-finite weight (the block-fixed-point numerics need finite parameters; numerics='fp32' carries them)");
    std::memcpy(out, packed_fp32, sizeof(float) * TOTAL);
    std::memset(out + TOTAL, 0, sizeof(float) * (size_t)(TOTAL_I8 - TOTAL));
    unsigned char *frag = reinterpret_cast<unsigned char *>(out + FRAG);
    const float *w = packed_fp32;
    float *tab = out + TAB0;
    pack_layer(w + L1_W, OBS, HID, 0, 4, 1, frag + FR_L1, tab + T_SW + O_L1);
    pack_layer(w + L2_W, HID, HID, 0, 4, 4, frag + FR_L2, tab + T_SW + O_L2);
    for (int gate = 0; gate < 3; gate++) {   // M-block index of the GRU matrices: gate * 4 + wave
        pack_layer(w + GI_W, HID, 3 * HID, gate * HID, 4, 4, frag + FR_GI + gate * 4 * MB_BYTES_K4, tab + T_SW + O_GI + gate * HID);
        pack_layer(w + GH_W, HID, 3 * HID, gate * HID, 4, 4, frag + FR_GH + gate * 4 * MB_BYTES_K4, tab + T_SW + O_GH + gate * HID);
    }
    pack_layer(w + A1_W, HID, HID, 0, 4, 4, frag + FR_A1, tab + T_SW + O_A1);
    pack_layer(w + A2_W, HID, HID, 0, 4, 4, frag + FR_A2, tab + T_SW + O_A2);
    std::memcpy(tab + T_BIAS + O_L1, w + L1_B, sizeof(float) * HID);
    std::memcpy(tab + T_BIAS + O_L2, w + L2_B, sizeof(float) * HID);
    std::memcpy(tab + T_BIAS + O_GI, w + GI_B, sizeof(float) * 3 * HID);
    std::memcpy(tab + T_BIAS + O_GH, w + GH_B, sizeof(float) * 3 * HID);
    std::memcpy(tab + T_BIAS + O_A1, w + A1_B, sizeof(float) * HID);
    std::memcpy(tab + T_BIAS + O_A2, w + A2_B, sizeof(float) * HID);
    const int ln_g[6] = {LN0_G, LN1_G, LN2_G, LN3_G, LN4_G, LN5_G}, ln_b[6] = {LN0_B, LN1_B, LN2_B, LN3_B, LN4_B, LN5_B};
    for (int k = 1; k < 6; k++) {
        std::memcpy(tab + T_LN + 256 * (k - 1), w + ln_g[k], sizeof(float) * HID);
        std::memcpy(tab + T_LN + 256 * (k - 1) + 128, w + ln_b[k], sizeof(float) * HID);
    }
    std::memcpy(tab + T_HEAD, w + HD_W, sizeof(float) * 4 * HID);
    std::memcpy(tab + T_HEAD + 4 * HID, w + HD_B, sizeof(float) * 4);
    for (int k = 0; k < 6; k++) {
        float gm = 0.0f, bm = 0.0f;
        for (int j = 0; j < (k == 0 ? OBS : HID); j++) {
            gm = std::fmax(gm, std::fabs(w[ln_g[k] + j]));
            bm = std::fmax(bm, std::fabs(w[ln_b[k] + j]));
        }
        tab[T_LNMAX + 2 * k] = gm;
        tab[T_LNMAX + 2 * k + 1] = bm;
    }
    return 0;
}
