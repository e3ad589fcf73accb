// np_policy.hip — the rollout policy's inference step (PPOPolicy.get_actions, reference algorithms/ppo/ppo_policy.py:26-32) as ONE launch:
// the actor (ppo_actor.py:38-64, sampled actions + log-probabilities) and the critic (ppo_critic.py:38-50) of the shapes the training
// scripts build (scripts/train_heading.sh:17: "128 128" everywhere, GRU 128 — the frozen controller's) on the same observation, one
// 32-row tile per workgroup and network (grid.y: 0 = actor, 1 = critic) through the controller's fp32 matrix-core body (np_actor.h).
// Numerics: the ordered chains of np_actor.h up to the output layer (the CPU restatement f16_actor.inc states them), then
//   mean = tanh(mu); a = fl(fl(eps * std) + mean)                       torch.normal(mean, std): normal_(0, 1).mul_(std).add_(mean)
//   lp_j = (-(d * d)) / (2 * (std * std)) - log_std - 0.9189385f, d = a - mean;   log-prob = ((lp_0 + lp_1) + lp_2) + lp_3
//   value = the critic's head (column 0 of its head block), no activation
// eps is handed in (the caller draws it with its own generator, as the reference's sample() does).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/neuralplane_amd.h"
#define NPACT_NO_KERNELS 1
#include "np_actor.h"
#include "np_policy.h"

namespace nppol {

using namespace npact;

template <int NOBS>   // observations per row: 22 (control / heading / tracking) or 15 (the 1v1 combat env)
__global__ __launch_bounds__(MTHREADS, 2) void policy_act_kernel(const ActArgs a) {
    __shared__ float lds[ACTOR32_LDS_FLOATS];
    const unsigned tid = threadIdx.x;
    const int net = __builtin_amdgcn_readfirstlane(a.first_net + (int)blockIdx.y);   // 0 = actor, 1 = critic
    const float *weights = a.w[net];
    const float *h_in = a.h_in[net];
    float *h_out = a.h_out[net];
    const long long n = a.n;
    const int lane = (int)(tid % TILE), row = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(tid / TILE));
    const int blk = 2 * wave + hi;
    const long long i = (long long)blockIdx.x * T32 + row;
    const bool valid = i < n;
    const long long ic = valid ? i : n - 1;
    Actor32Pre pre;
    actor32_request_l1(weights, tid, pre);
    // the recurrent state's mask: masks[i], or — with the previous env step's flags — the runner's insert rule applied here
    float mk;
    bool ended = false;
    if (a.prev) {
        const bool d = a.prev[ic] != 0, b = a.prev[n + ic] != 0;
        ended = d || b || a.prev[2 * n + ic] != 0;
        mk = 1.0f;   // done implies ended: the state is zero there anyway
        if (valid && net == a.first_net && wave == 0 && hi == 0) {
            a.masks_out[i] = d ? 0.0f : 1.0f;
            a.bad_masks_out[i] = b ? 0.0f : 1.0f;
        }
    } else {
        mk = a.mask[ic];
    }
    float hm[BLK];
    {
        const float4 *hp = reinterpret_cast<const float4 *>(h_in + ic * HID + blk * BLK);
#pragma unroll
        for (int j = 0; j < BLK / 4; j++) {
            const float4 q = hp[j];
            hm[4 * j] = ended ? 0.0f : q.x * mk;
            hm[4 * j + 1] = ended ? 0.0f : q.y * mk;
            hm[4 * j + 2] = ended ? 0.0f : q.z * mk;
            hm[4 * j + 3] = ended ? 0.0f : q.w * mk;
        }
        if (ended && valid) {   // the stored state of an env that ended is zero (F16SimRunner.insert): written back in place
            float4 *hz = const_cast<float4 *>(hp);
#pragma unroll
            for (int j = 0; j < BLK / 4; j++) hz[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    }
    actor32_stage_head(lds, weights, tid);
    float xr[OBS];
#pragma unroll
    for (int j = 0; j < NOBS; j++) xr[j] = a.obs[ic * NOBS + j];
    float hn[BLK], mu;
    actor32_body<false, NOBS>(lds, weights, pre, xr, hm, hn, mu, tid);   // mu of (row, head column = wave); lanes with hi == 0 hold it
    if (net == 0) {
        float *lp = lds;   // bufA: nobody reads it after the last layer's barrier
        const int A = a.act_dim;
        if (hi == 0 && wave < A) {
            const float mean = act_tanh(mu);
            float act = mean;
            if (!(a.flags & NP_POLICY_DETERMINISTIC)) {
                const float e = a.noise[ic * A + wave] * a.std[wave];
                act = e + mean;
            }
            const float d = act - mean;
            const float q = -(d * d);
            const float var = a.std[wave] * a.std[wave];
            float t = q / (2.0f * var);
            t = t - a.log_std[wave];
            t = t - 0.9189385f;
            lp[wave * T32 + row] = t;
            if (valid) a.actions[i * A + wave] = act;
        }
        __syncthreads();
        if (wave == 0 && hi == 0 && valid) {
            float s = lp[row];
            for (int j = 1; j < A; j++) s = s + lp[j * T32 + row];
            a.log_probs[i] = s;
        }
    } else if (wave == 0 && hi == 0 && valid) {
        a.values[i] = mu;
    }
    if (valid) {
        float4 *hq = reinterpret_cast<float4 *>(h_out + i * HID + blk * BLK);
#pragma unroll
        for (int j = 0; j < BLK / 4; j++) hq[j] = make_float4(hn[4 * j], hn[4 * j + 1], hn[4 * j + 2], hn[4 * j + 3]);
    }
}

hipError_t launch_policy_act(const ActArgs &a, hipStream_t stream) {
    const int nets = ((a.flags & NP_POLICY_ACTOR) ? 1 : 0) + ((a.flags & NP_POLICY_CRITIC) ? 1 : 0);
    const dim3 grid((unsigned)((a.n + T32 - 1) / T32), (unsigned)nets);
    if (a.obs_dim == 15)
        hipLaunchKernelGGL(policy_act_kernel<15>, grid, dim3(MTHREADS), 0, stream, a);
    else
        hipLaunchKernelGGL(policy_act_kernel<OBS>, grid, dim3(MTHREADS), 0, stream, a);
    return hipGetLastError();
}

}  // namespace nppol
