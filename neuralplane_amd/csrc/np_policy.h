// np_policy.h — arguments of the rollout policy's inference launch (np_policy.hip); the C ABI entry np_policy_act fills them.
#pragma once
#include <hip/hip_runtime.h>

namespace nppol {

struct ActArgs {
    const float *w[2];      // packed networks: [0] actor, [1] critic (value head in column 0 of the head block)
    const float *h_in[2];   // recurrent states [n][128]
    float *h_out[2];
    const float *obs, *mask, *noise;
    float *values, *actions, *log_probs;
    long long n;
    int act_dim, flags, first_net, obs_dim;   // obs_dim: 22 or 15
    float std[4], log_std[4];
    // optional: the flags [3][n] of the env step that produced `obs` — the runner's insert rule applied on the fly (see np_policy_step.prev_flags)
    const unsigned char *prev;
    float *masks_out, *bad_masks_out;
};

hipError_t launch_policy_act(const ActArgs &a, hipStream_t stream);

}  // namespace nppol

namespace npact8 {
hipError_t launch_policy_act_i8(const nppol::ActArgs &a, hipStream_t stream);   // np_actor_i8.hip: both networks in the block-fixed-point numerics
}

// the library's thread-local error string (np_last_error), for the entry points that live outside np_f16_kernels.hip; returns 1
int np_internal_fail(const char *msg);
