// np_rollout.h — the returns of one rollout (GAE or discounted sums) for the device-resident rollout storage.
//
// Reference: ReplayBuffer.compute_returns (algorithms/utils/buffer.py:139-173), called once per PPO update on buffers of
// shape [buffer_size(+1), n_rollout_threads, num_agents, 1]; there it is a Python loop over time with whole-array float32
// numpy operations.  Here: one lane per column (thread x agent), a backward scan over the T rows in registers; every array is
// touched once (HBM-bound: 16-20 B read and 4 B written per element), consecutive lanes read consecutive floats of a row.
//
// Arithmetic: the reference's, operation by operation (each numpy operation rounds to float32; gamma * gae_lambda is a double
// product rounded once; evaluation order as written there) — bit-exact to the CPU restatement the tests compare with:
//   GAE      td = (r[t] + (g * V[t+1]) * m[t+1]) - V[t];  gae = td + ((gl * m[t+1]) * gae);  [proper: gae *= bad[t+1]];
//            ret[t] = gae + V[t];   V[T] = next_value
//   no GAE   ret[t] = ((ret[t+1] * g) * m[t+1]) + r[t];   ret[T] = next_value
//            proper: ret[t] = (that * bad[t+1]) + ((1 - bad[t+1]) * V[t])
#pragma once
#include <hip/hip_runtime.h>

namespace nproll {

constexpr int THREADS = 256;
// U = time steps whose operands are requested together.  The scan itself is two dependent operations per step, so a lone wave
// (the reference trains with N = 3000 columns and T = 3000) is bound by memory round trips: 32 steps in flight.  With millions
// of columns there are enough waves to cover the latency and the smallest register footprint wins (measured 1 / 8 / 32 steps at
// T = 64, N = 1e6: 5.7 / 5.0 / 5.1 TB/s).
constexpr int U_FEW_COLUMNS = 32, U_MANY_COLUMNS = 1;
constexpr long long MANY_COLUMNS = 500000;

template <bool GAE, bool PROPER, int U>
__global__ __launch_bounds__(THREADS) void returns_kernel(long long T, long long N, float g, float gl, const float *__restrict__ r,
                                                          float *__restrict__ V, const float *__restrict__ m,
                                                          const float *__restrict__ bad, const float *__restrict__ nv,
                                                          float *__restrict__ ret) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // 64-thread workgroups when there are few columns: one wave per CU
    if (j >= N) return;
    float carry = nv[j];   // GAE: V[t+1]; otherwise ret[t+1]
    float gae = 0.0f;
    if (GAE) V[T * N + j] = carry;
    else ret[T * N + j] = carry;
    for (long long hi = T; hi > 0; hi -= U) {       // block of time steps hi-1, hi-2, ..., max(hi-U, 0)
        float rw[U], vv[U], mk[U], bd[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long long t = hi - 1 - u > 0 ? hi - 1 - u : 0;   // clamped: rows below 0 are loaded (in bounds) and not used
            rw[u] = r[t * N + j];
            mk[u] = m[(t + 1) * N + j];
            if (GAE || PROPER) vv[u] = V[t * N + j];
            if (PROPER) bd[u] = bad[(t + 1) * N + j];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long long t = hi - 1 - u;
            if (t < 0) break;
            if (GAE) {
                const float x = g * carry;
                const float y = x * mk[u];
                const float z = rw[u] + y;
                const float td = z - vv[u];
                const float c = gl * mk[u];
                const float d = c * gae;
                gae = td + d;
                if (PROPER) gae = gae * bd[u];
                ret[t * N + j] = gae + vv[u];
                carry = vv[u];
            } else {
                const float x = carry * g;
                const float y = x * mk[u];
                float z = y + rw[u];
                if (PROPER) {
                    const float p = z * bd[u];
                    const float q = 1.0f - bd[u];
                    const float s = q * vv[u];
                    z = p + s;
                }
                ret[t * N + j] = z;
                carry = z;
            }
        }
    }
}

// ---- one collect step into the device-resident rollout storage: F16SimRunner.insert (runner/F16sim_runner.py:131-154: the recurrent states of
// envs that ended are zeroed, masks = 0 where an env is done, bad_masks = 0 where it is bad_done — `any` over the env's agents) followed by
// ReplayBuffer.insert (algorithms/utils/buffer.py:76-112: obs / masks / bad_masks / rnn states into slot step + 1, actions / rewards /
// log-probs / values into slot step) as ONE launch instead of ~20 small torch kernels (9 copies + the mask arithmetic).  Pure data movement.
struct InsertArgs {
    long long E, A, step;            // envs (rollout threads), agents per env, slot
    int obs_dim, act_dim, rnn_dim;   // rnn_dim = recurrent_hidden_layers * recurrent_hidden_size
    float *obs, *actions, *rewards, *masks, *bad_masks, *logp, *values, *rnn_a, *rnn_c;              // the storage, [T(+1)][E * A][...]
    const float *obs_in, *act_in, *rew_in, *logp_in, *val_in, *rnn_a_in, *rnn_c_in;                   // [E * A][...]
    const unsigned char *done, *bad, *tmo;                                                           // [E * A] (bool)
};
constexpr int INSERT_THREADS = 64;   // per row (env, agent): one wave copies its 1.2 KB
__global__ __launch_bounds__(INSERT_THREADS) void insert_kernel(InsertArgs a) {
    const long long row = blockIdx.x, N = a.E * a.A, env0 = (row / a.A) * a.A;
    const int t = threadIdx.x;
    bool d = false, b = false, any = false;
    for (long long k = 0; k < a.A; k++) {   // num_agents is 1 on this path (2 in the combat envs): a short loop
        const bool dk = a.done[env0 + k] != 0, bk = a.bad[env0 + k] != 0;
        d = d || dk; b = b || bk; any = any || dk || bk || a.tmo[env0 + k] != 0;
    }
    const long long s0 = a.step * N + row, s1 = (a.step + 1) * N + row;
    for (int j = t; j < a.obs_dim; j += INSERT_THREADS) a.obs[s1 * a.obs_dim + j] = a.obs_in[row * a.obs_dim + j];
    for (int j = t; j < a.act_dim; j += INSERT_THREADS) a.actions[s0 * a.act_dim + j] = a.act_in[row * a.act_dim + j];
    for (int j = t; j < a.rnn_dim; j += INSERT_THREADS) {
        a.rnn_a[s1 * a.rnn_dim + j] = any ? 0.0f : a.rnn_a_in[row * a.rnn_dim + j];
        a.rnn_c[s1 * a.rnn_dim + j] = any ? 0.0f : a.rnn_c_in[row * a.rnn_dim + j];
    }
    if (t == 0) {
        a.rewards[s0] = a.rew_in[row];
        a.logp[s0] = a.logp_in[row];
        a.values[s0] = a.val_in[row];
        a.masks[s1] = d ? 0.0f : 1.0f;
        a.bad_masks[s1] = b ? 0.0f : 1.0f;
    }
}

}  // namespace nproll
