// np_f16_kernels.hip — fused F-16 env.step / env.reset kernels for gfx950 and the C ABI around them
// (include/neuralplane_amd.h).  One kernel launch per BaseEnv.step (reference: envs/env_base.py:99-109).
//
// Mapping: one lane per aircraft, SoA state in HBM so every state load/store is a coalesced dword access; the [n][22]
// observation rows are transposed through LDS and stored as coalesced dwords.  The MLP weights are wave-uniform
// __constant__ data streamed through the scalar unit (s_load -> SGPR-pair operands of v_pk_fma_f32, np_mlp_asm.inc).
// Two variants of the same kernel: 128-thread workgroups of two independent waves (throughput), or four waves sharing
// one tile of 64 aircraft and splitting the net evaluations (latency, small batches) — bit-identical results.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see neuralplane_amd/build.py).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/neuralplane_amd.h"
#include "np_f16_device.h"
#include "np_f16_kargs.h"
#include "np_f16_combat.h"
#include "np_actor.h"
#include "np_rollout.h"
#include "np_policy.h"
#include "np_planning.h"
namespace npact8 {
hipError_t launch_actor_i8(const float *weights, long long n, const float *obs, const float *h_in, const float *masks, float *actions, float *h_out,
                           hipStream_t stream);   // np_actor_i8.hip
}
#include "np_dispatch.h"
#include "np_env_launch.h"

namespace npf16 {

// f16_env_kernel<TASK, SOLVER, STEP, CACHED, TILE, WPT, INNER, PW>: np_f16_env_kernel.h (instantiated by np_env_t*s*.hip, launched through
// np_env_launch.h::env_dispatch — one translation unit per task x solver so that the library builds in parallel)


// F16Model getters that need the dynamics or the atmosphere (F16_model.py:47-49, 132-198): out[23][ld_out]
__global__ __launch_bounds__(BLOCK) void f16_derived_kernel(const float *__restrict__ sp, const float *__restrict__ up,
                                                            long long ld, float *__restrict__ out, long long ld_out,
                                                            long long n, float airspeed, int tables, AeroWeights wt, Airframe af) {
    __shared__ float lds[NUM_LDS_SLOTS * BLOCK];
    float *coef = lds + threadIdx.x;
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    const bool valid = i < n;
    const long long ic = valid ? i : n - 1;  // keep control flow wave-uniform (weights stay in SGPRs)
    float s[12], u[4];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = sp[k * ld + ic];
#pragma unroll
    for (int k = 0; k < 4; k++) u[k] = up[k * ld + ic];
    Trig tr;
    float tt, spsi, cpsi;
    trig_of(s, tr, tt);
    np_sincos(s[5], spsi, cpsi);
    float xd[12];
    nlplant<true, AB_ALL, BLOCK>(wt, af, s, u, tr, tt, spsi, cpsi, coef, tables != 0, xd);
    float a3[3];
    body_acceleration(s, tr, xd, a3);
    const float inv_grav = (float)(1.0 / 32.174), minv_grav = (float)(-1.0 / 32.174);  // F16_model.py:166,176-178
    const float nx = inv_grav * a3[0] + tr.st;
    const float ny = inv_grav * a3[1] - tr.ct * tr.sphi;
    const float nz = minv_grav * a3[2] + tr.ct * tr.cphi;
    const float e2t = eas2tas_of(af, s[2]);
    const float eas = (s[6] + airspeed * 1.0f) / e2t;
    // F16Model.get_atmos (F16_model.py:183-198 == F16Dynamics.atmos, F16_dynamics.py:22-35): mach, qbar, ps
    const float tfac = 1.0f - af.atm_lapse * s[2];
    float temp = 519.0f * tfac;
    temp = (s[2] >= 35000.0f ? 1.0f : 0.0f) * 390.0f + (s[2] < 35000.0f ? 1.0f : 0.0f) * temp;
    const float rho = af.rho0 * np_pow_posexp(tfac, af.atm_exp);
    const float mach = s[6] / sqrtf((float)(1.4 * 1716.3) * temp);
    const float qbar = (0.5f * rho) * (s[6] * s[6]);
    float ps = (1715.0f * rho) * temp;
    ps = (ps == 0.0f ? 1.0f : 0.0f) * 1715.0f + (ps != 0.0f ? 1.0f : 0.0f) * ps;
    if (!valid) return;
#pragma unroll
    for (int k = 0; k < 12; k++) out[k * ld_out + i] = xd[k];
#pragma unroll
    for (int k = 0; k < 3; k++) out[(12 + k) * ld_out + i] = a3[k];
    out[15 * ld_out + i] = nx;
    out[16 * ld_out + i] = ny;
    out[17 * ld_out + i] = nz;
    out[18 * ld_out + i] = e2t;
    out[19 * ld_out + i] = eas;
    out[20 * ld_out + i] = mach;
    out[21 * ld_out + i] = qbar;
    out[22 * ld_out + i] = ps;
}

// hifi_F16.hifi_C / hifi_damping / hifi_C_lef / hifi_damping_lef / hifi_rudder / hifi_ailerons / hifi_other_coeffs
// (envs/models/F16/hifi_F16_AeroData.py:745-822) for arbitrary (alpha, beta, el) in degrees: the same normalisation and net
// bodies nlplant runs (so what this returns IS what the step kernels use), out[43][ld_out] in the reference's evaluation order
// (np_nets.h::NetId).  Row N_dCzq_lef (the net nlplant never reads, F16_dynamics.py:199) is not part of the device weights: 0.
__global__ __launch_bounds__(BLOCK) void f16_aero_kernel(const float *__restrict__ alpha_deg, const float *__restrict__ beta_deg,
                                                         const float *__restrict__ el, long long n, float *__restrict__ out,
                                                         long long ld_out, int tables, AeroWeights wt) {
    __shared__ float lds[NUM_LDS_SLOTS * BLOCK];
    float *coef = lds + threadIdx.x;
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    const bool valid = i < n;
    const long long ic = valid ? i : n - 1;  // wave-uniform control flow: the weights stay in SGPRs
    const float a = alpha_deg[ic], b = beta_deg[ic], e = el[ic];
    float xn[NUM_NORM_GROUPS];
    normalise_inputs(wt, a, b, e, xn);
    const float chk = ((a - a) + (b - b)) + (e - e);  // numerics spec: non-finite inputs poison every coefficient
    const bool ok = (chk == chk);
    eval_nets<BLOCK, AB_ALL, true>(wt, xn, coef, tables != 0);
    if (!valid) return;
    const float qnan = __builtin_nanf("");
#pragma unroll
    for (int net = 0; net < NUM_NETS; net++) {
        const int slot = slot_of(net);
        out[net * ld_out + i] = slot < 0 ? 0.0f : (ok ? coef[slot * BLOCK] : qnan);
    }
}

// PlanningEnv.low_level_obs (planning_env.py:60-142): ControlTask-style observation for caller-supplied targets, no noise
// ACTION = false: targets from tp[3][ld].  ACTION = true (PlanningEnv.step's prelude, planning_env.py:146-152): targets = (pitch, heading,
// vt) + clamp(action, -1, 1) * (0.3, 0.3, 30) — one product and one sum per element as the reference's torch expressions — written to
// tgt_out[3][ld] as well
template <bool ACTION>
__global__ __launch_bounds__(BLOCK) void f16_lowlevel_obs_kernel(const float *__restrict__ sp, const float *__restrict__ up,
                                                                 const float *__restrict__ tp, long long ld, float *__restrict__ obs,
                                                                 long long n, DevCfg cfg, long long act_stride, float *__restrict__ tgt_out) {
    __shared__ float tile[BLOCK * OBS_LD];
    const int t = threadIdx.x;
    const long long i0 = (long long)blockIdx.x * BLOCK, i = i0 + t;
    const long long ic = i < n ? i : n - 1;
    float s[12], u[4], tgt[3];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = sp[k * ld + ic];
#pragma unroll
    for (int k = 0; k < 4; k++) u[k] = up[k * ld + ic];
    if constexpr (ACTION) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float a = tp[ic * act_stride + k];
            a = a < -1.0f ? -1.0f : a;   // torch.clamp: NaN stays NaN
            a = a > 1.0f ? 1.0f : a;
            tgt[k] = s[4 + k] + a * (k == 2 ? 30.0f : 0.3f);
            if (i < n) tgt_out[k * ld + i] = tgt[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 3; k++) tgt[k] = tp[k * ld + ic];
    }
    Trig tr;
    float tt;
    trig_of(s, tr, tt);
    float o[22];
    observe<1>(cfg, s, u, tgt, tr, o);
#pragma unroll
    for (int k = 0; k < 22; k++) tile[t * OBS_LD + k] = o[k];
    __syncthreads();
    const long long rows = (n - i0) < BLOCK ? (n - i0) : BLOCK;
    const int total = (int)rows * 22;
    float *dst = obs + i0 * 22;
#pragma unroll
    for (int it = 0; it < 22; it++) {
        const int L = it * BLOCK + t;
        if (L < total) {
            const int r = L / 22, c = L - r * 22;
            dst[L] = tile[r * OBS_LD + c];
        }
    }
}

// cached coefficients of a reset aircraft (alpha = beta = 0) -> out[14]; run once per context
__global__ __launch_bounds__(BLOCK) void f16_reset_coef_kernel(float *out, int tables, AeroWeights wt) {
    __shared__ float lds[NUM_LDS_SLOTS * BLOCK];
    float *coef = lds + threadIdx.x;
    float xn[NUM_NORM_GROUPS];
    const float r2d = (float)(180.0 / 3.141592653589793);
    normalise_inputs(wt, 0.0f * r2d, 0.0f * r2d, 0.0f, xn);
    eval_ab<BLOCK, AB_FORCE>(wt, xn, coef, tables != 0);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NUM_CACHED; k++) out[k] = coef[cached_slot(k) * BLOCK];
    }
}

// Self-check of the numerics spec's constant division on THIS device: np_divc(x, c) against the IEEE x / c for every one of the
// 2^32 bit patterns of x.  counts[0]: mismatches whose IEEE quotient is a normal number with |x| >= 2^-100 (the spec promises
// none), counts[1]: mismatches with a denormal / underflowing quotient or |x| < 2^-100 (allowed: the last place may differ),
// counts[2]: inputs compared.
__global__ __launch_bounds__(256) void divc_sweep_kernel(float c, float rc, unsigned long long *counts) {
    unsigned long long bad = 0, soft = 0, seen = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long b = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b < (1ull << 32); b += stride) {
        const float x = __uint_as_float((unsigned)b);
        const float got = np_divc(x, c, rc), want = x / c;
        seen++;
        const bool same = (__float_as_uint(got) == __float_as_uint(want)) || (got != got && want != want);
        if (!same) {
            const float aw = fabsf(want), ax = fabsf(x);
            if (aw >= 1.17549435e-38f && aw <= 3.402823466e38f && ax >= 7.888609052e-31f) bad++;  // 2^-100
            else soft++;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        bad += __shfl_down(bad, off);
        soft += __shfl_down(soft, off);
        seen += __shfl_down(seen, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(counts + 0, bad);
        atomicAdd(counts + 1, soft);
        atomicAdd(counts + 2, seen);
    }
}

}  // namespace npf16

// =================================================================================================
// host side: C ABI
// =================================================================================================
using namespace npf16;

static thread_local std::string g_err;
static int fail(const std::string &msg) {
    g_err = msg;
    return 1;
}
#define NP_HIP(call)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) return fail(std::string(#call) + ": " + hipGetErrorString(e_));         \
    } while (0)

struct np_f16_ctx {
    int device;
    int task, solver;
    DevCfg cfg;
    float *d_reset_coef;  // [NUM_CACHED] device buffer owned by the context
    float *d_weights;     // KBLOB | PWL tables | PWL un-normalisation, one device allocation owned by the context
    AeroWeights wt;
    int variant;          // NP_KERNEL_AUTO / _LATENCY / _THROUGHPUT / _PAIR / _LATENCY8
    bool combat;  // created by np_f16_combat_ctx_create: only the combat entry points accept it
    CombatDevCfg ccfg;
    bool timing;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;  // recorded, not yet read
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    double t_sum_ms;
    int64_t t_count;
    std::vector<float> samples;  // per-launch durations in launch order since np_f16_set_timing(ctx, 1)
    unsigned long long *trace;   // np_f16_set_trace
    int64_t trace_cap;
    // np_planning_inner_loop: streams of row groups 1.. and the fork / join events (created on first use)
    std::vector<hipStream_t> group_streams;
    std::vector<hipEvent_t> group_events;  // [0] fork, [1 + g] join of group g + 1
    // persistent PlanningEnv kernel (np_planning.hip): the (tile, iteration) queue words, owned by the context, grown on demand
    unsigned *d_queue;
    int64_t queue_cap;
    unsigned queue_next = 0, flag_base = 0;   // where the item counter stands / the base of the progress words for the next launch
    bool queue_dirty = true;
    // bounded waits of the guest / queue schedules (round 5): the launch's sticky error word (device), the record the kernel leaves in pinned
    // host memory when a wait expires, the snapshot of a macro-step's inputs (check = sync: restored before NP_E_PLANNING_STALLED is returned)
    // and the event behind the last unchecked launch (check = deferred)
    unsigned *d_plan_err = nullptr, *h_plan_err = nullptr, *h_plan_err_dev = nullptr;
    char *d_snap = nullptr;
    size_t snap_cap = 0;
    hipEvent_t plan_done = nullptr;
    bool plan_unchecked = false;
    const char *plan_unchecked_mode = "";
    int num_cus;  // multiProcessorCount of the context's device
};

namespace {

#pragma pack(push, 1)
struct BlobRec {  // NPF16MLP v1 record (tools/export_weights.py)
    char name[24];
    uint32_t input_mask, n_linear;
    uint32_t dims[6];
    double in_mean[3], in_std[3];
    double out_mean, out_std;
    uint32_t param_offset, n_params;
};
#pragma pack(pop)
static_assert(sizeof(BlobRec) == 128, "blob record is 128 bytes");

// asset blob (torch layout W[out][in]) -> kernel-order blob (np_nets.h)
int pack_kblob(const void *blob, size_t nbytes, std::vector<float> &kb, std::vector<float> &kbd, std::vector<float> &pwl,
               std::vector<float> &pwl_unnorm) {
    const unsigned char *p = (const unsigned char *)blob;
    if (!p || nbytes < 16 || std::memcmp(p, "NPF16MLP", 8) != 0) return fail("weights blob: bad magic");
    uint32_t ver, nn;
    std::memcpy(&ver, p + 8, 4);
    std::memcpy(&nn, p + 12, 4);
    if ((ver != 1 && ver != 2) || nn != NUM_NETS) return fail("weights blob: unsupported version / net count");
    const size_t hdr = 16 + (size_t)nn * sizeof(BlobRec);
    if (nbytes < hdr) return fail("weights blob: truncated header");
    const size_t pfloats = (nbytes - hdr) / 4;
    std::vector<float> par(pfloats);
    std::memcpy(par.data(), p + hdr, pfloats * 4);
    kb.assign(KBLOB_FLOATS, 0.0f);
    bool grp_set[NUM_NORM_GROUPS] = {};
    for (int cl = 0; cl < NUM_CLASSES; cl++) {
        const NetClass c = CLASSES[cl];
        const int hid[3] = {c.h1, c.h2, c.h3};
        const int n_hidden = c.h3 > 0 ? 3 : 2;
        for (int m = 0; m < c.count; m++) {
            const int net = c.nets[m];
            BlobRec r;
            std::memcpy(&r, p + 16 + (size_t)net * sizeof(BlobRec), sizeof(r));
            const std::string nm(r.name, strnlen(r.name, sizeof(r.name)));
            if ((int)r.n_linear != n_hidden + 1 || (int)r.dims[0] != c.n_in) return fail("weights blob: shape of net " + nm + " does not match its class");
            for (int l = 0; l < n_hidden; l++)
                if ((int)r.dims[l + 1] != hid[l]) return fail("weights blob: hidden widths of net " + nm + " do not match its class");
            if ((int)r.n_params != class_params(c)) return fail("weights blob: parameter count of net " + nm);
            if ((size_t)r.param_offset + r.n_params > pfloats) return fail("weights blob: truncated parameters");
            // input slot k of the net <- k-th set bit of input_mask (bit0 alpha, bit1 beta, bit2 el)
            int slot = 0;
            for (int k = 0; k < 3; k++) {
                if (!(r.input_mask & (1u << k))) continue;
                if (slot >= c.n_in) return fail("weights blob: input mask of net " + nm);
                const int g = c.grp[slot++];
                const bool kind_ok = (k == 0 && g >= G_A_C && g <= G_A_RUD) || (k == 1 && (g == G_B_C || g == G_B_O)) ||
                                     (k == 2 && (g == G_E_C || g == G_E_ETA));
                if (!kind_ok) return fail("weights blob: input kind of net " + nm + " does not match its class");
                const float mean = (float)r.in_mean[k], sd = (float)r.in_std[k];  // torch rounds the CSV doubles to fp32
                if (!grp_set[g]) {
                    kb[KBLOB_NORM_STRIDE * g] = mean;
                    kb[KBLOB_NORM_STRIDE * g + 1] = sd;
                    kb[KBLOB_NORM_STRIDE * g + 2] = (float)(1.0 / (double)sd);  // np_divc: the normalisation divides by a per-context constant
                    grp_set[g] = true;
                } else if (kb[KBLOB_NORM_STRIDE * g] != mean || kb[KBLOB_NORM_STRIDE * g + 1] != sd) {
                    return fail("weights blob: normalisation of net " + nm + " differs from its class");
                }
            }
            if (slot != c.n_in) return fail("weights blob: input mask of net " + nm);
            const float *src = par.data() + r.param_offset;
            float *dst = kb.data() + class_base(cl) + (size_t)m * class_stride(cl);
            int in = c.n_in;
            // Hidden activations are carried divided by 2^ACT_SHIFT (np_nets.h): the ReLU is then the `clamp` output modifier
            // of the layer's last FMA, which costs no instruction.  Powers of two commute with every rounding, so the first
            // layer's weights and all hidden biases are stored x 2^-ACT_SHIFT and the output layer's weights x 2^+ACT_SHIFT —
            // checked here to be exact (normal, finite) for every parameter of the asset.
            bool exact = true;
            auto scaled = [&exact](float w, int e) {
                const float v = std::ldexp(w, e);
                if (w != 0.0f && !(std::isnormal(v) && std::ldexp(v, -e) == w)) exact = false;
                return v;
            };
            for (int l = 0; l < n_hidden; l++) {  // hidden layers: bias row + `in` weight rows, rows padded to even
                const int out = hid[l], row = pad2(out);
                const float *W = src, *bias = src + (size_t)in * out;
                for (int j = 0; j < out; j++) dst[j] = scaled(bias[j], -ACT_SHIFT);
                for (int k = 0; k < in; k++)
                    for (int j = 0; j < out; j++) dst[row * (k + 1) + j] = l == 0 ? scaled(W[j * in + k], -ACT_SHIFT) : W[j * in + k];
                src += (size_t)in * out + out;
                dst += (size_t)row * (in + 1);
                in = out;
            }
            {  // output layer in -> 1: (bias, 0), then W[0][0..in) padded to even (two interleaved partial chains, np_nets.h)
                const float *W = src, *bias = src + in;
                dst[0] = bias[0];
                dst[1] = 0.0f;
                for (int k = 0; k < in; k++) dst[2 + k] = scaled(W[k], ACT_SHIFT);
                dst += 2 + pad2(in);
            }
            if (!exact) return fail("weights blob: a parameter of net " + nm + " cannot be rescaled by 2^ACT_SHIFT exactly");
            dst[0] = (float)r.out_std;
            dst[1] = (float)r.out_mean;
        }
    }
    for (int g = 0; g < NUM_NORM_GROUPS; g++)
        if (!grp_set[g]) return fail("weights blob: a normalisation group is unused");
    // the same records in the layout of the two-set bodies (np_nets.h::dual_record_len): pure re-arrangement of `kb`
    kbd.assign(KBLOB_DUAL_FLOATS, 0.0f);
    for (int cl = 0; cl < NUM_CLASSES; cl++) {
        const NetClass c = CLASSES[cl];
        const int hid[3] = {c.h1, c.h2, c.h3};
        const int n_hidden = c.h3 > 0 ? 3 : 2;
        for (int m = 0; m < c.count; m++) {
            const float *src = kb.data() + class_base(cl) + (size_t)m * class_stride(cl);
            float *dst = kbd.data() + dual_class_base(cl) + (size_t)m * dual_class_stride(cl);
            int in = c.n_in;
            for (int l = 0; l < n_hidden; l++) {
                const int out = hid[l], row = pad2(out);
                for (int j = 0; j < out; j++) {
                    dst[2 * j] = src[row + j];  // W[j][0]
                    dst[2 * j + 1] = src[j];    // bias[j]
                }
                for (int k = 1; k < in; k++)
                    for (int j = 0; j < out; j++) dst[2 * out + (k - 1) * out + j] = src[row * (k + 1) + j];
                src += (size_t)row * (in + 1);
                dst += pad2(out * (in + 1));
                in = out;
            }
            dst[0] = src[2];  // W[0][0]
            dst[1] = src[0];  // bias
            for (int k = 1; k < in; k++) dst[1 + k] = src[2 + k];
            src += 2 + pad2(in);
            dst += pad2(in + 1);
            dst[0] = src[0];  // out_std
            dst[1] = src[1];  // out_mean
        }
    }
    // PWL section (blob v2): "PWL1", n_tables, seg_cap, then {net_index, n_segments, t[64], a[64], x0[64], c[64]}
    pwl.clear();
    pwl_unnorm.clear();
    if (ver >= 2) {
        size_t n_par = 0;
        for (int i = 0; i < NUM_NETS; i++) {
            BlobRec r;
            std::memcpy(&r, p + 16 + (size_t)i * sizeof(BlobRec), sizeof(r));
            n_par = std::max(n_par, (size_t)r.param_offset + r.n_params);
        }
        const unsigned char *q = p + hdr + n_par * 4, *end = p + nbytes;
        uint32_t nt, cap;
        if (q + 12 > end || std::memcmp(q, "PWL1", 4) != 0) return fail("weights blob: PWL section missing");
        std::memcpy(&nt, q + 4, 4);
        std::memcpy(&cap, q + 8, 4);
        if (nt != NUM_PWL_TABLES || cap != PWL_SEG) return fail("weights blob: PWL section shape");
        q += 12;
        pwl.assign((size_t)NUM_PWL_TABLES * PWL_TABLE_FLOATS, 0.0f);
        pwl_unnorm.assign((size_t)NUM_PWL_TABLES * 2, 0.0f);
        for (uint32_t k = 0; k < nt; k++) {
            uint32_t idx, nseg;
            if (q + 8 + PWL_TABLE_FLOATS * 4 > end) return fail("weights blob: truncated PWL section");
            std::memcpy(&idx, q, 4);
            std::memcpy(&nseg, q + 4, 4);
            if (idx >= (uint32_t)NUM_NETS || pwl_index((int)idx) != (int)k || nseg < 1 || nseg > (uint32_t)PWL_SEG)
                return fail("weights blob: PWL table order does not match the kernel's");
            std::memcpy(pwl.data() + (size_t)k * PWL_TABLE_FLOATS, q + 8, PWL_TABLE_FLOATS * 4);
            BlobRec r;
            std::memcpy(&r, p + 16 + (size_t)idx * sizeof(BlobRec), sizeof(r));
            pwl_unnorm[2 * k] = (float)r.out_std;
            pwl_unnorm[2 * k + 1] = (float)r.out_mean;
            q += 8 + PWL_TABLE_FLOATS * 4;
        }
    }
    return 0;
}

// np_f16_airframe -> the fp32 constants of the kernels.  Every value is rounded where the literal expression of the reference rounds it:
// plain literals are Python doubles that enter fp32 tensor arithmetic (one rounding), derived constants are folded in double first
// (F16_dynamics.py:221-227 evaluates e.g. Jz * (Jz - Jy) + Jxz ** 2 in Python before it meets a tensor).
np_f16_airframe airframe_defaults() {
    np_f16_airframe a = {};
    a.g = 32.17; a.mass = 636.94; a.B = 30.0; a.S = 300.0; a.cbar = 11.32; a.xcgr = 0.35; a.xcg = 0.30; a.Heng = 0.0;
    a.Jy = 55814.0; a.Jxz = 982.0; a.Jz = 63100.0; a.Jx = 9496.0;
    a.ail_ref = 21.5; a.rud_ref = 30.0;
    a.atm_lapse = 0.703e-5; a.atm_exp = 4.14; a.rho0 = 2.377e-3;
    a.lag_keep = 0.9; a.lag_new = 0.1; a.thrust_frac = 0.225; a.thrust_max = 76300.0; a.thrust_unit = 0.3048;
    a.surf_max[0] = a.surf_max[1] = a.surf_max[2] = 45.0;
    return a;
}

bool airframe_is_zero(const np_f16_airframe &a) {
    const unsigned char *p = reinterpret_cast<const unsigned char *>(&a);
    for (size_t k = 0; k < sizeof(a); k++)
        if (p[k]) return false;
    return true;
}

const char *airframe_error(const np_f16_airframe &a) {
    const double pos[] = {a.mass, a.B, a.S, a.cbar, a.Jy, a.Jz, a.Jx, a.ail_ref, a.rud_ref, a.rho0, a.thrust_unit, a.atm_exp};
    for (double v : pos)
        if (!(v > 0.0) || !std::isfinite(v)) return "airframe: mass, B, S, cbar, Jx, Jy, Jz, ail_ref, rud_ref, rho0, thrust_unit and atm_exp must be positive and finite";
    if (!(a.Jx * a.Jz - a.Jxz * a.Jxz > 0.0)) return "airframe: Jx Jz - Jxz^2 must be positive";
    const double fin[] = {a.g, a.xcgr, a.xcg, a.Heng, a.Jxz, a.atm_lapse, a.atm_exp, a.lag_keep, a.lag_new, a.thrust_frac, a.thrust_max, a.surf_max[0], a.surf_max[1], a.surf_max[2]};
    for (double v : fin)
        if (!std::isfinite(v)) return "airframe: non-finite value";
    return nullptr;
}

Airframe make_airframe(const np_f16_airframe &in) {
    const np_f16_airframe a = airframe_is_zero(in) ? airframe_defaults() : in;
    auto rcp = [](float c) { return (float)(1.0 / (double)c); };   // NP_RCP_CONST: RN(1 / c) of the fp32 divisor
    Airframe d;
    d.g = (float)a.g; d.mass = (float)a.mass; d.r_mass = rcp(d.mass); d.B = (float)a.B; d.S = (float)a.S; d.cbar = (float)a.cbar; d.Heng = (float)a.Heng;
    d.Jy = (float)a.Jy; d.r_Jy = rcp(d.Jy); d.Jxz = (float)a.Jxz; d.Jz = (float)a.Jz; d.Jx = (float)a.Jx;
    d.xc = (float)(a.xcgr - a.xcg);
    d.cbar_over_B = (float)(a.cbar / a.B);
    d.c1 = (float)(a.Jz * (a.Jz - a.Jy) + a.Jxz * a.Jxz);
    d.c2 = (float)(a.Jxz * (a.Jx - a.Jy + a.Jz));
    d.c3 = (float)(a.Jz - a.Jx);
    d.c4 = (float)(a.Jx * (a.Jx - a.Jy) + a.Jxz * a.Jxz);
    d.denom = (float)(a.Jx * a.Jz - a.Jxz * a.Jxz);
    d.r_denom = rcp(d.denom);
    d.ail_ref = (float)a.ail_ref; d.r_ail_ref = rcp(d.ail_ref); d.rud_ref = (float)a.rud_ref; d.r_rud_ref = rcp(d.rud_ref);
    d.atm_lapse = (float)a.atm_lapse; d.atm_exp = (double)(float)a.atm_exp; d.rho0 = (float)a.rho0; d.pad_ = 0.0f;
    d.lag_keep = (float)a.lag_keep; d.lag_new = (float)a.lag_new; d.thrust_frac = (float)a.thrust_frac; d.thrust_max = (float)a.thrust_max;
    d.thrust_unit = (float)a.thrust_unit; d.r_thrust_unit = rcp(d.thrust_unit);
    for (int k = 0; k < 3; k++) d.surf_max[k] = (float)a.surf_max[k];
    return d;
}

DevCfg make_devcfg(const np_f16_cfg &c) {
    DevCfg d;
    d.af = make_airframe(c.airframe);
    d.dt = (float)c.dt - 0.0f;  // t = tensor([0., dt]); dt = t1 - t0   (F16_model.py:66)
    d.airspeed = (float)c.airspeed;
    d.noise_scale = (float)c.noise_scale;
    d.altitude_limit = (float)c.altitude_limit;
    d.acceleration_limit = (float)c.acceleration_limit;
    d.max_velocity = (float)c.max_velocity;
    d.min_velocity = (float)c.min_velocity;
    d.min_alpha = (float)c.min_alpha;
    d.max_alpha = (float)c.max_alpha;
    d.min_beta = (float)c.min_beta;
    d.max_beta = (float)c.max_beta;
    d.max_check_interval = c.max_check_interval;
    d.min_check_interval = c.min_check_interval;
    d.init_T = (float)c.init_T;
    d.alt_span = (float)(c.max_altitude - c.min_altitude);  // Python arithmetic first, then one rounding
    d.min_altitude = (float)c.min_altitude;
    d.vt_span = (float)(c.max_vt - c.min_vt);
    d.min_vt = (float)c.min_vt;
    d.max_heading_increment = (float)c.max_heading_increment;
    d.max_pitch_increment = (float)c.max_pitch_increment;
    d.max_velocities_u_increment = (float)c.max_velocities_u_increment;
    d.dist_span = (float)(c.max_distance - c.min_distance);
    d.min_distance = (float)c.min_distance;
    d.aero_1d_tables = c.aero_1d_tables ? 1 : 0;
    return d;
}

struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t enter(int dev) {
        hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess) return e;
        if (prev != dev) {
            e = hipSetDevice(dev);
            switched = (e == hipSuccess);
        }
        return e;
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

// Turn the pending event pairs into durations (waits for the launches they belong to).  The sample list is for short
// measurement windows: beyond MAX_TIMING_SAMPLES only the running sum / count keep growing.
constexpr size_t MAX_TIMING_SAMPLES = 1u << 20, MAX_PENDING_EVENTS = 8192;
int resolve_events(np_f16_ctx *ctx) {
    size_t k = 0;
    for (; k < ctx->events.size(); k++) {
        auto &e = ctx->events[k];
        hipError_t err = hipEventSynchronize(e.second);
        float ms = 0.0f;
        if (err == hipSuccess) err = hipEventElapsedTime(&ms, e.first, e.second);
        if (err != hipSuccess) {
            ctx->events.erase(ctx->events.begin(), ctx->events.begin() + (long)k);
            return fail(std::string("timing events: ") + hipGetErrorString(err));
        }
        ctx->t_sum_ms += ms;
        ctx->t_count += 1;
        if (ctx->samples.size() < MAX_TIMING_SAMPLES) ctx->samples.push_back(ms);
        ctx->pool.push_back(e);
    }
    ctx->events.clear();
    return 0;
}

// A start / stop event pair taken from the context's pool for one timed launch; goes back to the pool unless the launch was
// enqueued (commit() hands it to ctx->events, where np_f16_get_timing resolves it)
struct EventLease {
    np_f16_ctx *ctx;
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    explicit EventLease(np_f16_ctx *c) : ctx(c) {}
    hipError_t take() {
        if (!ctx->pool.empty()) {
            ev = ctx->pool.back();
            ctx->pool.pop_back();
            return hipSuccess;
        }
        hipError_t e = hipEventCreate(&ev.first);
        if (e != hipSuccess) return e;
        e = hipEventCreate(&ev.second);
        if (e != hipSuccess) {
            (void)hipEventDestroy(ev.first);
            ev = {nullptr, nullptr};
        }
        return e;
    }
    void commit() {
        ctx->events.push_back(ev);
        ev = {nullptr, nullptr};
    }
    ~EventLease() {
        if (ev.first) ctx->pool.push_back(ev);
    }
};

// a launch whose stream is being captured into a HIP graph (PlanningEnv.enable_graph, a caller's torch.cuda.graph)
bool stream_is_capturing(hipStream_t st) {
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &status) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return status != hipStreamCaptureStatusNone;
}

constexpr int LAT_TILE = npdispatch::LAT_TILE;
static_assert(NP_KERNEL_AUTO == npdispatch::K_AUTO && NP_KERNEL_LATENCY == npdispatch::K_LATENCY && NP_KERNEL_THROUGHPUT == npdispatch::K_THROUGHPUT &&
                  NP_KERNEL_PAIR == npdispatch::K_PAIR && NP_KERNEL_LATENCY8 == npdispatch::K_LATENCY8 && NP_KERNEL_LATENCY2 == npdispatch::K_LATENCY2 &&
                  NP_KERNEL_LATENCY4W == npdispatch::K_LATENCY4W,
              "np_dispatch.h mirrors the NP_KERNEL_* numbers");
// Which variant a launch gets is a pure function of (n, CU count of the context's device, options): np_dispatch.h, where every threshold is
// written per CU.  What they mean on the 256-CU device they were measured on (profiles/r03_n_sweep.json, r03d_mid_n_lat4w.json,
// tools/microbench/ab_pair.sh): latency family up to 98 304 aircraft, the pair variant above (N = 1e5 0.060 / 0.062 ms against the
// throughput variant, 2e5 0.112 / 0.170, 1e6 0.384 / 0.402, 1e7 3.31 / 3.37); eight waves per tile up to 16 384 (a CU per tile); four waves
// per tile fill one generation of 768 tiles at three waves per SIMD up to 49 152 (30.6 us; 65 536 needs a second round: 45.6 us), built
// for FOUR waves per SIMD (128 VGPRs: it needs 132, a handful of dwords in scratch) 1 024 tiles up to 65 536 (34.6 us against 38.1 with
// two waves per tile); two waves per tile keep 1 024 - 1 536 tiles in ONE generation up to 98 304.  SingleCombat: latency variant up to
// 40 000 aircraft (16 384 engagements 0.116 vs 0.127 ms, 20 000 a tie, 24 576 0.213 vs 0.198).
int env_kernel_override() {  // process-wide override for experiments: NPF16_KERNEL=latency|throughput|pair
    static const int forced = [] {
        const char *e = std::getenv("NPF16_KERNEL");
        if (!e) return (int)NP_KERNEL_AUTO;
        return std::strcmp(e, "latency") == 0 ? (int)NP_KERNEL_LATENCY : std::strcmp(e, "throughput") == 0 ? (int)NP_KERNEL_THROUGHPUT
               : std::strcmp(e, "pair") == 0 ? (int)NP_KERNEL_PAIR : (int)NP_KERNEL_AUTO;
    }();
    return forced;
}
bool combat_dual_enabled() {
    static const bool on = [] { const char *e = std::getenv("NPF16_COMBAT_DUAL"); return !e || atoi(e) != 0; }();
    return on;
}

int pair_waves_override() {
    static const int pw_env = [] { const char *e = std::getenv("NPF16_PAIR_WAVES"); return e ? atoi(e) : 0; }();
    return pw_env;
}

// One kernel launch; when the context is being timed, the start / stop events are attached to the dispatch itself
// (hipExtLaunchKernelGGL: the timestamps of the kernel's own packet — no extra barrier packets between consecutive launches,
// and the duration is the kernel's execution time, as rocprofv3 reports it)
#define NP_DISPATCH(ARGS, ...)                                                                                       \
    do {                                                                                                             \
        if (timed) hipExtLaunchKernelGGL((__VA_ARGS__), grid, block, lds_pad, st, ev.first, ev.second, 0, ARGS);      \
        else hipLaunchKernelGGL((__VA_ARGS__), grid, block, lds_pad, st, ARGS);                                       \
    } while (0)

template <bool STEP>
int launch_env(np_f16_ctx *ctx, int64_t n, const np_f16_io *io, void *stream) {
    if (!ctx || !io) return fail("null ctx/io");
    if (ctx->combat) return fail("combat context: use np_f16_combat_reset / np_f16_combat_step");
    if (n <= 0) return 0;
    if (!io->s || !io->u || !io->tgt || !io->step_count || !io->done_in || !io->bad_in || !io->timeout_in ||
        !io->done_out || !io->bad_out || !io->timeout_out)
        return fail("null state/flag buffer");
    if (io->ld < n) return fail("ld < n");
    if (io->ld >= (1ll << 30)) return fail("ld must be below 2^30 rows (32-bit byte offsets inside a row-indexed array)");
    if (!STEP && io->inner_step) return fail("inner_step applies to np_f16_step only");
    const bool update_only = STEP && io->inner_step == NP_INNER_UPDATE_ONLY;   // F16Model.update(action) alone
    if (STEP && io->inner_step != 0 && io->inner_step != 1 && !update_only) return fail("inner_step: 0, 1 or NP_INNER_UPDATE_ONLY");
    if (STEP && !update_only && (!io->action || !io->reward || io->act_stride < 4 || (!io->obs && !(io->inner_step && io->ll_obs))))
        return fail("step needs action (>=4 columns), obs and reward buffers (obs may be NULL only for an inner step that writes ll_obs)");
    if (update_only && (!io->action || io->act_stride < 4)) return fail("update needs action (>=4 columns)");
    if (io->done_out == io->done_in || io->bad_out == io->bad_in || io->timeout_out == io->timeout_in)
        return fail("flag outputs may not alias flag inputs");
    DeviceGuard guard;
    NP_HIP(guard.enter(ctx->device));
    KArgs a;
    a.s = io->s; a.u = io->u; a.tgt = io->tgt; a.ld = io->ld; a.step_count = (long long *)io->step_count;
    a.fin0 = io->done_in; a.fin1 = io->bad_in; a.fin2 = io->timeout_in;
    a.fout0 = io->done_out; a.fout1 = io->bad_out; a.fout2 = io->timeout_out;
    a.action = io->action; a.act_stride = io->act_stride; a.obs = io->obs; a.reward = io->reward;
    a.inner = update_only ? 2 : io->inner_step ? 1 : 0;
    a.rand_u = io->rand_u; a.noise = io->noise; a.cache = io->coef_cache; a.seed = io->seed; a.call_idx = io->call_idx;
    a.call_idx_base = io->call_idx_base;
    a.term_counters = update_only ? nullptr : io->term_counters;   // update alone evaluates no condition for anybody to see
    a.term_reasons = update_only ? nullptr : io->term_reasons;
    a.reward_task = (STEP && !update_only) ? io->reward_task : nullptr;
    a.ll_tgt = (STEP && io->inner_step && !update_only) ? io->ll_tgt : nullptr;
    a.ll_obs = (STEP && io->inner_step && !update_only) ? io->ll_obs : nullptr;
    if (a.ll_obs && !a.ll_tgt) return fail("ll_obs needs ll_tgt");
    a.row0 = io->row0; a.n = n; a.cfg = ctx->cfg;
    a.reset_coef = ctx->d_reset_coef;
    a.wt = ctx->wt;
    a.trace = nullptr;
    if (STEP && ctx->trace) {
        const int64_t wgs = (n + 63) / 64;  // upper bound over the variants' grids
        if (wgs > ctx->trace_cap) return fail("np_f16_set_trace: buffer too small for this launch");
        a.trace = ctx->trace;
    }
    // small batches: four waves per 64-aircraft tile (latency variant); NPF16_KERNEL=throughput|latency overrides
    // pair: both numerics (round 3: the table mode's multi-input nets on the two-set bodies).  latency8: eight waves per tile, small
    // batches of the MLP numerics (the table mode's classes are too short to split further).  latency2: two waves per tile, above the
    // four-wave variant's single generation (both numerics; Euler).  latency4w: four waves per tile at four waves per SIMD, one generation
    // of four tiles per CU (MLP numerics; the table mode's kernels are not built for it).
    const npdispatch::EnvChoice ch = npdispatch::env_choice(n, ctx->num_cus, STEP, ctx->solver, ctx->cfg.aero_1d_tables != 0, ctx->variant,
                                                            env_kernel_override(), pair_waves_override(), BLOCK);
    const bool pair = ch.pair, latency = ch.latency, latency8 = ch.latency8, latency2 = ch.latency2, latency4w = ch.latency4w;
    const dim3 grid((unsigned)ch.grid), block((unsigned)ch.block);
    a.cus = ctx->num_cus;
    hipStream_t st = (hipStream_t)stream;
    // Pair variant at three waves per SIMD (six workgroups per CU) or at two (four per CU)?  Measured per grid size (heading,
    // one session, profiles/r02b_ab_sessions.md s25): long grids gain 5-9 % from the third wave; grids of up to three generations
    // are quantised — 1 025-1 536 workgroups fit ONE generation of six per CU (-11..-15 %), but <= 1 024 fill the chip evenly at
    // four per CU (a kernel that MAY hold three waves per SIMD is placed unevenly there: +13..+20 %).  1 537-3 071 workgroups were
    // better off in rounds of 1 024 only while the three-wave build ran them in lock-step: with the first-generation delay applied
    // to every three-wave grid (round 3, see the kernel) it wins there too (229 376 aircraft 124.2 -> 88.9 us, 327 680 126.4 ->
    // 118.9; profiles/r03f_mid_large_n.log).  PlanningEnv's inner steps take the same rule since the end of round 3 (165 VGPRs, no
    // scratch; the macro-step at n = 150 000: 29.2 -> 27.6 ms, 262 144: 45.8 -> 44.2 ms, profiles/r03f_planning_inner_pair3.log).
    const bool pair3 = ch.pair3;
    if (io->cache_valid && !io->coef_cache) return fail("cache_valid set without a coef_cache buffer");
    const bool cached = STEP && io->coef_cache && io->cache_valid;
    const bool timed = STEP && ctx->timing && !stream_is_capturing(st);  // event-attached dispatches cannot be captured into a graph
    EventLease lease(ctx);
    if (timed && ctx->events.size() >= MAX_PENDING_EVENTS && resolve_events(ctx)) return 1;  // a caller that never polls
    if (timed) NP_HIP(lease.take());
    const std::pair<hipEvent_t, hipEvent_t> &ev = lease.ev;
    // the kernels live in one translation unit per task x solver (np_env_t*s*.hip, np_env_launch.h); a reset has no solver: Euler unit
    EnvLaunch l;
    l.a = a; l.grid = grid.x; l.block = block.x; l.st = st; l.timed = timed; l.start = ev.first; l.stop = ev.second;
    l.step = STEP; l.inner = STEP && a.inner; l.cached = cached;
    l.pair = pair; l.pair3 = pair3; l.latency = latency; l.latency8 = latency8; l.latency2 = latency2; l.latency4w = latency4w;
    const int key = ctx->task * 2 + (STEP ? ctx->solver : 0);
    switch (key) {
    case 0: env_dispatch_t0s0(l); break;
    case 1: env_dispatch_t0s1(l); break;
    case 2: env_dispatch_t1s0(l); break;
    case 3: env_dispatch_t1s1(l); break;
    case 4: env_dispatch_t2s0(l); break;
    case 5: env_dispatch_t2s1(l); break;
    default: return fail("bad task/solver");
    }
    NP_HIP(hipGetLastError());
    if (timed) lease.commit();
    return 0;
}


PidDev make_pid(const np_pid_gains &g, double dt) {
    PidDev p;
    p.Kp = (float)g.Kp;
    p.Ki = (float)g.Ki;
    p.Kd = (float)g.Kd;
    p.Kff = (float)g.Kff;
    p.Kimax = (float)g.Kimax;
    p.tau = (float)(g.tau < 0.05 ? 0.05 : g.tau);  // rollController.py:44-45
    p.rmax_pos = (float)g.rmax_pos;
    p.rmax_neg = (float)g.rmax_neg;
    p.ki_on = (g.Ki != 0.0 && dt > 0.0) ? 1 : 0;
    return p;
}

CombatDevCfg make_combat_devcfg(const np_f16_combat_cfg &c) {
    CombatDevCfg d;
    d.af = make_airframe(c.airframe);
    d.dt = (float)c.dt - 0.0f;
    d.dt_pid = (float)c.dt;
    d.airspeed = (float)c.airspeed;
    d.altitude_limit = (float)c.altitude_limit;
    d.acceleration_limit = (float)c.acceleration_limit;
    d.max_velocity = (float)c.max_velocity;
    d.min_velocity = (float)c.min_velocity;
    d.min_alpha = (float)c.min_alpha;
    d.max_alpha = (float)c.max_alpha;
    d.min_beta = (float)c.min_beta;
    d.max_beta = (float)c.max_beta;
    d.dist_limit_sq = (float)(c.distance_limit * c.distance_limit);  // `self.distance_limit ** 2` is Python arithmetic
    d.max_steps = c.max_steps;
    d.init_T = (float)c.init_T;
    d.alt_span = (float)(c.max_altitude - c.min_altitude);
    d.min_altitude = (float)c.min_altitude;
    d.vt_span = (float)(c.max_vt - c.min_vt);
    d.min_vt = (float)c.min_vt;
    d.yaw_span = (float)(c.max_heading - c.min_heading);
    d.min_heading = (float)c.min_heading;
    d.npos_span = (float)(c.max_npos - c.min_npos);
    d.min_npos = (float)c.min_npos;
    d.epos_span = (float)(c.max_epos - c.min_epos);
    d.min_epos = (float)c.min_epos;
    d.roll = make_pid(c.roll, c.dt);
    d.pitch = make_pid(c.pitch, c.dt);
    d.yaw = make_pid(c.yaw, c.dt);
    d.roll_ff = (float)c.roll_ff;
    d.gravity = (float)c.gravity;
    d.scale_min = (float)std::min(0.5, 1000.0 / (2.0 * c.airspeed_max));   // controller.py:36-37
    d.scale_max = (float)std::max(2.0, 1000.0 / (0.7 * c.airspeed_min));
    d.inner_steps = c.inner_steps;
    d.aero_1d_tables = c.aero_1d_tables ? 1 : 0;
    return d;
}

template <bool STEP>
int launch_combat(np_f16_ctx *ctx, int64_t num_envs, const np_f16_combat_io *io, void *stream) {
    if (!ctx || !io) return fail("null ctx/io");
    if (!ctx->combat) return fail("context was not created by np_f16_combat_ctx_create");
    if (num_envs <= 0) return 0;
    const int64_t n = 2 * num_envs;
    if (!io->s || !io->u || !io->blood || !io->step_count || !io->done_in || !io->bad_in || !io->timeout_in || !io->done_out ||
        !io->bad_out || !io->timeout_out)
        return fail("null state/flag buffer");
    if (io->ld < n) return fail("ld < n");
    if (STEP && (!io->pid || !io->action || !io->obs || !io->reward || io->act_stride < 4))
        return fail("step needs pid, action (>=4 columns), obs and reward buffers");
    if (io->done_out == io->done_in || io->bad_out == io->bad_in || io->timeout_out == io->timeout_in)
        return fail("flag outputs may not alias flag inputs");
    if (io->row0 & 1) return fail("row0 must be even (2 * first env of the shard)");
    DeviceGuard guard;
    NP_HIP(guard.enter(ctx->device));
    CombatArgs a;
    a.s = io->s; a.u = io->u; a.pid = io->pid; a.blood = io->blood; a.ld = io->ld; a.step_count = (long long *)io->step_count;
    a.fin0 = io->done_in; a.fin1 = io->bad_in; a.fin2 = io->timeout_in;
    a.fout0 = io->done_out; a.fout1 = io->bad_out; a.fout2 = io->timeout_out;
    a.action = io->action; a.act_stride = io->act_stride; a.obs = io->obs; a.reward = io->reward; a.rand_u = io->rand_u;
    a.pid_first = io->pid_first; a.seed = io->seed; a.call_idx = io->call_idx; a.row0 = io->row0; a.n = n; a.cfg = ctx->ccfg;
    a.term_counters = io->term_counters;
    a.wt = ctx->wt;
    a.action_opp = STEP ? io->action_opp : nullptr;
    a.obs_opp = io->obs_opp;
    if (io->obs_opp && !io->obs) return fail("obs_opp needs obs (the ego half)");
    // small batches: the latency variant (one generation of 4-wave workgroups at 2 waves per SIMD = 512 x 64 aircraft)
    const npdispatch::Limits lim = npdispatch::limits_for(ctx->num_cus);
    // more than one 64-aircraft tile per CU: the dual family (np_combat_lat.hip) — tiles of 128 aircraft, eight waves per tile up to one tile per
    // CU, four up to two; NP_KERNEL_LATENCY pins the four-waves-per-64-aircraft kernel, NPF16_COMBAT_DUAL=0 switches the family off (A/B)
    const bool dual_ok = STEP && ctx->solver == 0 && !ctx->ccfg.aero_1d_tables;
    const bool dual_pin = ctx->variant == NP_KERNEL_DUAL8 || ctx->variant == NP_KERNEL_DUAL4;
    const int dual = !dual_ok ? 0 : dual_pin ? (ctx->variant == NP_KERNEL_DUAL8 ? 8 : 4)
                     : (ctx->variant == NP_KERNEL_AUTO && combat_dual_enabled()) ? npdispatch::combat_dual_waves(n, ctx->num_cus) : 0;
    const bool auto_rule = ctx->variant == NP_KERNEL_AUTO || dual_pin;  // a dual pin that does not apply (reset, rk4, table numerics) leaves the size rule
    const bool latency = !dual && STEP && ctx->solver == 0 &&
                         (ctx->variant == NP_KERNEL_LATENCY || (auto_rule && n <= lim.combat_lat_max_n));
    const dim3 grid((unsigned)(dual ? (n + COMBAT_DUAL_TILE - 1) / COMBAT_DUAL_TILE : latency ? (n + LAT_TILE - 1) / LAT_TILE : (n + COMBAT_BLOCK - 1) / COMBAT_BLOCK)),
        block(latency ? LAT_TILE * 4 : COMBAT_BLOCK);
    hipStream_t st = (hipStream_t)stream;
    const bool timed = STEP && ctx->timing && !stream_is_capturing(st);
    const unsigned lds_pad = 0;
    EventLease lease(ctx);
    if (timed && ctx->events.size() >= MAX_PENDING_EVENTS && resolve_events(ctx)) return 1;  // a caller that never polls
    if (timed) NP_HIP(lease.take());
    const std::pair<hipEvent_t, hipEvent_t> &ev = lease.ev;
    // pair variant (Euler, MLP numerics): the default above the latency variant's range; NP_KERNEL_THROUGHPUT pins the single-set kernel
    const bool pair = STEP && !latency && ctx->solver == 0 && !ctx->ccfg.aero_1d_tables && ctx->variant != NP_KERNEL_THROUGHPUT;
    // three waves per SIMD (168 VGPRs, 4 dwords per lane in scratch since the state is pinned before the Overload phase — round 2: 46)
    // from 1 025 workgroups on: 70 000 engagements 0.275 vs 0.290 ms, 100 000 0.355 vs 0.379, 500 000 1.294 vs 1.412; a tie below
    // (profiles/r03b_combat_pair_waves.log)
    const int pw_env = pair_waves_override();
    const bool pair3 = pair && (pw_env ? pw_env == 3 : (int64_t)grid.x > lim.pair3_min_grid);
    if (dual) launch_combat_dual(a, dual, grid.x, st, timed, ev.first, ev.second);
    else if (latency) NP_DISPATCH(a, f16_combat_kernel<0, STEP, LAT_TILE, 4>);
    else if (pair3) NP_DISPATCH(a, f16_combat_kernel<0, STEP, COMBAT_BLOCK, 2, 3>);
    else if (pair) NP_DISPATCH(a, f16_combat_kernel<0, STEP, COMBAT_BLOCK, 2>);
    else if (STEP && ctx->solver == 1) NP_DISPATCH(a, f16_combat_kernel<1, STEP>);
    else NP_DISPATCH(a, f16_combat_kernel<0, STEP>);
    NP_HIP(hipGetLastError());
    if (timed) lease.commit();
    return 0;
}

}  // namespace

extern "C" {

int np_abi_version(void) { return NP_ABI_VERSION; }

void np_f16_airframe_default(np_f16_airframe *out) {
    if (out) *out = airframe_defaults();
}

int np_dispatch_plan(int64_t n, int32_t num_cus, int32_t step, int32_t solver, int32_t tables, int32_t variant, np_dispatch_info *out) {
    if (!out) return fail("null argument");
    if (n <= 0 || num_cus <= 0) return fail("np_dispatch_plan: n and num_cus must be positive");
    if (variant < NP_KERNEL_AUTO || variant > NP_KERNEL_DUAL4) return fail("np_dispatch_plan: unknown variant");
    const bool dual_pin = variant == NP_KERNEL_DUAL8 || variant == NP_KERNEL_DUAL4;   // SingleCombat only: the env kernels keep their size rule
    const npdispatch::EnvChoice c = npdispatch::env_choice(n, num_cus, step != 0, solver, tables != 0, dual_pin ? NP_KERNEL_AUTO : variant, NP_KERNEL_AUTO, 0, BLOCK);
    const npdispatch::Limits l = npdispatch::limits_for(num_cus);
    out->pair = c.pair; out->pair3 = c.pair3; out->latency = c.latency; out->latency8 = c.latency8; out->latency2 = c.latency2;
    out->latency4w = c.latency4w; out->block = c.block; out->grid = c.grid;
    out->planning_groups = npdispatch::planning_groups(n, num_cus);
    out->actor_tile32 = npdispatch::actor_tile32(n, num_cus) ? 1 : 0;
    const int dual = !(step && solver == 0 && !tables) ? 0 : dual_pin ? (variant == NP_KERNEL_DUAL8 ? 8 : 4)
                     : variant == NP_KERNEL_AUTO ? npdispatch::combat_dual_waves(n, num_cus) : 0;
    out->combat_latency = dual == 8 ? 2 : dual == 4 ? 3 : n <= l.combat_lat_max_n ? 1 : 0;
    out->planning_mode = npdispatch::planning_mode(n, num_cus);   // one eight-wave workgroup per CU
    out->planning_mode_i8 = npdispatch::planning_mode(n, num_cus, true);
    return 0;
}
int64_t np_f16_cache_floats(int64_t n) { return n <= 0 ? 0 : ((n + BLOCK - 1) / BLOCK) * (int64_t)BLOCK * NUM_CACHE_ROWS; }
const char *np_last_error(void) { return g_err.c_str(); }
}  // extern "C"
int np_internal_fail(const char *msg) { return fail(msg ? msg : "unknown error"); }
extern "C" {

static int ctx_create_common(const void *weights_blob, size_t nbytes, int tables, int device, np_f16_ctx **out) {
    std::vector<float> kb, kbd, pwl, pwl_unnorm;
    if (pack_kblob(weights_blob, nbytes, kb, kbd, pwl, pwl_unnorm)) return 1;
    if (tables && pwl.empty()) return fail("cfg.aero_1d_tables needs a version-2 weights blob (PWL section)");
    int ndev = 0;
    NP_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail("no such HIP device (this library has no CPU fallback)");
    DeviceGuard guard;
    NP_HIP(guard.enter(device));
    hipDeviceProp_t prop;
    NP_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(std::string("device arch ") + prop.gcnArchName + " is not gfx950 (MI355X)");
    // one device allocation per context: KBLOB | PWL tables | PWL (std, mean) | KBLOB in the two-set layout — a context is one aircraft type
    const size_t n_kb = KBLOB_FLOATS, n_pwl = (size_t)NUM_PWL_TABLES * PWL_TABLE_FLOATS, n_un = (size_t)NUM_PWL_TABLES * 2;
    const size_t off_dual = (n_kb + n_pwl + n_un + 63) / 64 * 64;  // 256-byte aligned
    float *d_w = nullptr, *d_rc = nullptr;
    NP_HIP(hipMalloc(&d_w, sizeof(float) * (off_dual + KBLOB_DUAL_FLOATS)));
    hipError_t e0 = hipMemcpy(d_w, kb.data(), sizeof(float) * n_kb, hipMemcpyHostToDevice);
    if (e0 == hipSuccess) e0 = hipMemcpy(d_w + off_dual, kbd.data(), sizeof(float) * KBLOB_DUAL_FLOATS, hipMemcpyHostToDevice);
    if (e0 == hipSuccess && !pwl.empty()) {
        e0 = hipMemcpy(d_w + n_kb, pwl.data(), sizeof(float) * n_pwl, hipMemcpyHostToDevice);
        if (e0 == hipSuccess) e0 = hipMemcpy(d_w + n_kb + n_pwl, pwl_unnorm.data(), sizeof(float) * n_un, hipMemcpyHostToDevice);
    }
    AeroWeights wt;
    wt.kblob = d_w;
    wt.kblob_dual = d_w + off_dual;
    wt.pwl = pwl.empty() ? nullptr : d_w + n_kb;
    wt.pwl_unnorm = pwl.empty() ? nullptr : d_w + n_kb + n_pwl;
    hipError_t e1 = e0 == hipSuccess ? hipMalloc(&d_rc, sizeof(float) * NUM_CACHED) : e0;
    hipError_t e2 = hipSuccess, e3 = hipSuccess;
    if (e1 == hipSuccess) {  // coefficients of a reset aircraft, evaluated by the device code itself (bit-identical to in-line evaluation)
        hipLaunchKernelGGL(f16_reset_coef_kernel, dim3(1), dim3(BLOCK), 0, 0, d_rc, tables ? 1 : 0, wt);
        e2 = hipGetLastError();
        e3 = hipDeviceSynchronize();
    }
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
        (void)hipFree(d_w);
        if (d_rc) (void)hipFree(d_rc);
        NP_HIP(e1);
        NP_HIP(e2);
        NP_HIP(e3);
    }
    np_f16_ctx *ctx = new np_f16_ctx();
    ctx->device = device;
    ctx->task = 0;
    ctx->solver = 0;
    ctx->combat = false;
    ctx->variant = NP_KERNEL_AUTO;
    ctx->d_reset_coef = d_rc;
    ctx->d_weights = d_w;
    ctx->wt = wt;
    ctx->timing = false;
    ctx->t_sum_ms = 0.0;
    ctx->t_count = 0;
    ctx->trace = nullptr;
    ctx->trace_cap = 0;
    ctx->d_queue = nullptr;
    ctx->queue_cap = 0;
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    *out = ctx;
    return 0;
}

int np_f16_ctx_create(const void *weights_blob, size_t nbytes, const np_f16_cfg *cfg, int device, np_f16_ctx **out) {
    if (!out || !cfg) return fail("null argument");
    *out = nullptr;
    if (cfg->task < 0 || cfg->task > 2) return fail("cfg.task must be NP_TASK_HEADING/CONTROL/TRACKING");
    if (cfg->solver < 0 || cfg->solver > 1) return fail("cfg.solver must be NP_SOLVER_EULER/RK4");
    if (!airframe_is_zero(cfg->airframe))
        if (const char *e = airframe_error(cfg->airframe)) return fail(e);
    if (ctx_create_common(weights_blob, nbytes, cfg->aero_1d_tables, device, out)) return 1;
    (*out)->task = cfg->task;
    (*out)->solver = cfg->solver;
    (*out)->cfg = make_devcfg(*cfg);
    return 0;
}

int np_f16_combat_ctx_create(const void *weights_blob, size_t nbytes, const np_f16_combat_cfg *cfg, int device,
                             np_f16_ctx **out) {
    if (!out || !cfg) return fail("null argument");
    *out = nullptr;
    if (cfg->solver < 0 || cfg->solver > 1) return fail("cfg.solver must be NP_SOLVER_EULER/RK4");
    if (cfg->inner_steps < 1 || cfg->inner_steps > 16) return fail("cfg.inner_steps must be in 1..16");
    if (!(cfg->dt > 0.0)) return fail("cfg.dt must be positive");
    if (!airframe_is_zero(cfg->airframe))
        if (const char *e = airframe_error(cfg->airframe)) return fail(e);
    if (ctx_create_common(weights_blob, nbytes, cfg->aero_1d_tables, device, out)) return 1;
    (*out)->solver = cfg->solver;
    (*out)->combat = true;
    (*out)->ccfg = make_combat_devcfg(*cfg);
    (*out)->cfg = DevCfg();
    (*out)->cfg.airspeed = (float)cfg->airspeed;
    (*out)->cfg.aero_1d_tables = cfg->aero_1d_tables ? 1 : 0;
    return 0;
}

int np_f16_combat_reset(np_f16_ctx *ctx, int64_t num_envs, const np_f16_combat_io *io, void *stream) {
    return launch_combat<false>(ctx, num_envs, io, stream);
}

int np_f16_combat_step(np_f16_ctx *ctx, int64_t num_envs, const np_f16_combat_io *io, void *stream) {
    return launch_combat<true>(ctx, num_envs, io, stream);
}

void np_f16_ctx_destroy(np_f16_ctx *ctx) {
    if (!ctx) return;
    {
        DeviceGuard guard;
        if (guard.enter(ctx->device) == hipSuccess) {
            if (ctx->d_reset_coef) (void)hipFree(ctx->d_reset_coef);  // hipFree waits for kernels still reading the buffers
            if (ctx->d_weights) (void)hipFree(ctx->d_weights);
        }
    }
    for (auto &e : ctx->events) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    for (auto &e : ctx->pool) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    if (ctx->d_queue) (void)hipFree(ctx->d_queue);
    if (ctx->d_plan_err) (void)hipFree(ctx->d_plan_err);
    if (ctx->h_plan_err) (void)hipHostFree(ctx->h_plan_err);
    if (ctx->d_snap) (void)hipFree(ctx->d_snap);
    if (ctx->plan_done) (void)hipEventDestroy(ctx->plan_done);
    if (!ctx->group_streams.empty() || !ctx->group_events.empty()) {
        DeviceGuard guard;
        if (guard.enter(ctx->device) == hipSuccess) {
            for (auto st : ctx->group_streams) {
                (void)hipStreamSynchronize(st);
                (void)hipStreamDestroy(st);
            }
            for (auto ev : ctx->group_events) (void)hipEventDestroy(ev);
        }
    }
    delete ctx;
}

int np_f16_reset(np_f16_ctx *ctx, int64_t n, const np_f16_io *io, void *stream) { return launch_env<false>(ctx, n, io, stream); }

int np_f16_step(np_f16_ctx *ctx, int64_t n, const np_f16_io *io, void *stream) { return launch_env<true>(ctx, n, io, stream); }

int np_f16_derived(np_f16_ctx *ctx, int64_t n, const float *s, const float *u, int64_t ld, float *out, int64_t ld_out,
                   void *stream) {
    if (!ctx || !s || !u || !out) return fail("null argument");
    if (n <= 0) return 0;
    if (ld < n || ld_out < n) return fail("leading dimension < n");
    DeviceGuard guard;
    NP_HIP(guard.enter(ctx->device));
    const dim3 grid((unsigned)((n + BLOCK - 1) / BLOCK)), block(BLOCK);
    hipLaunchKernelGGL(f16_derived_kernel, grid, block, 0, (hipStream_t)stream, s, u, (long long)ld, out, (long long)ld_out,
                       (long long)n, ctx->cfg.airspeed, ctx->cfg.aero_1d_tables, ctx->wt, ctx->combat ? ctx->ccfg.af : ctx->cfg.af);
    NP_HIP(hipGetLastError());
    return 0;
}

int np_f16_aero_coefficients(np_f16_ctx *ctx, int64_t n, const float *alpha_deg, const float *beta_deg, const float *el, float *out,
                             int64_t ld_out, void *stream) {
    if (!ctx || !alpha_deg || !beta_deg || !el || !out) return fail("null argument");
    if (n <= 0) return 0;
    if (ld_out < n) return fail("leading dimension < n");
    DeviceGuard guard;
    NP_HIP(guard.enter(ctx->device));
    const dim3 grid((unsigned)((n + BLOCK - 1) / BLOCK)), block(BLOCK);
    hipLaunchKernelGGL(f16_aero_kernel, grid, block, 0, (hipStream_t)stream, alpha_deg, beta_deg, el, (long long)n, out,
                       (long long)ld_out, ctx->cfg.aero_1d_tables, ctx->wt);
    NP_HIP(hipGetLastError());
    return 0;
}

int np_f16_lowlevel_obs(np_f16_ctx *ctx, int64_t n, const float *s, const float *u, const float *tgt3, int64_t ld, float *obs,
                        void *stream) {
    if (!ctx || !s || !u || !tgt3 || !obs) return fail("null argument");
    if (n <= 0) return 0;
    if (ld < n) return fail("leading dimension < n");
    DeviceGuard guard;
    NP_HIP(guard.enter(ctx->device));
    const dim3 grid((unsigned)((n + BLOCK - 1) / BLOCK)), block(BLOCK);
    hipLaunchKernelGGL(f16_lowlevel_obs_kernel<false>, grid, block, 0, (hipStream_t)stream, s, u, tgt3, (long long)ld, obs, (long long)n,
                       ctx->cfg, 0ll, (float *)nullptr);
    NP_HIP(hipGetLastError());
    return 0;
}

int np_planning_targets_obs(np_f16_ctx *ctx, int64_t n, const float *s, const float *u, const float *action, int64_t act_stride, float *tgt3,
                            int64_t ld, float *obs, void *stream) {
    if (!ctx || !s || !u || !action || !tgt3 || !obs) return fail("null argument");
    if (n <= 0) return 0;
    if (ld < n) return fail("leading dimension < n");
    if (act_stride < 3) return fail("act_stride < 3");
    DeviceGuard guard;
    NP_HIP(guard.enter(ctx->device));
    const dim3 grid((unsigned)((n + BLOCK - 1) / BLOCK)), block(BLOCK);
    hipLaunchKernelGGL(f16_lowlevel_obs_kernel<true>, grid, block, 0, (hipStream_t)stream, s, u, action, (long long)ld, obs, (long long)n,
                       ctx->cfg, (long long)act_stride, tgt3);
    NP_HIP(hipGetLastError());
    return 0;
}

int np_actor_forward(const float *weights, int64_t num_floats, int64_t n, const float *obs, const float *h_in, const float *masks,
                     float *actions, float *h_out, int device, void *stream) {
    if (!weights || !obs || !h_in || !masks || !actions || !h_out) return fail("null argument");
    if (num_floats != npact::TOTAL && num_floats != NP_ACTOR_I8_NUM_FLOATS)
        return fail("packed actor weights: wrong size (expected NP_ACTOR_NUM_FLOATS, or NP_ACTOR_I8_NUM_FLOATS for the block-fixed-point numerics)");
    if (((uintptr_t)h_in | (uintptr_t)h_out) & 15) return fail("h_in / h_out must be 16-byte aligned");
    if (n <= 0) return 0;
    int ndev = 0;
    NP_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail("no such HIP device (this library has no CPU fallback)");
    DeviceGuard guard;
    NP_HIP(guard.enter(device));
    if (num_floats == NP_ACTOR_I8_NUM_FLOATS) {   // the second numerics spec: one kernel, 32-aircraft tiles (np_actor_i8.hip)
        if ((uintptr_t)weights & 15) return fail("packed actor weights must be 16-byte aligned");
        NP_HIP(npact8::launch_actor_i8(weights, (long long)n, obs, h_in, masks, actions, h_out, (hipStream_t)stream));
        return 0;
    }
#if NPACT_MFMA
    // small batches: 32-row tiles (twice the workgroups, half the MFMA chain per tile); NP_ACTOR_TILE=32|64 overrides (benchmarks)
    const char *tile_str = std::getenv("NP_ACTOR_TILE");  // read per call: the tests switch it
    const int tile_env = tile_str ? atoi(tile_str) : 0;
    // which tiling: a round of the 32-row kernel is 512 tiles = 16 384 rows in ~60 us, of the 64-row kernel 32 768 rows in ~98 us; the
    // 32-row kernel wins wherever it needs fewer or shorter rounds (tools/microbench/actor_bench.py, profiles/r03h_actor_tilings.log:
    // n = 20 000 84.6 vs 98.7 us, 24 576 86.9 vs 98.3, 40 960 138.8 vs 149.2; 28 672 110.3 vs 98.5 and everything from 49 152 on the other way)
    // (the bounds are per CU in np_dispatch.h)
    static int cus_of[64] = {};
    int cus = npdispatch::REF_CUS;
    if (device < 64) {
        if (cus_of[device] == 0) {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || v <= 0) v = npdispatch::REF_CUS;
            cus_of[device] = v;
        }
        cus = cus_of[device];
    }
    const bool tile32_auto = npdispatch::actor_tile32(n, cus);
    if (tile_env == 32 || (tile_env != 64 && tile32_auto)) {
        if ((uintptr_t)weights & 15) return fail("packed actor weights must be 16-byte aligned");
        const dim3 grid32((unsigned)((n + npact::T32 - 1) / npact::T32)), block32(npact::MTHREADS);
        hipLaunchKernelGGL(npact::actor_forward_mfma32_kernel, grid32, block32, 0, (hipStream_t)stream, weights, (long long)n, obs, h_in,
                           masks, actions, h_out);
        NP_HIP(hipGetLastError());
        return 0;
    }
    const auto kernel = npact::actor_forward_mfma_kernel;
    const dim3 grid((unsigned)((n + npact::TILE - 1) / npact::TILE)), block(npact::MTHREADS);
#else
    const auto kernel = npact::actor_forward_kernel;
    const dim3 grid((unsigned)((n + npact::TILE - 1) / npact::TILE)), block(npact::THREADS);
#endif
    static bool lds_set[64] = {};
    if (device < 64 && !lds_set[device]) {  // 68 KB of LDS per workgroup: above the 64 KB a kernel may use without asking
        NP_HIP(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)npact::ACTOR_LDS_BYTES));
        lds_set[device] = true;
    }
    hipLaunchKernelGGL(kernel, grid, block, npact::ACTOR_LDS_BYTES, (hipStream_t)stream, weights, (long long)n, obs, h_in, masks,
                       actions, h_out);
    NP_HIP(hipGetLastError());
    return 0;
}

#if NPACT_TRACE
extern "C" int np_actor_trace_read(long long *out) {  // diagnostics builds only (tools/microbench/actor_phases.py)
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(npact::npact_trace), sizeof(long long) * 64) == hipSuccess ? 0 : 1;
}
#endif

// ---- bounded waits of the guest / queue schedules: snapshot / restore of a macro-step's inputs, the stall report ---------------------------
namespace {
struct CopySeg { const void *src; void *dst; unsigned long long bytes; };
constexpr int MAX_COPY_SEGS = 10;
struct CopyArgs { CopySeg seg[MAX_COPY_SEGS]; };
// blockIdx.y = segment; 16-byte pieces where pointers and size allow, bytes otherwise (the flag / reason arrays: n bytes each)
__global__ void __launch_bounds__(256) plan_copy_kernel(CopyArgs a) {
    const CopySeg sg = a.seg[blockIdx.y];
    const unsigned long long stride = (unsigned long long)gridDim.x * 256ull, i0 = (unsigned long long)blockIdx.x * 256ull + threadIdx.x;
    if ((((unsigned long long)sg.src | (unsigned long long)sg.dst | sg.bytes) & 15ull) == 0ull) {
        for (unsigned long long i = i0; i < sg.bytes / 16; i += stride) reinterpret_cast<uint4 *>(sg.dst)[i] = reinterpret_cast<const uint4 *>(sg.src)[i];
    } else if ((((unsigned long long)sg.src | (unsigned long long)sg.dst | sg.bytes) & 3ull) == 0ull) {
        for (unsigned long long i = i0; i < sg.bytes / 4; i += stride) reinterpret_cast<unsigned *>(sg.dst)[i] = reinterpret_cast<const unsigned *>(sg.src)[i];
    } else {
        for (unsigned long long i = i0; i < sg.bytes; i += stride) reinterpret_cast<unsigned char *>(sg.dst)[i] = reinterpret_cast<const unsigned char *>(sg.src)[i];
    }
}

// everything the persistent kernel updates IN PLACE or accumulates into (np_planning.hip: plan_fdm_step's exports, the recurrent state and
// low-level observation ping-pong whose final values land in buffer 0 again for an even number of iterations, the termination counters);
// pure outputs (obs, reward, reward_task, ll_act, the second ping-pong buffers) are rewritten by a re-run and are not kept
struct PlanSnapshot {
    CopyArgs save, restore;
    int count = 0;
    size_t bytes = 0;
    void add(void *p, size_t nbytes) {
        if (!p || nbytes == 0 || count >= MAX_COPY_SEGS) return;
        save.seg[count].src = p; save.seg[count].bytes = nbytes;
        restore.seg[count].dst = p; restore.seg[count].bytes = nbytes;
        save.seg[count].dst = nullptr; restore.seg[count].src = nullptr;   // the context's buffer: resolved in bind()
        offs[count] = bytes;
        bytes += (nbytes + 15) & ~(size_t)15;
        count++;
    }
    void bind(char *base) {
        for (int i = 0; i < count; i++) {
            save.seg[i].dst = base + offs[i];
            restore.seg[i].src = base + offs[i];
        }
    }
    hipError_t run(bool back, hipStream_t st) const {
        if (count == 0) return hipSuccess;
        hipLaunchKernelGGL(plan_copy_kernel, dim3(128, (unsigned)count), dim3(256), 0, st, back ? restore : save);
        return hipGetLastError();
    }
    size_t offs[MAX_COPY_SEGS] = {};
};

int plan_err_buffers(np_f16_ctx *ctx) {
    if (ctx->d_plan_err) return 0;
    NP_HIP(hipMalloc((void **)&ctx->d_plan_err, sizeof(unsigned) * 4));
    NP_HIP(hipMemset(ctx->d_plan_err, 0, sizeof(unsigned) * 4));
    NP_HIP(hipHostMalloc((void **)&ctx->h_plan_err, sizeof(unsigned) * PLAN_ERR_WORDS, hipHostMallocMapped));
    std::memset(ctx->h_plan_err, 0, sizeof(unsigned) * PLAN_ERR_WORDS);
    NP_HIP(hipHostGetDevicePointer((void **)&ctx->h_plan_err_dev, ctx->h_plan_err, 0));
    NP_HIP(hipEventCreateWithFlags(&ctx->plan_done, hipEventDisableTiming));
    return 0;
}

// the record a stalled launch left (h_plan_err[0] != 0) as a message; clears it and marks the queue words for clearing
std::string plan_stall_message(np_f16_ctx *ctx, const char *mode, bool restored) {
    volatile unsigned *e = ctx->h_plan_err;
    std::string m = std::string("np_planning_inner_loop (") + mode + "): workgroup " + std::to_string(e[0] - 1u) + " waited " + std::to_string(e[4]) +
                    " ms for tile " + std::to_string(e[1]) + " to reach iteration " + std::to_string(e[2]) + " (published: " + std::to_string(e[3]) +
                    "): the schedule's workgroups did not make progress together (the guest / queue schedules assume a resident grid or in-order "
                    "workgroup dispatch).  The kernel drained and ended; ";
    m += restored ? "every buffer it updates in place (s, u, step_count, flags[0], rnn[0], ll_obs[0], coef_cache, term_reasons, term_counters) was "
                    "restored to its value before the call: re-run the macro-step with mode = NP_PLANNING_LAUNCHES"
                  : "check = deferred keeps no copy of the inputs: the state buffers hold a partially advanced macro-step";
    for (int i = 0; i < PLAN_ERR_WORDS; i++) e[i] = 0u;
    ctx->queue_dirty = true;
    return m;
}

// check = deferred: the verdict on the previous guest / queue launch of this context, before the next one may reuse the words
int plan_check_pending(np_f16_ctx *ctx) {
    if (!ctx->plan_unchecked) return 0;
    NP_HIP(hipEventSynchronize(ctx->plan_done));
    ctx->plan_unchecked = false;
    if (ctx->h_plan_err[0] != 0u) {
        g_err = plan_stall_message(ctx, ctx->plan_unchecked_mode, false);
        NP_HIP(hipMemset(ctx->d_plan_err, 0, sizeof(unsigned) * 4));
        return NP_E_PLANNING_STALLED_LOST;
    }
    return 0;
}
}  // namespace

// All iterations in one launch of the persistent kernel (np_planning.hip).  mode: NP_PLANNING_PERSISTENT (one workgroup per tile) or
// NP_PLANNING_PERSISTENT_QUEUE (resident workgroups pull (tile, iteration) items).
static int planning_persistent(np_f16_ctx *ctx, int64_t n, const np_f16_io *io, const np_planning_loop *lp, hipStream_t st, int mode, int waves, int block, int check) {
    if (!io->s || !io->u || !io->tgt || !io->step_count || !io->reward || !io->coef_cache)
        return fail("np_planning_inner_loop (persistent): needs state, reward and coef_cache buffers");
    if (io->ld < n) return fail("ld < n");
    if (io->ld >= (1ll << 30)) return fail("ld must be below 2^30 rows (32-bit byte offsets inside a row-indexed array)");
    if (((uintptr_t)lp->rnn[0] | (uintptr_t)lp->rnn[1] | (uintptr_t)lp->actor_weights) & 15) return fail("rnn buffers / actor weights must be 16-byte aligned");
    const bool i8 = lp->actor_weights_floats == NP_ACTOR_I8_NUM_FLOATS;
    if (i8 && mode == NP_PLANNING_PERSISTENT_DUAL) return fail("np_planning_inner_loop: the dual workgroups serve the fp32 controller only; use _PERSISTENT, _GUESTS, _QUEUE or _LAUNCHES with the block-fixed-point weights");
    PlanArgs pa;
    KArgs &a = pa.k;
    a.s = io->s; a.u = io->u; a.tgt = io->tgt; a.ld = io->ld; a.step_count = (long long *)io->step_count;
    a.fin0 = a.fin1 = a.fin2 = nullptr; a.fout0 = a.fout1 = a.fout2 = nullptr;
    a.action = lp->ll_act; a.act_stride = 4; a.obs = io->obs; a.reward = io->reward;
    a.rand_u = nullptr; a.noise = nullptr; a.inner = 1; a.cache = io->coef_cache; a.seed = io->seed; a.call_idx = io->call_idx;
    a.call_idx_base = io->call_idx_base;
    a.term_counters = io->term_counters; a.term_reasons = io->term_reasons; a.reward_task = io->reward_task;
    a.ll_tgt = lp->ll_tgt; a.ll_obs = nullptr;
    a.row0 = io->row0; a.n = n; a.cfg = ctx->cfg; a.reset_coef = ctx->d_reset_coef; a.wt = ctx->wt; a.trace = nullptr;
    a.cus = ctx->num_cus;
    pa.actor_w = lp->actor_weights;
    pa.ll_obs[0] = lp->ll_obs[0]; pa.ll_obs[1] = lp->ll_obs[1];
    pa.rnn[0] = lp->rnn[0]; pa.rnn[1] = lp->rnn[1];
    pa.masks = lp->masks; pa.ll_act = lp->ll_act;
    pa.flags[0] = lp->flags[0]; pa.flags[1] = lp->flags[1];
    pa.final_obs = io->obs;
    pa.iterations = lp->iterations;
    pa.cache_valid0 = io->cache_valid ? 1 : 0;
    pa.tiles = (n + PLAN_ROWS - 1) / PLAN_ROWS;
    pa.queue = nullptr;
    pa.queue_base = pa.flag_base = 0;
    pa.block = block > 0 ? (block < lp->iterations ? block : lp->iterations) : 1;
    pa.guest_blocks = 0;
    pa.err = pa.err_host = nullptr;
    pa.wait_ticks = 0;
    pa.debug_stall = 0;
    if ((mode == NP_PLANNING_PERSISTENT_QUEUE || mode == NP_PLANNING_PERSISTENT_GUESTS) && stream_is_capturing(st))
        // the item counter's / progress words' bases are host state baked into the kernel arguments: a second replay of the captured launch
        // would find the counter past its items and the words already raised — silently wrong results
        return fail("np_planning_inner_loop: the guest and queue schedules cannot be captured into a graph (their bases advance per launch on the host); "
                    "use NP_PLANNING_PERSISTENT, _DUAL or _LAUNCHES on a capturing stream");
    unsigned grid = (unsigned)pa.tiles;
    if (mode == NP_PLANNING_PERSISTENT_GUESTS) {
        // tiles beyond the resident workgroups are guests: cut into as many blocks as there are hosts to go round (np_planning.hip)
        const int per_cu = planning_persistent_workgroups_per_cu(ctx->task, waves, i8);
        if (per_cu <= 0) return fail("np_planning_inner_loop (persistent): occupancy query failed");
        const int64_t resident = (int64_t)per_cu * ctx->num_cus;
        const int64_t guests = pa.tiles - resident;
        if (guests <= 0) mode = NP_PLANNING_PERSISTENT;
        else if (guests > resident) return fail("np_planning_inner_loop (guests): more than two tiles per resident workgroup; use the queue schedule");
        else {
            // blocks per guest: as many as there are hosts to go round, but no more than 7 — a guest tile changes CU between blocks (an export and an
            // import through the coherence point: ~0.85 of an iteration), so its own chain is iterations + 0.85 B while a host's is iterations +
            // iterations / B: B = 25 for ten guests (n = 8 500) made the guests' chain the makespan (1.97 ms; the queue 1.70)
            int64_t b = resident / guests;
            if (const char *e = std::getenv("NP_PLANNING_GUEST_BLOCKS")) b = atoi(e) > 0 ? atoi(e) : b;   // experiments
            else if (b > 7) b = 7;
            if (b * guests > resident) b = resident / guests;
            pa.guest_blocks = (int)(b < lp->iterations ? b : lp->iterations);
            pa.block = block >= 0 && block <= lp->iterations ? block : 1;   // slack (iterations) per block index
        }
    }
    if (mode == NP_PLANNING_PERSISTENT_QUEUE || mode == NP_PLANNING_PERSISTENT_GUESTS) {
        const int per_cu = planning_persistent_workgroups_per_cu(ctx->task, waves, i8);
        if (per_cu <= 0) return fail("np_planning_inner_loop (persistent): occupancy query failed");
        const int64_t resident = (int64_t)per_cu * ctx->num_cus;
        if (pa.tiles * (int64_t)lp->iterations >= (1ll << 31)) return fail("np_planning_inner_loop (queue): too many items");
        if (ctx->queue_cap < 1 + pa.tiles) {
            if (ctx->d_queue) NP_HIP(hipFree(ctx->d_queue));
            ctx->d_queue = nullptr;
            ctx->queue_cap = 0;
            NP_HIP(hipMalloc((void **)&ctx->d_queue, sizeof(unsigned) * (size_t)(1 + pa.tiles)));
            ctx->queue_cap = 1 + pa.tiles;
            ctx->queue_dirty = true;
        }
        // the queue words are cleared once, not per call (a memset node in front of every macro-step costs ~10 us): the item counter runs
        // on and the progress words carry a base that moves past whatever the previous launch left (np_planning.hip); cleared again after a
        // failed launch and long before the bases could wrap
        if (ctx->queue_dirty || ctx->flag_base > (1u << 30) || ctx->queue_next > (1u << 30)) {
            NP_HIP(hipMemsetAsync(ctx->d_queue, 0, sizeof(unsigned) * (size_t)ctx->queue_cap, st));
            ctx->queue_dirty = false;
            ctx->queue_next = ctx->flag_base = 0;
        }
        pa.queue = ctx->d_queue;
        pa.queue_base = ctx->queue_next;
        pa.flag_base = ctx->flag_base;
        grid = (unsigned)(pa.tiles < resident ? pa.tiles : resident);
        const int per = pa.block > 0 ? pa.block : 1;
        if (pa.guest_blocks == 0) ctx->queue_next += (unsigned)(pa.tiles * ((lp->iterations + per - 1) / per)) + grid;   // every workgroup draws one id past the end
        ctx->flag_base += (unsigned)lp->iterations + 1u;
        ctx->queue_dirty = true;   // until the launch below is enqueued
    }
    if (mode == NP_PLANNING_PERSISTENT_DUAL) {   // two tiles per eight-wave workgroup; any n (workgroups beyond the resident ones queue up)
        NP_HIP(launch_planning_dual(ctx->task, pa, (unsigned)((pa.tiles + 1) / 2), st));
        return 0;
    }
    if (pa.queue) {
        const char *mode_name = mode == NP_PLANNING_PERSISTENT_GUESTS ? "guests" : "queue";
        if (plan_err_buffers(ctx)) return 1;
        pa.err = ctx->d_plan_err;
        pa.err_host = ctx->h_plan_err_dev;
        double wait_ms = 2000.0;   // one wait; an iteration is ~40 us, a block of a guest tile ~0.5 ms
        if (const char *e = std::getenv("NP_PLANNING_WAIT_MS")) wait_ms = atof(e) > 0.0 ? atof(e) : wait_ms;
        pa.wait_ticks = (unsigned long long)(wait_ms * 1e5);   // wall_clock64(): 100 MHz
        if (const char *e = std::getenv("NP_PLANNING_DEBUG_STALL")) pa.debug_stall = atoi(e) != 0;   // fault injection (tests/test_gpu_actor.py)
        // check = sync: keep what the kernel updates in place until its verdict is in
        PlanSnapshot snap;
        if (check == NP_PLANNING_CHECK_SYNC) {
            snap.add(io->s, sizeof(float) * (size_t)(11 * io->ld + n));
            snap.add(io->u, sizeof(float) * (size_t)(3 * io->ld + n));
            snap.add(io->step_count, sizeof(int64_t) * (size_t)n);
            snap.add(io->coef_cache, sizeof(float) * (size_t)np_f16_cache_floats(n));
            snap.add(lp->flags[0], (size_t)(3 * n));
            snap.add(lp->rnn[0], sizeof(float) * (size_t)n * npact::HID);
            snap.add(lp->ll_obs[0], sizeof(float) * (size_t)n * npact::OBS);
            snap.add(io->term_reasons, (size_t)n);
            snap.add(io->term_counters, sizeof(uint32_t) * NP_NUM_TERM_COUNTERS);
            if (ctx->snap_cap < snap.bytes) {
                if (ctx->d_snap) NP_HIP(hipFree(ctx->d_snap));
                ctx->d_snap = nullptr;
                ctx->snap_cap = 0;
                NP_HIP(hipMalloc((void **)&ctx->d_snap, snap.bytes));
                ctx->snap_cap = snap.bytes;
            }
            snap.bind(ctx->d_snap);
            NP_HIP(snap.run(false, st));
        }
        // The guest and queue schedules need every workgroup of the grid resident at once (a host spins on a progress word until the
        // workgroup that owns the previous block raises it).  Two such kernels running side by side — two contexts stepped from two
        // streams — could each hold the CUs the other's missing workgroups need: they are therefore chained, per device, by an event
        // (stream order already serialises launches on one stream).  Other kernels beside them only delay them: they end.
        {
            static std::mutex mu;
            static hipEvent_t last_ev[64] = {};
            static hipStream_t last_st[64] = {};
            std::lock_guard<std::mutex> lock(mu);
            const int dev = ctx->device;
            const bool chain = dev >= 0 && dev < 64;
            if (chain && last_ev[dev] && last_st[dev] != st) NP_HIP(hipStreamWaitEvent(st, last_ev[dev], 0));
            const bool timed = ctx->timing;   // (these schedules are refused on a capturing stream)
            EventLease lease(ctx);
            if (timed && ctx->events.size() >= MAX_PENDING_EVENTS && resolve_events(ctx)) return 1;
            if (timed) NP_HIP(lease.take());
            NP_HIP(launch_planning_persistent(ctx->task, waves, i8, pa, grid, st, timed ? lease.ev.first : nullptr, timed ? lease.ev.second : nullptr));
            if (timed) lease.commit();
            ctx->queue_dirty = false;
            if (chain) {
                if (!last_ev[dev]) NP_HIP(hipEventCreateWithFlags(&last_ev[dev], hipEventDisableTiming));
                NP_HIP(hipEventRecord(last_ev[dev], st));
                last_st[dev] = st;
            }
        }
        if (check == NP_PLANNING_CHECK_DEFERRED) {   // the verdict is read by the context's next loop call / np_planning_check
            NP_HIP(hipEventRecord(ctx->plan_done, st));
            ctx->plan_unchecked = true;
            ctx->plan_unchecked_mode = mode_name;
            return 0;
        }
        NP_HIP(hipStreamSynchronize(st));
        if (ctx->h_plan_err[0] != 0u) {
            NP_HIP(snap.run(true, st));
            NP_HIP(hipMemsetAsync(ctx->d_plan_err, 0, sizeof(unsigned) * 4, st));
            NP_HIP(hipStreamSynchronize(st));
            g_err = plan_stall_message(ctx, mode_name, true);
            return NP_E_PLANNING_STALLED;
        }
        return 0;
    }
    {   // np_f16_set_timing: one sample per macro-step covers the whole persistent launch (these launches used to record nothing)
        const bool timed = ctx->timing && !stream_is_capturing(st);
        EventLease lease(ctx);
        if (timed && ctx->events.size() >= MAX_PENDING_EVENTS && resolve_events(ctx)) return 1;
        if (timed) NP_HIP(lease.take());
        NP_HIP(launch_planning_persistent(ctx->task, waves, i8, pa, grid, st, timed ? lease.ev.first : nullptr, timed ? lease.ev.second : nullptr));
        if (timed) lease.commit();
    }
    ctx->queue_dirty = false;
    return 0;
}

int np_planning_inner_loop(np_f16_ctx *ctx, int64_t n, const np_f16_io *io, const np_planning_loop *lp, void *stream) {
    if (!ctx || !io || !lp) return fail("null ctx/io/loop");
    if (ctx->combat) return fail("combat context");
    if (n <= 0 || lp->iterations <= 0) return 0;
    if (!lp->actor_weights || !lp->ll_obs[0] || !lp->ll_obs[1] || !lp->rnn[0] || !lp->rnn[1] || !lp->masks || !lp->ll_act || !lp->flags[0] ||
        !lp->flags[1] || !lp->ll_tgt)
        return fail("np_planning_loop: null buffer");
    if (lp->groups < 0 || lp->groups > 8) return fail("np_planning_loop: groups must be 0 (automatic) .. 8");
    if (lp->rnn[0] == lp->rnn[1] || lp->ll_obs[0] == lp->ll_obs[1] || lp->flags[0] == lp->flags[1])
        return fail("np_planning_loop: the two buffers of a ping-pong pair must differ");
    if (lp->mode < NP_PLANNING_AUTO || lp->mode > NP_PLANNING_PERSISTENT_DUAL) return fail("np_planning_loop: unknown mode");
    if (lp->waves != 0 && lp->waves != 8) return fail("np_planning_loop: waves must be 0 (automatic) or 8 (the four-wave builds of the persistent kernel were retired in ABI 15)");
    if (lp->actor_weights_floats != 0 && lp->actor_weights_floats != NP_ACTOR_NUM_FLOATS && lp->actor_weights_floats != NP_ACTOR_I8_NUM_FLOATS)
        return fail("np_planning_loop: actor_weights_floats must be 0 / NP_ACTOR_NUM_FLOATS (fp32 numerics) or NP_ACTOR_I8_NUM_FLOATS (block fixed point)");
    if (((uintptr_t)lp->actor_weights) & 15) return fail("np_planning_loop: actor_weights must be 16-byte aligned");
    if (lp->block < 0) return fail("np_planning_loop: block must be >= 0");
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard;
    NP_HIP(guard.enter(ctx->device));
    if (const int rc = plan_check_pending(ctx)) return rc;   // check = deferred: the previous guest / queue launch's verdict comes first
    {
        int mode = lp->mode, waves = lp->waves;
        if (const char *e = std::getenv("NP_PLANNING_MODE")) {  // read per call: benchmarks and the parity tests switch it
            mode = std::strcmp(e, "launches") == 0 ? NP_PLANNING_LAUNCHES : std::strcmp(e, "persistent") == 0 ? NP_PLANNING_PERSISTENT
                   : std::strcmp(e, "queue") == 0 ? NP_PLANNING_PERSISTENT_QUEUE : std::strcmp(e, "guests") == 0 ? NP_PLANNING_PERSISTENT_GUESTS : std::strcmp(e, "dual") == 0 ? NP_PLANNING_PERSISTENT_DUAL : mode;
        }
        if (const char *e = std::getenv("NP_PLANNING_WAVES")) waves = atoi(e) == 8 ? 8 : atoi(e) == 4 ? 4 : waves;
        const bool eligible = ctx->solver == 0 && !ctx->cfg.aero_1d_tables && io->coef_cache && !io->rand_u && !io->noise && io->reward && planning_persistent_built(ctx->task);
        if (mode >= NP_PLANNING_PERSISTENT && !eligible)
            return fail("np_planning_loop: the persistent kernel serves tracking contexts (the reference's PlanningEnv task) with the Euler solver and the MLP numerics, and needs coef_cache / reward buffers");
        if (mode == NP_PLANNING_AUTO) {
            // measured per size (profiles/r04_planning_modes.log, ms per PlanningEnv.step): up to one 32-row tile per resident workgroup the
            // persistent kernel with eight waves per tile (n = 8 192: 2.49 -> 2.07); up to 1.5 tiles per workgroup its guest schedule
            // (n = 1e4: 3.16 -> 2.6); up to two tiles per workgroup the dual workgroups (two tiles share an eight-wave workgroup and one FDM
            // step: n = 16 384 3.39 -> 3.21); beyond that the launches, whose 64-row controller tiles and row groups fill the chip better
            mode = NP_PLANNING_LAUNCHES;
            if (eligible && !stream_is_capturing(st)) {
                const int per_cu = planning_persistent_workgroups_per_cu(ctx->task, 8, lp->actor_weights_floats == NP_ACTOR_I8_NUM_FLOATS);
                static_assert(NP_PLANNING_LAUNCHES == npdispatch::PL_LAUNCHES && NP_PLANNING_PERSISTENT == npdispatch::PL_PERSISTENT &&
                                  NP_PLANNING_PERSISTENT_GUESTS == npdispatch::PL_GUESTS && NP_PLANNING_PERSISTENT_DUAL == npdispatch::PL_DUAL && NP_PLANNING_PERSISTENT_QUEUE == npdispatch::PL_QUEUE && PLAN_ROWS == 32,
                              "np_dispatch.h mirrors the NP_PLANNING_* numbers");
                mode = npdispatch::planning_mode(n, (int64_t)per_cu * ctx->num_cus, lp->actor_weights_floats == NP_ACTOR_I8_NUM_FLOATS);   // np_dispatch.h: by tiles per resident workgroup and controller numerics
                waves = 8;
            }
        }
        int block = lp->block;
        if (const char *e = std::getenv("NP_PLANNING_BLOCK")) block = atoi(e) > 0 ? atoi(e) : block;
        if (mode != NP_PLANNING_LAUNCHES) {
            int check = lp->check;
            if (const char *e = std::getenv("NP_PLANNING_CHECK")) check = std::strcmp(e, "deferred") == 0 ? NP_PLANNING_CHECK_DEFERRED : std::strcmp(e, "sync") == 0 ? NP_PLANNING_CHECK_SYNC : check;
            if (check != NP_PLANNING_CHECK_SYNC && check != NP_PLANNING_CHECK_DEFERRED) return fail("np_planning_loop: check must be NP_PLANNING_CHECK_SYNC or NP_PLANNING_CHECK_DEFERRED");
            return planning_persistent(ctx, n, io, lp, st, mode, waves ? waves : 8, mode == NP_PLANNING_PERSISTENT_GUESTS ? (block > 0 ? block : 1) : block > 0 ? block : 5, check);
        }
    }
    // automatic choice, measured per size (profiles/r03g_planning_groups.log; ms per PlanningEnv.step, one group -> the choice):
    // n = 1e4 3.51 -> 3.23, 16 384 3.73 -> 3.44, 20 000 5.67 -> 4.39, 24 576 5.86 -> 4.89, 28 672 6.84 -> 5.64, 32 768 6.92 -> 6.29,
    // 40 960 8.53 -> 7.76, 49 152 10.3 -> 9.2, 57 344 12.35 -> 10.45, 65 536 12.57 -> 11.81, 81 920 15.4 -> 14.9; one group is the
    // best up to 8 192 (one 32-row controller tile per CU) and above ~82 000 (throughput)
    int groups = lp->groups ? lp->groups : npdispatch::planning_groups(n, ctx->num_cus);
    if (stream_is_capturing(st)) groups = 1;  // a captured graph runs its branches one after the other
    const int64_t per = ((n + groups - 1) / groups + 63) / 64 * 64;  // rows per group: boundaries on cache / kernel tiles
    if (per * (groups - 1) >= n) groups = (int)((n + per - 1) / per);
    while ((int)ctx->group_streams.size() < groups - 1) {
        hipStream_t s2;
        NP_HIP(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        ctx->group_streams.push_back(s2);
    }
    while ((int)ctx->group_events.size() < groups) {
        hipEvent_t ev;
        NP_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        ctx->group_events.push_back(ev);
    }
    if (groups > 1) {
        NP_HIP(hipEventRecord(ctx->group_events[0], st));
        for (int g = 1; g < groups; g++) NP_HIP(hipStreamWaitEvent(ctx->group_streams[g - 1], ctx->group_events[0], 0));
    }
    int rc = 0;  // a launch that fails ends the enqueueing, but the streams are joined all the same
    for (int k = 0; k < lp->iterations && !rc; k++) {
        const bool last = k == lp->iterations - 1;
        const int a = k & 1, b = a ^ 1;
        for (int g = 0; g < groups && !rc; g++) {
            const int64_t r0 = g * per, m = (r0 + per <= n ? per : n - r0);
            hipStream_t sg = g == 0 ? st : ctx->group_streams[g - 1];
            if (np_actor_forward(lp->actor_weights, lp->actor_weights_floats ? lp->actor_weights_floats : NP_ACTOR_NUM_FLOATS, m, lp->ll_obs[a] + r0 * npact::OBS, lp->rnn[a] + r0 * npact::HID, lp->masks + r0,
                                 lp->ll_act + r0 * 4, lp->rnn[b] + r0 * npact::HID, ctx->device, (void *)sg)) {
                rc = 1;
                break;
            }
            np_f16_io q = *io;
            q.s = io->s + r0; q.u = io->u + r0; q.tgt = io->tgt + r0;
            q.step_count = io->step_count + r0;
            q.done_in = lp->flags[a] + r0; q.bad_in = lp->flags[a] + n + r0; q.timeout_in = lp->flags[a] + 2 * n + r0;
            q.done_out = lp->flags[b] + r0; q.bad_out = lp->flags[b] + n + r0; q.timeout_out = lp->flags[b] + 2 * n + r0;
            q.action = lp->ll_act + r0 * 4; q.act_stride = 4;
            q.obs = last && io->obs ? io->obs + r0 * npact::OBS : nullptr;
            q.reward = io->reward ? io->reward + r0 : nullptr;
            q.rand_u = nullptr; q.noise = nullptr;
            q.coef_cache = io->coef_cache ? io->coef_cache + (r0 / CACHE_TILE) * (int64_t)NUM_CACHE_ROWS * CACHE_TILE : nullptr;
            q.cache_valid = k == 0 ? io->cache_valid : (io->coef_cache ? 1 : 0);
            q.inner_step = 1;
            q.call_idx = io->call_idx + (uint64_t)k;
            q.row0 = io->row0 + r0;
            q.term_reasons = io->term_reasons ? io->term_reasons + r0 : nullptr;
            q.reward_task = io->reward_task ? io->reward_task + r0 : nullptr;
            q.ll_tgt = last ? nullptr : lp->ll_tgt + r0;
            q.ll_obs = last ? nullptr : lp->ll_obs[b] + r0 * npact::OBS;
            if (launch_env<true>(ctx, m, &q, (void *)sg)) rc = 1;
        }
    }
    for (int g = 1; g < groups; g++) {
        NP_HIP(hipEventRecord(ctx->group_events[g], ctx->group_streams[g - 1]));
        NP_HIP(hipStreamWaitEvent(st, ctx->group_events[g], 0));
    }
    return rc;
}

int np_rollout_insert(const np_rollout_step *q, int device, void *stream) {
    if (!q) return fail("null argument");
    if (!q->obs || !q->actions || !q->rewards || !q->masks || !q->bad_masks || !q->action_log_probs || !q->value_preds || !q->rnn_states_actor ||
        !q->rnn_states_critic || !q->obs_in || !q->actions_in || !q->rewards_in || !q->action_log_probs_in || !q->values_in || !q->rnn_states_actor_in ||
        !q->rnn_states_critic_in || !q->done_in || !q->bad_done_in || !q->exceed_time_limit_in)
        return fail("np_rollout_insert: null buffer");
    if (q->num_envs < 0 || q->num_agents <= 0 || q->step < 0 || q->obs_dim <= 0 || q->act_dim <= 0 || q->rnn_dim <= 0) return fail("np_rollout_insert: bad sizes");
    if (q->num_envs == 0) return 0;
    if (q->num_envs * q->num_agents >= (1ll << 31)) return fail("np_rollout_insert: too many rows for one launch");
    int ndev = 0;
    NP_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail("no such HIP device (this library has no CPU fallback)");
    DeviceGuard guard;
    NP_HIP(guard.enter(device));
    nproll::InsertArgs a;
    a.E = q->num_envs; a.A = q->num_agents; a.step = q->step;
    a.obs_dim = q->obs_dim; a.act_dim = q->act_dim; a.rnn_dim = q->rnn_dim;
    a.obs = q->obs; a.actions = q->actions; a.rewards = q->rewards; a.masks = q->masks; a.bad_masks = q->bad_masks; a.logp = q->action_log_probs;
    a.values = q->value_preds; a.rnn_a = q->rnn_states_actor; a.rnn_c = q->rnn_states_critic;
    a.obs_in = q->obs_in; a.act_in = q->actions_in; a.rew_in = q->rewards_in; a.logp_in = q->action_log_probs_in; a.val_in = q->values_in;
    a.rnn_a_in = q->rnn_states_actor_in; a.rnn_c_in = q->rnn_states_critic_in;
    a.done = q->done_in; a.bad = q->bad_done_in; a.tmo = q->exceed_time_limit_in;
    hipLaunchKernelGGL(nproll::insert_kernel, dim3((unsigned)(q->num_envs * q->num_agents)), dim3(nproll::INSERT_THREADS), 0, (hipStream_t)stream, a);
    NP_HIP(hipGetLastError());
    return 0;
}

int np_policy_act(const np_policy_step *q, int device, void *stream) {
    if (!q) return fail("null argument");
    const bool actor = (q->flags & NP_POLICY_ACTOR) != 0, critic = (q->flags & NP_POLICY_CRITIC) != 0, det = (q->flags & NP_POLICY_DETERMINISTIC) != 0;
    if (!actor && !critic) return fail("np_policy_act: flags select neither network");
    if (q->flags & ~(NP_POLICY_ACTOR | NP_POLICY_CRITIC | NP_POLICY_DETERMINISTIC)) return fail("np_policy_act: unknown flag");
    if (q->act_dim < 1 || q->act_dim > 4) return fail("np_policy_act: 1 to 4 continuous actions");
    if (q->n < 0) return fail("np_policy_act: bad size");
    if (q->obs_dim != 0 && q->obs_dim != 22 && q->obs_dim != 15) return fail("np_policy_act: obs_dim must be 22 (0) or 15");
    if (q->reserved_ != 0) return fail("np_policy_act: reserved_ must be 0");
    if (q->weights_floats != 0 && q->weights_floats != NP_ACTOR_NUM_FLOATS && q->weights_floats != NP_ACTOR_I8_NUM_FLOATS)
        return fail("np_policy_act: weights_floats must be 0 / NP_ACTOR_NUM_FLOATS (fp32 chains) or NP_ACTOR_I8_NUM_FLOATS (block fixed point)");
    if (!q->obs || (!q->masks && !q->prev_flags)) return fail("np_policy_act: null obs / masks");
    if (q->prev_flags && (!q->masks_out || !q->bad_masks_out)) return fail("np_policy_act: prev_flags needs masks_out and bad_masks_out");
    if (actor && (!q->actor_weights || !q->rnn_states_actor_in || !q->rnn_states_actor_out || !q->actions || !q->action_log_probs || (!det && !q->noise)))
        return fail("np_policy_act: null actor buffer");
    if (critic && (!q->critic_weights || !q->rnn_states_critic_in || !q->rnn_states_critic_out || !q->values)) return fail("np_policy_act: null critic buffer");
    uintptr_t al = 0;
    if (actor) al |= (uintptr_t)q->actor_weights | (uintptr_t)q->rnn_states_actor_in | (uintptr_t)q->rnn_states_actor_out;
    if (critic) al |= (uintptr_t)q->critic_weights | (uintptr_t)q->rnn_states_critic_in | (uintptr_t)q->rnn_states_critic_out;
    if (al & 15) return fail("np_policy_act: packed weights and recurrent states must be 16-byte aligned");
    if ((actor && q->rnn_states_actor_in == q->rnn_states_actor_out) || (critic && q->rnn_states_critic_in == q->rnn_states_critic_out))
        return fail("np_policy_act: recurrent states in and out may not alias");
    if (actor)
        for (int j = 0; j < q->act_dim; j++)
            if (!(q->std[j] > 0.0f)) return fail("np_policy_act: std must be positive");
    if (q->n == 0) return 0;
    {   // no input range may overlap an output range: actor and critic workgroups of the one launch run in any order (the one allowed
        // in-place write is the collector's: zeroing the recurrent-state INPUT of an env that ended, prev_flags mode)
        struct Range { const void *p; size_t bytes; };
        const size_t n = (size_t)q->n, od = (size_t)(q->obs_dim ? q->obs_dim : 22);
        Range in[8], out[8];
        int ni = 0, no = 0;
        in[ni++] = {q->obs, n * od * 4};
        if (q->masks) in[ni++] = {q->masks, n * 4};
        if (q->prev_flags) {
            in[ni++] = {q->prev_flags, n * 3};
            out[no++] = {q->masks_out, n * 4};
            out[no++] = {q->bad_masks_out, n * 4};
        }
        if (actor) {
            in[ni++] = {q->rnn_states_actor_in, n * 128 * 4};
            if (!det) in[ni++] = {q->noise, n * (size_t)q->act_dim * 4};
            out[no++] = {q->rnn_states_actor_out, n * 128 * 4};
            out[no++] = {q->actions, n * (size_t)q->act_dim * 4};
            out[no++] = {q->action_log_probs, n * 4};
        }
        if (critic) {
            in[ni++] = {q->rnn_states_critic_in, n * 128 * 4};
            out[no++] = {q->rnn_states_critic_out, n * 128 * 4};
            out[no++] = {q->values, n * 4};
        }
        auto overlap = [](const Range &a, const Range &b) {
            const uintptr_t a0 = (uintptr_t)a.p, b0 = (uintptr_t)b.p;
            return a0 < b0 + b.bytes && b0 < a0 + a.bytes;
        };
        for (int i = 0; i < ni; i++)
            for (int j = 0; j < no; j++)
                if (overlap(in[i], out[j])) return fail("np_policy_act: an input buffer overlaps an output buffer");
        for (int i = 0; i < no; i++)
            for (int j = i + 1; j < no; j++)
                if (overlap(out[i], out[j])) return fail("np_policy_act: two output buffers overlap");
    }
    int ndev = 0;
    NP_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail("no such HIP device (this library has no CPU fallback)");
    DeviceGuard guard;
    NP_HIP(guard.enter(device));
    nppol::ActArgs a = {};
    a.w[0] = q->actor_weights; a.w[1] = q->critic_weights;
    a.h_in[0] = q->rnn_states_actor_in; a.h_in[1] = q->rnn_states_critic_in;
    a.h_out[0] = q->rnn_states_actor_out; a.h_out[1] = q->rnn_states_critic_out;
    a.obs = q->obs; a.mask = q->masks; a.noise = q->noise;
    a.values = q->values; a.actions = q->actions; a.log_probs = q->action_log_probs;
    a.n = (long long)q->n; a.act_dim = q->act_dim; a.flags = q->flags; a.first_net = actor ? 0 : 1;
    a.obs_dim = q->obs_dim ? q->obs_dim : 22;
    a.prev = q->prev_flags; a.masks_out = q->masks_out; a.bad_masks_out = q->bad_masks_out;
    for (int j = 0; j < 4; j++) {
        a.std[j] = q->std[j];
        a.log_std[j] = q->log_std[j];
    }
    if (q->weights_floats == NP_ACTOR_I8_NUM_FLOATS)
        NP_HIP(npact8::launch_policy_act_i8(a, (hipStream_t)stream));
    else
        NP_HIP(nppol::launch_policy_act(a, (hipStream_t)stream));
    return 0;
}

int np_planning_check(np_f16_ctx *ctx) {
    if (!ctx) return fail("null ctx");
    DeviceGuard guard;
    NP_HIP(guard.enter(ctx->device));
    return plan_check_pending(ctx);
}

int np_rollout_returns(int64_t T, int64_t N, double gamma, double gae_lambda, int use_gae, int use_proper_time_limits,
                       const float *rewards, float *value_preds, const float *masks, const float *bad_masks, const float *next_value,
                       float *returns, int device, void *stream) {
    if (!rewards || !value_preds || !masks || !next_value || !returns) return fail("null argument");
    if (use_proper_time_limits && !bad_masks) return fail("use_proper_time_limits needs bad_masks");
    if (T < 0 || N < 0) return fail("negative size");
    if (N == 0) return 0;
    int ndev = 0;
    NP_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail("no such HIP device (this library has no CPU fallback)");
    DeviceGuard guard;
    NP_HIP(guard.enter(device));
    const float g = (float)gamma, gl = (float)(gamma * gae_lambda);
    const int threads = N >= nproll::MANY_COLUMNS ? nproll::THREADS : 64;
    const dim3 grid((unsigned)((N + threads - 1) / threads)), block(threads);
    hipStream_t st = (hipStream_t)stream;
#define NP_RET(G, P)                                                                                                              \
    do {                                                                                                                          \
        if (N >= nproll::MANY_COLUMNS)                                                                                            \
            hipLaunchKernelGGL((nproll::returns_kernel<G, P, nproll::U_MANY_COLUMNS>), grid, block, 0, st, (long long)T, (long long)N, g, gl, \
                               rewards, value_preds, masks, bad_masks, next_value, returns);                                      \
        else                                                                                                                      \
            hipLaunchKernelGGL((nproll::returns_kernel<G, P, nproll::U_FEW_COLUMNS>), grid, block, 0, st, (long long)T, (long long)N, g, gl,  \
                               rewards, value_preds, masks, bad_masks, next_value, returns);                                      \
    } while (0)
    if (use_gae && use_proper_time_limits) NP_RET(true, true);
    else if (use_gae) NP_RET(true, false);
    else if (use_proper_time_limits) NP_RET(false, true);
    else NP_RET(false, false);
#undef NP_RET
    NP_HIP(hipGetLastError());
    return 0;
}

int np_f16_set_kernel_variant(np_f16_ctx *ctx, int variant) {
    if (!ctx) return fail("null ctx");
    if (variant == NP_KERNEL_DUAL8 || variant == NP_KERNEL_DUAL4) {
        if (!ctx->combat) return fail("NP_KERNEL_DUAL8 / NP_KERNEL_DUAL4 are SingleCombat variants (np_f16_combat_ctx_create)");
    } else if (variant != NP_KERNEL_AUTO && variant != NP_KERNEL_LATENCY && variant != NP_KERNEL_THROUGHPUT && variant != NP_KERNEL_PAIR &&
               variant != NP_KERNEL_LATENCY8 && variant != NP_KERNEL_LATENCY2 && variant != NP_KERNEL_LATENCY4W)
        return fail("unknown kernel variant");
    ctx->variant = variant;
    return 0;
}

int np_f16_set_timing(np_f16_ctx *ctx, int enable) {
    if (!ctx) return fail("null ctx");
    ctx->timing = enable != 0;
    if (!enable) return 0;  // stop attaching events; what was recorded so far stays readable (a new series starts on enable)
    ctx->t_sum_ms = 0.0;
    ctx->t_count = 0;
    // pairs still pending belong to launches that may not have finished: wait for them before they go back to the pool (a
    // recycled pair would otherwise be re-recorded while its old recording is in flight)
    {
        DeviceGuard guard;
        NP_HIP(guard.enter(ctx->device));
        for (auto &e : ctx->events) {
            (void)hipEventSynchronize(e.second);
            ctx->pool.push_back(e);
        }
    }
    ctx->events.clear();
    ctx->t_sum_ms = 0.0;
    ctx->t_count = 0;
    ctx->samples.clear();
    return 0;
}

int np_f16_get_timing(np_f16_ctx *ctx, double *avg_ms, int64_t *count) {
    if (!ctx) return fail("null ctx");
    DeviceGuard guard;
    NP_HIP(guard.enter(ctx->device));
    if (resolve_events(ctx)) return 1;
    if (avg_ms) *avg_ms = ctx->t_count ? ctx->t_sum_ms / (double)ctx->t_count : 0.0;
    if (count) *count = ctx->t_count;
    return 0;
}

int np_selfcheck_divc(float c, uint64_t *counts3, int device) {
    if (!counts3) return fail("null argument");
    if (!(c == c) || c == 0.0f || std::isinf(c)) return fail("np_selfcheck_divc: the divisor must be a finite non-zero constant");
    int ndev = 0;
    NP_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail("no such HIP device (this library has no CPU fallback)");
    DeviceGuard guard;
    NP_HIP(guard.enter(device));
    unsigned long long *d = nullptr;
    NP_HIP(hipMalloc(&d, 3 * sizeof(unsigned long long)));
    hipError_t e = hipMemset(d, 0, 3 * sizeof(unsigned long long));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(divc_sweep_kernel, dim3(256 * 64), dim3(256), 0, 0, c, (float)(1.0 / (double)c), d);
        e = hipGetLastError();
    }
    unsigned long long h[3] = {0, 0, 0};
    if (e == hipSuccess) e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    NP_HIP(e);
    for (int k = 0; k < 3; k++) counts3[k] = h[k];
    return 0;
}

int np_f16_set_trace(np_f16_ctx *ctx, uint64_t *dev_buf, int64_t capacity_workgroups) {
    if (!ctx) return fail("null ctx");
    if (dev_buf && capacity_workgroups <= 0) return fail("np_f16_set_trace: capacity must be positive");
    ctx->trace = (unsigned long long *)dev_buf;
    ctx->trace_cap = dev_buf ? capacity_workgroups : 0;
    return 0;
}

int np_f16_get_timing_samples(np_f16_ctx *ctx, float *ms_out, int64_t capacity, int64_t *count) {
    if (!ctx) return fail("null ctx");
    if (capacity < 0 || (capacity > 0 && !ms_out)) return fail("bad sample buffer");
    if (np_f16_get_timing(ctx, nullptr, nullptr)) return 1;  // resolves the events still pending
    const int64_t n = (int64_t)ctx->samples.size(), m = n < capacity ? n : capacity;
    for (int64_t k = 0; k < m; k++) ms_out[k] = ctx->samples[(size_t)k];
    if (count) *count = n;
    return 0;
}

}  // extern "C"
