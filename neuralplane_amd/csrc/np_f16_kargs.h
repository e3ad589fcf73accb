// np_f16_kargs.h — what the env.step kernels (np_f16_kernels.hip) and the persistent PlanningEnv kernel (np_planning.hip) share:
// the kernel-argument record of one np_f16_step / np_f16_reset launch, the row-indexed addressing helper and the tile constants.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "np_f16_device.h"

#ifndef NPF16_BLOCK
#define NPF16_BLOCK 128
#endif
#ifndef NPF16_MINWAVES
#define NPF16_MINWAVES 3  // waves per SIMD the register allocator must leave room for
#endif

namespace npf16 {

constexpr int BLOCK = NPF16_BLOCK;
constexpr int CACHE_TILE = 64;  // rows per tile of the cross-step coefficient cache
constexpr int OBS_LD = 23;  // odd row pitch: conflict-free ds_write_b32 of a 22-float row per lane
// LDS scratch of a workgroup: first the per-lane columns of the 42 aero coefficients
// (coef[slot][lane]), later re-used as the [BLOCK][OBS_LD] observation transpose tile.
constexpr int LDS_FLOATS = (NUM_LDS_SLOTS * BLOCK > BLOCK * OBS_LD) ? NUM_LDS_SLOTS * BLOCK : BLOCK * OBS_LD;

// row-indexed access as (uniform base) + (32-bit BYTE offset): selects the SGPR-base + VGPR-offset addressing mode
// (every such array is device memory handed over through np_f16_io: the reference is typed as GLOBAL address space, so that pointers
// re-read from the kernel-argument segment — plain generic pointers to the compiler — do not turn into flat_store + a 64-bit VALU add)
template <class T>
__device__ __forceinline__ __attribute__((address_space(1))) T &at_off(T *base, unsigned byte_off) {
    typedef __attribute__((address_space(1))) T GT;
    return *reinterpret_cast<GT *>(reinterpret_cast<uintptr_t>(base) + byte_off);
}

struct KArgs;
typedef const KArgs __attribute__((address_space(4))) *KArgsC;
#ifndef NP_REREAD_ARGS
#define NP_REREAD_ARGS(ap) asm volatile("" : "+s"(ap) : : "memory")
#endif

struct KArgs {
    float *s, *u, *tgt;
    long long ld;
    long long *step_count;
    const uint8_t *fin0, *fin1, *fin2;
    uint8_t *fout0, *fout1, *fout2;
    const float *action;
    long long act_stride;
    float *obs, *reward;
    const float *rand_u, *noise;
    int inner;     // one low-level iteration of PlanningEnv.step: no auto-reset, flagged rows frozen, flags accumulate
    int cus;       // multiProcessorCount of the launching context's device: the de-phasing rules count generations of resident workgroups
    float *cache;  // [row / 64][14][row % 64] force-side alpha/beta-only coefficients at the current state (may be null)
    uint64_t seed, call_idx;
    const uint64_t *call_idx_base;  // optional device word added to call_idx (launches replayed from a HIP graph)
    unsigned *term_counters;        // optional [NP_NUM_TERM_COUNTERS] per-condition counters (one atomic per wave and condition)
    unsigned char *term_reasons;    // optional [n]: the same conditions per aircraft, bit k = counter k
    float *reward_task;             // optional [n]: the task's reward function alone (reward = this + the event term)
    const float *ll_tgt;            // INNER, optional [3][ld]: the low-level controller's targets
    float *ll_obs;                  // INNER, optional [n][22]: PlanningEnv.low_level_obs of the state this launch reaches
    long long row0, n;
    DevCfg cfg;
    // the 14 cached (force-side alpha/beta-only) coefficients of a freshly reset aircraft (alpha = beta = 0), evaluated once
    // per context by the same device code (f16_reset_coef_kernel), so they are bit-identical to an in-line evaluation;
    // per context (a small device buffer, not __constant__): contexts with different numerics options can coexist on a
    // device; read only inside the `flagged` branches
    const float *reset_coef;
    AeroWeights wt;
    // profiling hook (np_f16_set_trace; null otherwise): per workgroup NP_TRACE_WORDS 64-bit words — shader-clock counter at
    // entry, after the de-phasing delay and at exit, the constant 100 MHz counter at entry and exit, XCC / CU / SIMD ids
    unsigned long long *trace;
};

}  // namespace npf16
