// np_dispatch.h — which kernel variant, tiling and row grouping a launch gets, as a pure function of (batch size, CU count, options).
//
// Every threshold below was measured on the 256-CU MI355X (SPX) and is a statement about tiles per CU or wave slots per SIMD: "one
// 64-aircraft tile per CU", "one generation at three waves per SIMD".  They are therefore written per CU and multiplied by the
// multiProcessorCount of the context's device (np_f16_ctx_create), so that a partitioned device (CPX: 32 CUs, DPX: 128) gets the same
// tiles-per-CU decisions; on 256 CUs they reproduce the measured numbers exactly (tests/test_abi_cpu.py pins both).  Speed only: every
// variant computes bit-identical results.  No HIP in here: the selection runs in CPU tests through np_dispatch_plan().
#pragma once
#include <cstdint>

namespace npdispatch {

constexpr int REF_CUS = 256;  // the device the thresholds were measured on

struct Limits {
    int64_t lat8_max_n;        // eight waves per 64-aircraft tile while every tile has a CU of its own            (256 CUs: 16 384)
    int64_t lat4_max_n;        // four waves per tile: one generation of three tiles per CU                          (49 152)
    int64_t lat4w_max_n;       // four waves per tile built for four waves per SIMD: four tiles per CU               (65 536)
    int64_t lat_max_n;         // two waves per tile: up to six tiles per CU in one generation; the pair variant above (98 304)
    int64_t combat_lat_max_n;  // SingleCombat: latency variant up to here (aircraft)                                (40 000)
    int64_t combat_dual8_min_n, combat_dual8_max_n;  // SingleCombat: eight waves per 128-aircraft tile, one tile per CU       (16 385 .. 32 768)
    int64_t combat_dual4_max_n;  // four waves per 128-aircraft tile, two tiles per CU                                      (.. 65 536)
    int64_t pair3_min_grid;    // pair variant at three waves per SIMD from more than four workgroups per CU         (grid > 1 024)
    int64_t actor_tile32_a, actor_tile32_b, actor_tile32_c, actor_tile32_d;  // np_actor_forward: 32-row tiles for n <= a, <= b and c < n <= d
    int64_t groups_n[6];       // np_planning_inner_loop, launch-by-launch: upper bounds of the 1, 2, 3, 4, 2, 3 row-group ranges
};

inline Limits limits_for(int cus) {
    const int64_t c = cus > 0 ? cus : REF_CUS;
    Limits l;
    l.lat8_max_n = 64 * c;
    l.lat4_max_n = 192 * c;
    l.lat4w_max_n = 256 * c;
    l.lat_max_n = 384 * c;
    l.combat_lat_max_n = 40000 * c / REF_CUS;
    l.combat_dual8_min_n = 64 * c + 1;   // up to one 64-aircraft tile per CU the four-wave latency kernel is as fast (r04_combat_dual_ab.log)
    l.combat_dual8_max_n = 128 * c;
    l.combat_dual4_max_n = 256 * c;
    l.pair3_min_grid = 4 * c;
    l.actor_tile32_a = 64 * c;   // 16 384: two 32-row tiles per CU
    l.actor_tile32_b = 104 * c;  // 26 624
    l.actor_tile32_c = 128 * c;  // 32 768
    l.actor_tile32_d = 168 * c;  // 43 008
    const int64_t g[6] = {32 * c, 64 * c, 104 * c, 144 * c, 208 * c, 320 * c};  // 8 192, 16 384, 26 624, 36 864, 53 248, 81 920
    for (int k = 0; k < 6; k++) l.groups_n[k] = g[k];
    return l;
}

// np_f16_combat_step, automatic choice, Euler, MLP numerics: waves per 128-aircraft tile of the dual family (0 = not the dual family)
inline int combat_dual_waves(int64_t n, int cus) {
    const Limits l = limits_for(cus);
    return n < l.combat_dual8_min_n ? 0 : n <= l.combat_dual8_max_n ? 8 : n <= l.combat_dual4_max_n ? 4 : 0;
}

// np_f16_step (env kernels).  variant_pin: NP_KERNEL_* (0 = automatic); forced_latency / forced_pair / forced_off: the process-wide
// NPF16_KERNEL overrides as np_f16_kernels.hip reads them.
struct EnvChoice {
    int pair, pair3, latency, latency8, latency2, latency4w;
    int64_t grid;
    int block;
};

// kernel-variant numbers as include/neuralplane_amd.h's NP_KERNEL_* (static_asserts in np_f16_kernels.hip keep them equal)
enum { K_AUTO = 0, K_LATENCY = 1, K_THROUGHPUT = 2, K_PAIR = 3, K_LATENCY8 = 4, K_LATENCY2 = 5, K_LATENCY4W = 6 };
constexpr int LAT_TILE = 64;

// step: np_f16_step (false: np_f16_reset); solver 0 Euler / 1 rk4; tables: the 1-D table numerics; variant: the context's pin (K_AUTO = none);
// env_kernel: NPF16_KERNEL as a variant number (K_AUTO = unset; latency, throughput or pair); pair_waves_env: NPF16_PAIR_WAVES (0 = unset);
// block: NPF16_BLOCK (128)
inline EnvChoice env_choice(int64_t n, int cus, bool step, int solver, bool tables, int variant, int env_kernel, int pair_waves_env, int block) {
    const Limits l = limits_for(cus);
    EnvChoice c = {};
    const bool pair_auto = env_kernel == K_PAIR || (env_kernel == K_AUTO && n > l.lat_max_n);
    c.pair = step && (variant != K_AUTO ? variant == K_PAIR : pair_auto);
    const int v = variant != K_AUTO ? variant : (env_kernel == K_LATENCY || env_kernel == K_THROUGHPUT) ? env_kernel : K_AUTO;
    const bool lat_family = v != K_AUTO ? (v == K_LATENCY || v == K_LATENCY8 || v == K_LATENCY2 || v == K_LATENCY4W) : n <= l.lat_max_n;
    c.latency = !c.pair && step && solver == 0 && lat_family;
    c.latency8 = c.latency && !tables && (variant == K_LATENCY8 || (variant == K_AUTO && n <= l.lat8_max_n));
    c.latency2 = c.latency && !c.latency8 && (variant == K_LATENCY2 || (variant == K_AUTO && n > l.lat4w_max_n));
    c.latency4w = c.latency && !c.latency8 && !c.latency2 && !tables && (variant == K_LATENCY4W || (variant == K_AUTO && n > l.lat4_max_n));
    c.grid = c.latency ? (n + LAT_TILE - 1) / LAT_TILE : (n + block - 1) / block;
    c.block = c.latency8 ? LAT_TILE * 8 : c.latency2 ? LAT_TILE * 2 : c.latency ? LAT_TILE * 4 : block;
    c.pair3 = c.pair && (pair_waves_env ? pair_waves_env == 3 : c.grid > l.pair3_min_grid);
    return c;
}

// np_planning_inner_loop, automatic mode (Euler, MLP numerics, cache / reward buffers present, stream not capturing): which schedule of the
// persistent kernel — by 32-row tiles per resident eight-wave workgroup (one per CU: 248 VGPRs) — or the launches.  Numbers as
// include/neuralplane_amd.h's NP_PLANNING_* (static_assert in np_f16_kernels.hip).  Measured on 256 CUs (profiles/r04_planning_modes*.log,
// r04_planning_dual.log): n = 8 192 2.49 -> 2.05 ms (one tile per workgroup), 1e4 3.16 -> 2.6 (guest schedule, up to 1.5 tiles per
// workgroup), 16 384 3.39 -> 3.2 (dual workgroups, up to two); beyond that the launches (64-row controller tiles, row groups).
// With the block-fixed-point controller (round 5; the dual workgroups serve the fp32 controller only) the window between 1.5 and 2 tiles per
// workgroup goes to the queue up to 1.75 and to the guest schedule (one 50-iteration block per host) from there (ms per PlanningEnv.step, queue /
// guests / launches: n = 13 000 2.50 / 2.91 / 3.11, 14 000 2.70 / 2.88 / 3.20, 16 384 3.12 / 2.89 / 3.73; profiles/r05_planning_modes_i8.log).
enum { PL_LAUNCHES = 1, PL_PERSISTENT = 2, PL_QUEUE = 3, PL_GUESTS = 4, PL_DUAL = 5 };
inline int planning_mode(int64_t n, int64_t resident_workgroups, bool i8_controller = false) {
    const int64_t tiles = (n + 31) / 32, r = resident_workgroups;
    if (r <= 0) return PL_LAUNCHES;
    const int m = tiles <= r ? PL_PERSISTENT : tiles - r <= r / 2 ? PL_GUESTS : tiles <= 2 * r ? PL_DUAL : PL_LAUNCHES;
    if (m == PL_DUAL && i8_controller) return 4 * tiles <= 7 * r ? PL_QUEUE : PL_GUESTS;
    return m;
}

inline int planning_groups(int64_t n, int cus) {
    const Limits l = limits_for(cus);
    return n <= l.groups_n[0] ? 1 : n <= l.groups_n[1] ? 2 : n <= l.groups_n[2] ? 3 : n <= l.groups_n[3] ? 4 : n <= l.groups_n[4] ? 2 : n <= l.groups_n[5] ? 3 : 1;
}

inline bool actor_tile32(int64_t n, int cus) {
    const Limits l = limits_for(cus);
    return n <= l.actor_tile32_a || n <= l.actor_tile32_b || (n > l.actor_tile32_c && n <= l.actor_tile32_d);
}

}  // namespace npdispatch
