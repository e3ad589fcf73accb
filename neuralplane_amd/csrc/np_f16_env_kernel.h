// np_f16_env_kernel.h — the fused F-16 env.step / env.reset kernel template (reference: envs/env_base.py:83-109), shared by the
// translation units that instantiate it: np_env_t{0,1,2}s{0,1}.hip (one per task x solver, built in parallel; np_env_tu.inc) — the host
// side (context, dispatch rule, C ABI) is np_f16_kernels.hip, which hands a launch over through np_env_launch.h.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdint>

#include "../../include/neuralplane_amd.h"
#include "np_f16_device.h"
#include "np_f16_kargs.h"

namespace npf16 {

#ifndef NPF16_BLOCK
#define NPF16_BLOCK 128
#endif
// pair variant, de-phasing of the first generation: (blockIdx % groups) x stagger cycles; groups odd (co-resident workgroups differ
// by powers of two in index).  Two builds of the step kernel: PW = 2 waves per SIMD (an occupancy cap) and PW = 3; since round 3 both
// Euler builds need 159 VGPRs and no scratch (rk4: 184 / 0 B and 168 / 64 B per lane) — the numbers of the library at hand are printed by
// tools/code_object_table.py from the code object itself (DESIGN.md carries that table); launch_env picks per grid size.
#ifndef NPF16_PAIR_STAGGER
#define NPF16_PAIR_STAGGER 14000  // PW = 2
#endif
#ifndef NPF16_PAIR_GROUPS
#define NPF16_PAIR_GROUPS 3       // PW = 2
#endif
#ifndef NPF16_PAIR3_STAGGER
#define NPF16_PAIR3_STAGGER 7000  // PW = 3 (round 2: 7 x 9 000, profiles/r02b_ab_sessions.md s25; re-tuned on the round-3 kernel, profiles/r03b_ab_dephasing.log: 9 x 7 000 -1.1 % at N = 1e6, -0.7 % at 1e7)
#endif
#ifndef NPF16_PAIR3_GROUPS
#define NPF16_PAIR3_GROUPS 9      // PW = 3
#endif
#ifndef NPF16_MINWAVES
#define NPF16_MINWAVES 3  // waves per SIMD the register allocator must leave room for
#endif
#ifndef NPF16_STAGGER_CYCLES
#define NPF16_STAGGER_CYCLES 20000  // ~10 us at 2 GHz per phase step; 0 disables the de-phasing
#endif
#ifndef NPF16_ONEGEN_DELAY
#define NPF16_ONEGEN_DELAY 16000  // cycles; 0 disables (see f16_env_kernel)
#endif
#ifndef NPF16_STAGGER_MIN_GENS
#define NPF16_STAGGER_MIN_GENS 2  // de-phase only grids of at least this many generations (experiments: 0 = every grid)
#endif
#ifndef NPF16_STAGGER_CYCLES_LONG
#define NPF16_STAGGER_CYCLES_LONG 30000  // grids of 8 generations and more (N >= 1.6e6 on 256 CUs), see the kernel
#endif
// STEP=true : BaseEnv.step  (env_base.py:99-109)
// STEP=false: BaseEnv.reset (env_base.py:83-97)
// CACHED    : a.cache holds, for every row, the 14 force-side alpha/beta-only coefficients of its CURRENT
//             state (written by the previous step's Overload evaluation) -> the integrator skips them.
// TILE, WPT : aircraft per workgroup and how its waves cooperate.  Results are bit-identical between the variants.
//             WPT = 1, TILE = 128  throughput variant: two independent waves (rk4 fallbacks, the 1-D table mode).
//             WPT = 2, TILE = 128  pair variant (default for large batches): each wave owns 64 aircraft, but the two waves split
//                                  the NETS of every evaluation and evaluate their half for both waves' aircraft (dual asm
//                                  bodies, eval_nets in np_f16_device.h); inputs and coefficients cross through LDS.
//             WPT = 4, TILE = 64   latency variant (small batches): four waves hold the SAME 64 aircraft, split the net
//                                  evaluations and redo the cheap non-MLP arithmetic redundantly, so a step takes ~1/2 of a
//                                  lone wave's time; wave 0 stores.
// INNER     : one low-level iteration of PlanningEnv.step (np_f16_io.inner_step): no auto-reset, flagged rows keep their state, flags
//             accumulate.  A template parameter so that the plain env.step carries none of its selects (~30 VALU instructions).
template <int TASK, int SOLVER, bool STEP, bool CACHED, int TILE = BLOCK, int WPT = 1, bool INNER = false, int PW = 2>
// waves per SIMD the register allocator builds for: pair variant PW (PW = 2 is also an occupancy CAP — 159 VGPRs would fit three — for the
// grid sizes where six workgroups per CU are placed unevenly), latency4w four, everything else NPF16_MINWAVES
__global__ __launch_bounds__(TILE * lat_waves(WPT))
__attribute__((amdgpu_waves_per_eu((WPT == 2 ? PW : (WPT == 4 && PW == 4) ? 4 : NPF16_MINWAVES), (WPT == 2 && PW == 2 ? 2 : 8))))
void f16_env_kernel(const KArgs a) {
    // latency variant with shared scalar work (Euler step): wave w computes a quarter of the tile's serial fp64 chains and of its
    // observation noise for ALL four waves (np_f16_device.h::nlplant<.., SHARE>), wave 0 finishes the observation, wave 1 the
    // terminations / reward / state stores
    constexpr bool SHARED = WPT >= 4 && STEP && SOLVER == 0;
    constexpr int NOISE_COL0 = NUM_LDS_SLOTS + 2 * NUM_SHARED_SCALARS;  // 11 pairs x (radius x scale, cos, sin)
    constexpr int STATE_WAVE = SHARED ? 1 : 0;  // which wave of the latency variant stores state / flags / reward
    // pair variant (WPT == 2): nine more columns carry the normalised MLP inputs to the other wave of the workgroup
    constexpr int COLS = NUM_LDS_SLOTS + (WPT == 2 ? NUM_NORM_GROUPS : 0) + (SHARED ? 2 * NUM_SHARED_SCALARS + 33 : 0);
    constexpr int TILE_LDS = (COLS * TILE > TILE * OBS_LD) ? COLS * TILE : TILE * OBS_LD;
    __shared__ __attribute__((aligned(16))) float lds[TILE_LDS];
    float *obs_tile = lds;
    const int t = WPT < 4 ? (int)threadIdx.x : (int)(threadIdx.x % TILE);
    const int part = WPT < 4 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x / TILE));  // wave-uniform: which of the tile's 4 / 8 waves
    // what the net evaluation calls `part`: latency variant = which quarter of the nets, pair variant = which wave of the pair
    const int pw = WPT == 2 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x / 64)) : part;
    float *coef = lds + t;  // this lane's coefficient column, stride TILE
    const long long i0 = (long long)blockIdx.x * TILE;
    const long long i = i0 + t;
    const bool valid = i < a.n;
    const long long ic = valid ? i : a.n - 1;  // tail lanes shadow the last row; their stores are masked
    // The two MLP phases are asm statements that own s2-s101.  Whatever scalar value is live across them (the ~60 dwords of
    // scenario constants, the output pointers, ...) would be parked in VGPR lanes and fetched back with v_readlane — ~680 VALU
    // slots per step.  Instead every phase re-reads what it needs from the kernel-argument segment through `ap`, a pointer the
    // compiler cannot see through (NP_REREAD_ARGS) — scalar loads, off the VALU's critical path.
    KArgsC ap = (KArgsC)__builtin_amdgcn_kernarg_segment_ptr();  // `a` is the only kernel parameter: offset 0 of the segment
    const bool tables = a.cfg.aero_1d_tables != 0;
    const uint64_t call_idx = a.call_idx + (a.call_idx_base ? *a.call_idx_base : 0ull);
#ifdef NPF16_LAT_TRACE  // experiment builds only (tools/microbench/lat_trace.py): 100 MHz stamps at the phase boundaries of every wave
    unsigned long long lt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define NP_LT(k) lt[k] = wall_clock64()
#else
#define NP_LT(k)
#endif
    NP_LT(0);
    // profiling hook: the stamps go to the workgroup's record as they are taken (nothing stays live across the kernel; a.trace is
    // wave-uniform and null outside profiling runs)
    if (a.trace && threadIdx.x == 0) {
        unsigned long long *rec = a.trace + (unsigned long long)blockIdx.x * NP_TRACE_WORDS;
        rec[0] = __builtin_readcyclecounter();
        rec[3] = wall_clock64();
    }

    // ---- de-phasing -----------------------------------------------------------------------------------
    // Every workgroup does identical work: load state (HBM) -> ~45 K VALU cycles -> store.  Launched
    // together, the 3 waves that share a SIMD stay in lock-step, so the chip alternates between "everybody
    // waits for HBM" and "everybody computes" and the memory time ADDS to the compute time (measured 0.50 ms
    // vs 0.43 ms per step at N = 1e6, A/B in one session).  Delaying the workgroups of the first generation by
    // (blockIdx % 3) x ~10 us spreads the phases (co-resident waves come from workgroups whose indices differ
    // by a power of two, so % 3 separates them; keying on the SIMD wave-slot id of HW_REG_HW_ID measured
    // worse: 0.45 ms); later workgroups inherit the phase of the workgroup whose slot they take.  Only done
    // for grids that run several generations (large N), where 20 us is noise; dispatch order is an
    // assumption that affects speed only, never results.
    // workgroups resident at once = CUs (a.cus: multiProcessorCount of the launching context's device) x 4 SIMDs x wave slots / waves per workgroup
    const int cus = a.cus;
    const int FIRST_GEN = cus * 4 * (WPT == 2 ? PW : NPF16_MINWAVES) / (BLOCK / 64);
    const int FIRST_GENERATION_RT = cus * 4 * NPF16_MINWAVES / (BLOCK / 64);
    // Round 3 (profiles/r03f_mid_large_n.log, one session): the three-wave pair build gains from the delay on EVERY grid it is used
    // for, also a single or a partial generation that fills the chip (1 025 .. 3 071 workgroups: 163 840 aircraft 73.2 -> 70.6 us,
    // 196 608 86.4 -> 82.6, 229 376 124.2 -> 88.9, 262 144 106.5 -> 102.4, 327 680 126.4 -> 118.9) — unlike the two-wave build
    // on its half-filled chip (profiles/r03a_one_generation_dephasing.json: nothing).
    constexpr bool ALWAYS = WPT == 2 && PW == 3;
    if (WPT < 4 && STEP && NPF16_STAGGER_CYCLES > 0 && (ALWAYS ? (int)gridDim.x > 4 * cus : (int)gridDim.x >= NPF16_STAGGER_MIN_GENS * FIRST_GEN) &&
        (int)blockIdx.x < FIRST_GEN) {
        // The phase pattern has to survive the whole grid: measured on the current build (A/B in one session, 20 000 / 30 000 /
        // 40 000 cycles): N = 1e6 (5 generations) 0.393 / 0.398 / 0.399 ms — the delay itself is visible — but N = 3e6
        // 1.257 / 1.115 / 1.111 ms and N = 1e7 3.78-4.04 / 3.37-3.40 / 3.36-3.38 ms; the curves cross at ~8 generations.
        const long long unit = (int)gridDim.x >= 8 * FIRST_GENERATION_RT ? NPF16_STAGGER_CYCLES_LONG : NPF16_STAGGER_CYCLES;
        // pair variant: NPF16_PAIR_GROUPS phase groups, NPF16_PAIR_STAGGER cycles apart
        const long long wait = WPT != 2 ? (long long)(blockIdx.x % 3) * unit
                               : PW == 3 ? (long long)(blockIdx.x % NPF16_PAIR3_GROUPS) * NPF16_PAIR3_STAGGER
                                         : (long long)(blockIdx.x % NPF16_PAIR_GROUPS) * NPF16_PAIR_STAGGER;
        const long long t0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - t0 < wait) __builtin_amdgcn_s_sleep(32);
    }
    // a grid that fills the chip exactly once with the two-wave pair variant (897..1024 workgroups, four per CU: e.g. N = 131 072)
    // runs its co-resident workgroups in lock-step (all load, all compute, all store); every second generation of 256 starts
    // NPF16_ONEGEN_DELAY cycles late: 64.8 -> 60.0 us at N = 131 072, nothing below 897 workgroups (profiles/r03e_onegen_phase_ab.log)
    // (on 256 CUs: 897..1 024 workgroups, blockIdx & 256)
    if (WPT == 2 && PW == 2 && STEP && NPF16_ONEGEN_DELAY > 0 && 2 * (int)gridDim.x > 7 * cus && (int)gridDim.x <= 4 * cus && (((int)blockIdx.x / cus) & 1)) {
        const long long t0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - t0 < NPF16_ONEGEN_DELAY) __builtin_amdgcn_s_sleep(32);
    }
    if (a.trace && threadIdx.x == 0) a.trace[(unsigned long long)blockIdx.x * NP_TRACE_WORDS + 1] = __builtin_readcyclecounter();

    // row-indexed arrays are addressed as (uniform 64-bit base in SGPRs) + (32-bit per-lane offset): ld < 2^30 is checked on the
    // host, so the byte offset of a row fits 32 bits and no per-access 64-bit VALU address arithmetic is left
    const unsigned r32 = (unsigned)ic, o4 = r32 * 4u, o8 = r32 * 8u;
    float s[12], u[4], tgt[3];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = at_off(a.s + k * a.ld, o4);
#pragma unroll
    for (int k = 0; k < 4; k++) u[k] = at_off(a.u + k * a.ld, o4);
#pragma unroll
    for (int k = 0; k < 3; k++) tgt[k] = at_off(a.tgt + k * a.ld, o4);
    long long sc = at_off(a.step_count, o8);
    const unsigned fl_in = at_off(a.fin0, r32) | at_off(a.fin1, r32) | at_off(a.fin2, r32);
    if constexpr (SHARED) {
        // Observation noise first (round 3): it depends on (seed, call index, row) only, so this wave's share — its Philox block(s) and
        // their Box-Muller pairs, ~150-300 instructions — runs while the state loads above are in flight instead of between the two
        // net phases; the values wait in their own LDS columns for the wave that finishes the observation (same values, same fma).
        if (!a.noise && a.cfg.noise_scale != 0.0f) {
        // this wave's Philox block of the row -> its two or three Box-Muller pairs -> LDS (published by the barrier that opens the
        // Overload evaluation; wave 0 adds them to the observation afterwards: same values, same fma as add_rng_noise)
        const uint64_t call_idx2 = a.call_idx + (a.call_idx_base ? *a.call_idx_base : 0ull);
#pragma unroll
        for (int q = 0; q < (WPT == WPT_LAT2 ? 2 : 1); q++) {  // two waves per tile: blocks {0, 3} and {1, 2} (5 and 6 pairs)
        // eight waves: 0..3 are computing the state's trigonometry meanwhile
        const int nb = WPT == 8 ? part - 4 : WPT == WPT_LAT2 ? (q == 0 ? part : 3 - part) : part;
        if (nb >= 0) {
        uint32_t blk[4], k1[3], k2[3];
        rng_block(a.seed, call_idx2, a.row0 + ic, 2u + (uint32_t)nb, blk);
        noise_block_indices(blk, k1, k2);
        float *nz = coef + NOISE_COL0 * TILE;
        const float scale = a.cfg.noise_scale;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            if (j < 2 || nb < 3) {
                const int pair = j < 2 ? 2 * nb + j : 8 + nb;  // wave-uniform
                float rs, cs, sn;
                noise_pair(k1[j], k2[j], scale, rs, cs, sn);
                nz[(3 * pair) * TILE] = rs;
                nz[(3 * pair + 1) * TILE] = cs;
                nz[(3 * pair + 2) * TILE] = sn;
            }
        }
        }
        }
        }
    }
    const bool flagged = fl_in != 0;

    // a.inner == 2 (INNER kernels only): F16Model.update(action) on its own (F16_model.py:51-67) — controls and state advance for EVERY row,
    // nothing else of env.step happens: no hold, no step counter, flags handed through, no reward
    const bool update_only = INNER && a.inner == 2;  // wave-uniform
    const bool frozen = INNER && flagged && !update_only;  // planning_env.py:162-166: s[reset] = recent_s[reset]
    const bool tmo_prev = INNER && at_off(a.fin2, r32) != 0;
    // ---- self.reset(): re-initialise rows flagged by the previous step (env_base.py:83-95) ----
    if (flagged && !INNER) {
        float ru[5];
        if (a.rand_u) {
#pragma unroll
            for (int k = 0; k < 5; k++) ru[k] = a.rand_u[ic * 5 + k];
        } else {
            uint32_t w0[4], w1[4];
            rng_block(a.seed, call_idx, a.row0 + ic, 0, w0);
            rng_block(a.seed, call_idx, a.row0 + ic, 1, w1);
#pragma unroll
            for (int k = 0; k < 4; k++) ru[k] = (float)(w0[k] >> 8) * 5.9604644775390625e-08f;
            ru[4] = (float)(w1[0] >> 8) * 5.9604644775390625e-08f;
        }
        reset_row<TASK>(a.cfg, ru, s, u, tgt, sc);
    }

    // cache tile of this workgroup: NUM_CACHE_ROWS rows of 64 floats, contiguous
    // cache layout (private to the library, the same for every kernel variant): [row / 64][NUM_CACHE_ROWS][row % 64] — rows 0..13 the
    // force-side alpha/beta-only coefficients at the row's CURRENT state, rows 14..23 that state's trigonometry (np_nets.h)
    float *cache_blk = a.cache ? a.cache + ((i >> 6) * NUM_CACHE_ROWS) * CACHE_TILE + (i & (CACHE_TILE - 1)) : nullptr;
    constexpr bool TRIG_CACHED = STEP && CACHED && NUM_CACHED_TRIG > 0 && !SHARED;   // the integrator evaluation takes the state's trigonometry from the cache
    StateScalars sc_old;
    // the actions are requested before the cache check below: its branch ends the basic block, and a load issued behind it would start
    // only after every earlier load has returned (measured: -12 % at N = 1e6 with the four action loads behind the branch)
    float act_raw[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (STEP) {
#pragma unroll
        for (int k = 0; k < 4; k++) act_raw[k] = a.action[ic * a.act_stride + k];
    }
    if (STEP && CACHED) {  // coefficient columns <- cache (a reset aircraft sits at alpha = beta = 0)
        float key_af = cache_blk[CACHE_KEY0 * CACHE_TILE], key_bf = cache_blk[(CACHE_KEY0 + 1) * CACHE_TILE];
#pragma unroll
        for (int k = 0; k < NUM_CACHED; k++) {
            coef[cached_slot(k) * TILE] = cache_blk[k * CACHE_TILE];
        }
        // the two key loads stay up there, in front of the fourteen: left to itself the compiler sinks them into the (exec-masked) block
        // of their only use, behind the waits of everything above — a whole memory round trip on the critical path (measured: +5 %)
        asm volatile("" : "+v"(key_af), "+v"(key_bf));
        const unsigned key_a = __float_as_uint(key_af), key_b = __float_as_uint(key_bf);
        // Do the coefficients belong to THIS state?  The keys are the (alpha, beta) they were evaluated at (np_nets.h).  A mismatch on
        // any lane — the caller edited the state behind the library's back — sends the wave (latency family: every wave of the tile, they
        // hold the same rows) through the force-side evaluation the previous step would have left at the state at hand: the same nets, inputs
        // and statements as the Overload evaluation that fills the cache in every step, so the lanes that were fine get their cached values
        // again, bit for bit.  Cold path: two loads, two compares and a branch on the hot one.
#ifndef NPF16_CACHE_CHECK
#define NPF16_CACHE_CHECK 1   // 0: timing experiments only (tools/microbench): the round-3 behaviour, trust the caller's cache_valid
#endif
        if (NPF16_CACHE_CHECK) {
            // (lanes beyond the batch shadow its last row but would read the keys of rows that do not exist)
            const bool stale = valid && !(flagged && !INNER) && (key_a != __float_as_uint(s[7]) || key_b != __float_as_uint(s[8]));
            if (__ballot(stale) != 0ull) {
#if NPF16_CACHE_CHECK == 2   // timing experiment: the check with a cold path of one instruction
                __builtin_trap();
#else
                float xd_[12];
                if constexpr (SHARED) {
                    StateScalars scx;
                    nlplant<false, AB_FORCE, TILE, WPT, true, 1>(a.wt, airframe_via(ap), s, u, scx, coef, tables, xd_, pw);
                } else {   // one wave on its own (the pair variant's two waves hold different rows and decide for themselves)
                    StateScalars scx;
                    trig_of(s, scx.tr, scx.tt);
                    scx.spsi = scx.cpsi = 0.0f;
                    nlplant<false, AB_FORCE, TILE, 1, false, 0>(a.wt, airframe_via(ap), s, u, scx, coef, tables, xd_, 0);
                }
#endif
            }
        }
        if (flagged && !INNER) {  // a re-initialised aircraft: overwrite its column (LDS writes of the few flagged lanes instead of 14 selects for all)
#pragma unroll
            for (int k = 0; k < NUM_CACHED; k++) coef[cached_slot(k) * TILE] = a.reset_coef[k];
        }
        if constexpr (TRIG_CACHED) {
            float tv[10];  // NUM_CACHED_TRIG values when the switch is on
#pragma unroll
            for (int k = 0; k < NUM_CACHED_TRIG; k++) tv[k] = cache_blk[(NUM_CACHED + k) * CACHE_TILE];
            if (flagged && !INNER) {  // a re-initialised aircraft: every angle is 0 (F16_model.py:33-45); its altitude was just drawn
                tv[0] = tv[2] = tv[4] = tv[6] = tv[8] = 0.0f;
                tv[1] = tv[3] = tv[5] = tv[7] = 1.0f;
                tv[9] = atmos_pow(ap->cfg.af, s[2]);
            }
            sc_old.tr.sa = tv[0]; sc_old.tr.ca = tv[1]; sc_old.tr.sb = tv[2]; sc_old.tr.cb = tv[3];
            sc_old.tr.st = tv[4]; sc_old.tr.ct = tv[5]; sc_old.tr.sphi = tv[6]; sc_old.tr.cphi = tv[7];
            sc_old.tt = tv[8];
            sc_old.powv = tv[9];
        }
    }
    if (!STEP && a.term_reasons && valid && part == 0) a.term_reasons[i] = 0;  // reset(): every flag cleared, no condition evaluated
    if (!STEP && a.cache && flagged && valid && !INNER && part == 0) {  // reset(): keep the cache consistent for re-initialised rows
#pragma unroll
        for (int k = 0; k < NUM_CACHED; k++) cache_blk[k * CACHE_TILE] = a.reset_coef[k];
        if constexpr (NUM_CACHED_TRIG > 0) {
#pragma unroll
            for (int k = 0; k < 9; k++) cache_blk[(NUM_CACHED + k) * CACHE_TILE] = (k & 1) && k < 8 ? 1.0f : 0.0f;   // sin 0, cos 0 x 4, tan 0
            cache_blk[(NUM_CACHED + 9) * CACHE_TILE] = atmos_pow(ap->cfg.af, s[2]);
        }
        cache_blk[CACHE_KEY0 * CACHE_TILE] = s[7];
        cache_blk[(CACHE_KEY0 + 1) * CACHE_TILE] = s[8];
    }

    if (STEP) {
        // ---- F16Model.update (F16_model.py:51-67) ----
        float act[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float v = act_raw[k];
            v = v < -1.0f ? -1.0f : v;  // torch.clamp(action, -1, 1); NaN stays NaN
            v = v > 1.0f ? 1.0f : v;
            act[k] = v;
        }
        control_lag(ap->cfg.af, act, u);
        if (SOLVER == 0) {  // euler: y1 = y0 + dt*f(y0)
            float k1[12];
            NP_LT(1);
            if constexpr (SHARED) {
                StateScalars sc0;
                nlplant<true, (CACHED ? AB_REST : AB_ALL), TILE, WPT, true, 0>(a.wt, airframe_via(ap), s, u, sc0, coef, tables, k1, pw);
                NP_LT(2);
            } else if constexpr (TRIG_CACHED) {  // the state's trigonometry comes from the cache; only the heading's is evaluated here
                np_sincos(s[5], sc_old.spsi, sc_old.cpsi);
                nlplant<true, AB_REST, TILE, WPT, false, 0, true>(a.wt, airframe_via(ap), s, u, sc_old, coef, tables, k1, pw);
                NP_LT(2);
            } else {
                xdot_full<(CACHED ? AB_REST : AB_ALL), TILE, WPT>(a.wt, airframe_via(ap), s, u, coef, tables, k1, pw);
                NP_LT(2);
            }
            NP_REREAD_ARGS(ap);
            const float dt = ap->cfg.dt;
#pragma unroll
            for (int k = 0; k < 12; k++) s[k] = frozen ? s[k] : s[k] + dt * k1[k];
        } else {  // torchdiffeq 0.2.3 rk4_alt_step_func (3/8 rule)
            const float dt = a.cfg.dt;
            const float third = (float)(1.0 / 3.0);
            float y[12], k1[12], k2[12], k3[12];
#pragma unroll
            for (int k = 0; k < 12; k++) y[k] = s[k];
#pragma nounroll
            for (int stage = 0; stage < 4; stage++) {
                float kk[12];
                // weights pointer and numerics switch re-read per stage: nothing scalar but `ap` stays live across the asm phases (they
                // own s4-s101; the statement's record pointer and `ap` fill s0-s3)
                NP_REREAD_ARGS(ap);
                const AeroWeights wts = {ap->wt.kblob, ap->wt.kblob_dual, ap->wt.pwl, ap->wt.pwl_unnorm};
                const bool tbs = ap->cfg.aero_1d_tables != 0;
                if (CACHED && stage == 0) xdot_full<AB_REST, TILE, WPT>(wts, airframe_via(ap), y, u, coef, tbs, kk, pw);  // y == s: cached coefficients apply
                else xdot_full<AB_ALL, TILE, WPT>(wts, airframe_via(ap), y, u, coef, tbs, kk, pw);
                if (stage == 0) {
#pragma unroll
                    for (int k = 0; k < 12; k++) {
                        k1[k] = kk[k];
                        y[k] = s[k] + (dt * k1[k]) * third;
                    }
                } else if (stage == 1) {
#pragma unroll
                    for (int k = 0; k < 12; k++) {
                        k2[k] = kk[k];
                        y[k] = s[k] + dt * (k2[k] - k1[k] * third);
                    }
                } else if (stage == 2) {
#pragma unroll
                    for (int k = 0; k < 12; k++) {
                        k3[k] = kk[k];
                        y[k] = s[k] + dt * ((k1[k] - k2[k]) + k3[k]);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 12; k++) y[k] = s[k] + (((k1[k] + 3.0f * (k2[k] + k3[k])) + kk[k]) * dt) * 0.125f;
                }
            }
#pragma unroll
            for (int k = 0; k < 12; k++) s[k] = frozen ? s[k] : y[k];
            NP_REREAD_ARGS(ap);
        }
        sc += (INNER && ap->inner == 2) ? 0 : 1;  // env_base.py:102
    }

    // ---- observation at the new state (task.get_obs) ----
    // Every variant but the latency family builds it AFTER the Overload evaluation (below): its 22 values would otherwise stay live
    // across that asm phase — 20 of them were what the three-waves-per-SIMD build of the pair variant parked in scratch (round 2).
    Trig tr;
    float o[22];
    StateScalars sc1;  // SHARED: filled by the Overload evaluation below
    const bool gen_noise = !ap->noise && ap->cfg.noise_scale != 0.0f;
    float tt_new = 0.0f, pow_new = 0.0f;   // tan(theta) and the atmosphere power of the state this step reaches (cache rows)
    if constexpr (!SHARED) {
        trig_of(s, tr, tt_new);
        if (STEP) {
            // The Overload phase below is an asm statement that owns v70-v157.  Left alone, the compiler SINKS whatever the code before
            // that statement does not need past it — the moment equations of the integrator evaluation (the new P, Q, R are first read
            // by the force build-up after the statement: ~30 raw coefficients stayed live instead of 3 states) and the tails of the
            // fp64 sine / cosine sequences (their 64-bit intermediates stayed live instead of 8 floats): 93 registers live across the
            // statement, 20 of them in scratch in the three-waves-per-SIMD build.  Pin the new state and its trigonometry here.
#ifndef NPF16_PIN_MASK
#define NPF16_PIN_MASK 3
#endif
            if ((NPF16_PIN_MASK & 1) && (WPT == 2 || (NPF16_PIN_MASK & 4))) {
#pragma unroll
                for (int k = 0; k < 12; k++) asm volatile("" : "+v"(s[k]));
            }
            if ((NPF16_PIN_MASK & 2) && (WPT == 2 || (NPF16_PIN_MASK & 4)))
                asm volatile("" : "+v"(tr.sa), "+v"(tr.ca), "+v"(tr.sb), "+v"(tr.cb), "+v"(tr.st), "+v"(tr.ct), "+v"(tr.sphi), "+v"(tr.cphi));
        }
    }

    bool done = false, bad = false;
    float reward = 0.0f;
    if (STEP) {
        // Overload needs xdot[6..8] at the NEW (s,u) (overload.py:37-42): the 14 force-side alpha/beta-only
        // nets (kept for the next step's integrator -> cache) plus the force-side Cx, Cz
        float xd[12];
#if defined(NPF16_EXP) && (NPF16_EXP & 2)  // timing experiment only: no Overload evaluation
        for (int k = 0; k < 12; k++) xd[k] = s[k];
#else
        {
            const AeroWeights wt2 = {ap->wt.kblob, ap->wt.kblob_dual, ap->wt.pwl, ap->wt.pwl_unnorm};
            if constexpr (SHARED) {
                NP_LT(3);
                nlplant<false, AB_FORCE, TILE, WPT, true, 1>(wt2, airframe_via(ap), s, u, sc1, coef, ap->cfg.aero_1d_tables != 0, xd, pw);
                NP_LT(4);
                tr = sc1.tr;
                tt_new = sc1.tt;
                pow_new = sc1.powv;
            } else {
                NP_LT(3);
                StateScalars scn;
                scn.tr = tr;
                scn.tt = scn.spsi = scn.cpsi = 0.0f;
                nlplant<false, AB_FORCE, TILE, WPT, false, 0>(wt2, airframe_via(ap), s, u, scn, coef, ap->cfg.aero_1d_tables != 0, xd, pw);
                pow_new = scn.powv;
                NP_LT(4);
            }
        }
#endif
        NP_REREAD_ARGS(ap);
        if (!SHARED || part == STATE_WAVE) {
            float acc3[3];
            body_acceleration(s, tr, xd, acc3);
            // inner iterations: the env flags keep accumulating (env_base.py:72-74) and the event reward sees the sum
            const bool done_prev = INNER && at_off(ap->fin0, r32) != 0, bad_prev = INNER && at_off(ap->fin1, r32) != 0;
            unsigned reasons = 0;
            float reward_task = 0.0f;
            done_and_reward<TASK>(ap->cfg, s, tgt, acc3, sc, done_prev, bad_prev, done, bad, reward, reasons, reward_task);
            if (INNER && ap->inner == 2) {  // F16Model.update alone: the env's flags are not this call's business
                done = done_prev;
                bad = bad_prev;
            }
            if (ap->reward_task && valid && part == STATE_WAVE) ap->reward_task[i] = reward_task;  // wave-uniform pointer test
            if (ap->term_reasons && valid && part == STATE_WAVE) {  // wave-uniform pointer test
                // inner iterations of PlanningEnv.step: the bits accumulate like the flags they explain (the reset launch that opens
                // the macro-step cleared them), so a row that tripped a condition at inner step 3 still shows it after step 50
                if (INNER) reasons |= ap->term_reasons[i];
                ap->term_reasons[i] = (unsigned char)reasons;
            }
            if (ap->term_counters) {
                // the reference prints torch.sum(mask) per termination condition and step (a host sync each); here: one wave
                // ballot per condition, population count, ONE atomic per wave for a condition that fired at all
                const bool counted = valid && part == STATE_WAVE;
#pragma unroll
                for (int k = 0; k < NP_NUM_TERM_COUNTERS; k++) {
                    const unsigned long long m = __ballot(counted && ((reasons >> k) & 1u));
                    if (m != 0 && (threadIdx.x & 63) == 0) atomicAdd(ap->term_counters + k, (unsigned)__popcll(m));
                }
            }
        }
    }
    if constexpr (SHARED) {
        if (part == 0) {  // the wave that owns the observation: base values, then the noise the four waves prepared
            observe<TASK, true>(ap->cfg, s, u, tgt, tr, o, sc1.powv);
            if (ap->noise) {
#pragma unroll
                for (int k = 0; k < 22; k++) o[k] = o[k] + ap->noise[ic * 22 + k] * ap->cfg.noise_scale;
            } else if (gen_noise) {
                const float *nz = coef + NOISE_COL0 * TILE;
#pragma unroll
                for (int pair = 0; pair < 11; pair++) {
                    const float rs = nz[(3 * pair) * TILE], cs = nz[(3 * pair + 1) * TILE], sn = nz[(3 * pair + 2) * TILE];
                    o[2 * pair] = fmaf(rs, cs, o[2 * pair]);
                    o[2 * pair + 1] = fmaf(rs, sn, o[2 * pair + 1]);
                }
            }
        }
    }

    if (valid && part == STATE_WAVE) {
        // re-derive the store addresses from the row index here: without the empty asm the compiler keeps the ~25 64-bit
        // load addresses of the top of the kernel alive across both MLP phases (and spills some of them to scratch)
        unsigned iw = (unsigned)i;
        asm volatile("" : "+v"(iw));
        const unsigned w4 = iw * 4u;
#pragma unroll
        for (int k = 0; k < 12; k++) at_off(ap->s + k * ap->ld, w4) = s[k];
#pragma unroll
        for (int k = 0; k < 4; k++) at_off(ap->u + k * ap->ld, w4) = u[k];
        if (flagged && !INNER) {  // the targets change only when the row is re-initialised (task.reset)
#pragma unroll
            for (int k = 0; k < 3; k++) at_off(ap->tgt + k * ap->ld, w4) = tgt[k];
        }
        at_off(ap->step_count, iw * 8u) = sc;
        at_off(ap->fout0, iw) = done ? 1 : 0;
        at_off(ap->fout1, iw) = bad ? 1 : 0;
        at_off(ap->fout2, iw) = tmo_prev ? 1 : 0;
        if (STEP && (!INNER || ap->reward)) at_off(ap->reward, w4) = reward;  // (no reward buffer: F16Model.update alone)
        if (STEP && ap->cache) {
            float *cache_w = ap->cache + ((long long)(iw >> 6) * NUM_CACHE_ROWS) * CACHE_TILE + (iw & (CACHE_TILE - 1));
#pragma unroll
            for (int k = 0; k < NUM_CACHED; k++) cache_w[k * CACHE_TILE] = coef[cached_slot(k) * TILE];
            if constexpr (NUM_CACHED_TRIG > 0) {
                const float tv[10] = {tr.sa, tr.ca, tr.sb, tr.cb, tr.st, tr.ct, tr.sphi, tr.cphi, tt_new, pow_new};
#pragma unroll
                for (int k = 0; k < NUM_CACHED_TRIG; k++) cache_w[(NUM_CACHED + k) * CACHE_TILE] = tv[k];
            }
            cache_w[CACHE_KEY0 * CACHE_TILE] = s[7];   // the (alpha, beta) these coefficients belong to
            cache_w[(CACHE_KEY0 + 1) * CACHE_TILE] = s[8];
        }
    }

    if constexpr (!SHARED) {
        NP_REREAD_ARGS(ap);
        if (!INNER || ap->obs) {  // an intermediate inner iteration of PlanningEnv.step may not want the task observation at all
        observe<TASK>(ap->cfg, s, u, tgt, tr, o);
        if (ap->noise) {  // obs + randn_like(obs) * noise_scale
#pragma unroll
            for (int k = 0; k < 22; k++) o[k] = o[k] + ap->noise[ic * 22 + k] * ap->cfg.noise_scale;
        } else if (gen_noise) {
#if !(defined(NPF16_EXP) && (NPF16_EXP & 1))  // timing experiment only: no observation noise
            const uint64_t call_idx2 = ap->call_idx + (ap->call_idx_base ? *ap->call_idx_base : 0ull);
            add_rng_noise(ap->seed, call_idx2, ap->row0 + ic, ap->cfg.noise_scale, o);
#endif
        }
        }
    }

    NP_LT(5);
    // ---- [n][22] observation rows: transpose through LDS, store coalesced ----
    auto store_rows22 = [&](float *out_base, const float (&o)[22]) {
        __syncthreads();  // every lane is done with its coefficient column (or the previous tile) before the tile overwrites it
        const long long rows = (ap->n - i0) < TILE ? (ap->n - i0) : TILE;
        float *dst = out_base + i0 * 22;
        constexpr int THREADS = TILE * lat_waves(WPT);
        if (rows == TILE && ((uintptr_t)dst & 15) == 0) {  // workgroup-uniform: a full tile and a 16-byte aligned destination
            // unpadded rows (pitch 22 floats): 11 ds_write_b64 per lane, then the tile leaves as 16-byte vectors — 6 (2)
            // ds_read_b128 + global_store_dwordx4 per thread instead of 22 dword pairs; the LDS bank conflicts of the unpadded
            // pitch cost LDS cycles, which this VALU-bound kernel has to spare, and no VALU index arithmetic is left
            if (part == 0) {
                float2 *row = reinterpret_cast<float2 *>(obs_tile + t * 22);
#pragma unroll
                for (int k = 0; k < 11; k++) row[k] = make_float2(o[2 * k], o[2 * k + 1]);
            }
            __syncthreads();
            constexpr int VECS = TILE * 22 / 4;
            static_assert((TILE * 22) % 4 == 0, "a tile is a whole number of 16-byte vectors");
            const float4 *src4 = reinterpret_cast<const float4 *>(obs_tile);
            float4 *dst4 = reinterpret_cast<float4 *>(dst);
#pragma unroll
            for (int it = 0; it < (VECS + THREADS - 1) / THREADS; it++) {
                const int L = it * THREADS + (int)threadIdx.x;
                if ((it + 1) * THREADS <= VECS || L < VECS) dst4[L] = src4[L];
            }
        } else {  // the last, partial tile of a batch (or an unaligned caller buffer): dword by dword through the padded pitch
            if (part == 0) {
#pragma unroll
                for (int k = 0; k < 22; k++) obs_tile[t * OBS_LD + k] = o[k];
            }
            __syncthreads();
            const int total = (int)rows * 22;
#pragma nounroll
            for (int it = 0; it < (22 * TILE + THREADS - 1) / THREADS; it++) {
                const int L = it * THREADS + (int)threadIdx.x;
                if (L < total) {
                    // row r = L / 22 of the tile, padded pitch 23: element r * 23 + (L - 22 r) = L + r; L * 2979 >> 16 == L / 22
                    // for every L < 22 * 256 (24-bit product: one v_mul_u32_u24)
                    static_assert(OBS_LD == 23 && TILE <= 256, "index arithmetic of the observation transpose");
                    const unsigned r = ((unsigned)L * 2979u) >> 16;
                    dst[L] = obs_tile[(unsigned)L + r];
                }
            }
        }
    };
    if (ap->obs) store_rows22(ap->obs, o);
    if constexpr (INNER) {
        // PlanningEnv.low_level_obs (planning_env.py:60-142) of the state just reached, for the controller's next call: the
        // ControlTask-style observation against the caller's targets, no noise — the arithmetic of f16_lowlevel_obs_kernel on
        // the trigonometry this step already has
        if (ap->ll_obs) {  // wave-uniform
            float o2[22];
            if (part == 0) {
                float t3[3];
#pragma unroll
                for (int k = 0; k < 3; k++) t3[k] = at_off(ap->ll_tgt + k * ap->ld, o4);
                if constexpr (SHARED) observe<1, true>(ap->cfg, s, u, t3, tr, o2, sc1.powv);
                else observe<1>(ap->cfg, s, u, t3, tr, o2);
            }
            store_rows22(ap->ll_obs, o2);
        }
    }
#ifdef NPF16_LAT_TRACE
    NP_LT(6);
    if (ap->trace && (threadIdx.x & 63) == 0) {  // 8 words per wave (the caller sizes the buffer)
        constexpr int NW = TILE * lat_waves(WPT) / 64;
        unsigned long long *rec = ap->trace + ((unsigned long long)blockIdx.x * NW + threadIdx.x / 64) * 8;
        for (int k = 0; k < 7; k++) rec[k] = lt[k];
    }
    if (0) {
#else
    if (ap->trace && threadIdx.x == 0) {
#endif
        unsigned long long *rec = ap->trace + (unsigned long long)blockIdx.x * NP_TRACE_WORDS;
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        rec[2] = __builtin_readcyclecounter();
        rec[4] = wall_clock64();
        rec[5] = ((unsigned long long)xcc << 32) | hw;
    }
}

}  // namespace npf16
