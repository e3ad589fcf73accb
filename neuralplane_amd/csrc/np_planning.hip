// np_planning.hip — PlanningEnv's 50 low-level iterations as ONE persistent gfx950 kernel.
//
// Reference: envs/planning_env.py:153-176 — per iteration: ego_actions = controller(low_level_obs, rnn_states, masks);
// model.update(ego_actions); step_count += 1; terminations / reward; the next low_level_obs.  np_planning_inner_loop
// (np_f16_kernels.hip) enqueues that as 2 x 50 launches (np_actor_forward + np_f16_step with inner_step); here a workgroup owns a
// tile of 32 aircraft — one 32-row tile of the controller's MFMA kernel (np_actor.h::actor32_body) — and runs
//     controller call  ->  barrier  ->  inner FDM step (np_f16_device.h, the latency variant's device code on the tile's rows)  ->  barrier
// for every iteration inside one launch.  The tile's data never leaves the CU between the two halves of an iteration nor between
// iterations: the recurrent state stays in registers (16 per thread), the low-level observation, the actions, the aircraft state,
// counters, flags and the 14 cross-step aero coefficients stay in LDS (the "tile context" below).  Global memory is touched when a
// tile is IMPORTED (before its first iteration on this workgroup) and EXPORTED (after its last).
// The arithmetic is the launch-by-launch path's, operation by operation (same device functions, same generated statements, same
// order): results are bit-identical (tests/test_gpu_actor.py).
//
// Two schedules:
//   static : grid = tiles, workgroup b runs all iterations of tile b: one import, one export;
//   queue  : `grid` resident workgroups pull (tile, iteration) items, id = iteration * tiles + tile, from an atomic counter; an item
//            waits until its tile's previous iteration is published in queue[1 + tile].  A tile then moves between CUs (and XCDs,
//            whose L2s are not coherent with each other), so every item imports and exports, and these accesses are agent-scope
//            relaxed atomics (global_load / global_store ... sc1: they bypass the non-coherent cache levels) ordered by
//            s_waitcnt vmcnt(0) + the workgroup barrier before the flag store — a full agent-scope release / acquire pair per item
//            (buffer_wbl2 / buffer_inv) measured ~50 us per item, more than the item itself.  The lowest outstanding id never
//            waits on anything unfinished, so the schedule cannot deadlock as long as the grid is resident (the launcher sizes it by
//            the occupancy query).  313 tiles (n = 1e4) on 512 slots of two 4-wave workgroups per CU: whichever slot is free takes
//            the next item, and the two workgroups of a CU hide each other's barriers and load latencies.
//   guests : (np_planning_loop.mode = guests) every workgroup owns a tile and hosts one block of a guest tile's iterations; see the
//            schedule loop below.
// Forward progress of the waits (thread 0 polling a tile's progress word): a queue item waits for an item with a LOWER id, which an
// already running workgroup took from the counter before; a guest block j waits for block j - 1, hosted by workgroup blockIdx.x - 1.
// With the whole grid resident (what the launcher sizes it for) that is all there is to it.  If it is not, the argument leans on the
// command processor dispatching workgroups in index order (what it does; not an architectural promise): whatever a workgroup waits for
// belongs to a workgroup that was dispatched before it and is running or done — also when the grid is NOT fully resident (another
// process or stream holds CUs): the chain of waits always ends at a workgroup that waits for nothing (block 0 / the lowest id).
// What residency buys is speed (nobody waits long), not safety; the sizing by the occupancy query is for that.
// Because that order is observed behaviour and not a promise, the wait is BOUNDED (round 5): a wait that outlasts PlanArgs::wait_ticks
// (2 s unless NP_PLANNING_WAIT_MS says otherwise) claims the launch's sticky error word, every workgroup that meets the word drains, the
// kernel ENDS, and np_planning_inner_loop reports NP_E_PLANNING_STALLED with the buffers restored (np_f16_kernels.hip) — a violated
// assumption is an error code and a launch-by-launch re-run, not a hung GPU.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "../../include/neuralplane_amd.h"
#include "np_planning.h"
#define NPACT_NO_KERNELS 1
#include "np_actor.h"
#include "np_actor_i8.h"

#ifndef NP_PLAN_WIN
#define NP_PLAN_WIN 1   // 0: the static schedule keeps the 22 moment-side nets in the front of an inner step (A/B)
#endif
#ifndef NP_PLAN_WIN_QUEUE
#define NP_PLAN_WIN_QUEUE 0   // 1: the coherent kernels (guest / queue schedules) too — measured level to slightly worse (A/B builds only)
#endif
#ifndef NP_PLAN_PIPE
#define NP_PLAN_PIPE 1  // 0: eight-wave tiles run the inner step sequentially like the four-wave ones (A/B)
#endif
#ifndef NP_PLAN_BACK_PRIO
#define NP_PLAN_BACK_PRIO 0  // > 0: s_setprio of waves 0..3 during a controller call, waves 4..7 (the back) at 0 — A/B only (no effect measured)
#endif
#ifndef NP_PLAN_TRACE
#define NP_PLAN_TRACE 0  // 1: the last workgroup stamps the shader clock at the phase boundaries of its last-but-one iteration (tools/microbench/planning_phases.py); never shipped
#endif

namespace npf16 {

typedef const PlanArgs __attribute__((address_space(4))) *PlanArgsC;
__device__ __forceinline__ auto np_cfg_of(PlanArgsC ap) -> decltype(&ap->k.cfg) { return &ap->k.cfg; }   // (AirframeVia: the env record is member `k`)

#if NP_PLAN_TRACE
__device__ unsigned long long np_plan_trace_buf[8 * 16];   // [wave][stamp]
#define NP_PSTAMP(k) do { if (blockIdx.x == gridDim.x - 1 && (tid & 63) == 0 && it == ap->iterations - 2) np_plan_trace_buf[(tid >> 6) * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define NP_PSTAMP(k) do { } while (0)
#endif

constexpr int PLAN_TILE = 64;  // lanes per wave = LDS column pitch of the FDM device code; lanes 32..63 shadow rows 0..31 (stores masked)
constexpr int PLAN_STATE_WAVE = 1;  // as the latency variant: wave 1 evaluates terminations / reward and owns the state
constexpr int PLAN_COLS = NUM_LDS_SLOTS + 2 * NUM_SHARED_SCALARS;  // coefficient columns + the two sets of shared state scalars
// tile context (floats): [row][22] low-level observation | [row][4] actions | [18][32] state rows: s 0..11, u 12..15, step_count (int
// bits), flag / reason bits (bit 0 done, 1 bad_done, 2 exceed_time_limit, 8..14 the termination reasons accumulated over the iterations),
// then what the iterations only read: the controller's mask and its three targets (a global load per iteration otherwise, its latency
// exposed on wave 0 right after a barrier)
constexpr int CTX_OBS = 0, CTX_ACT = CTX_OBS + PLAN_ROWS * 22, CTX_ST = CTX_ACT + PLAN_ROWS * 4, CTX_ROWS = 22, CTX_SC = 16, CTX_FL = 17, CTX_MK = 18, CTX_LT = 19;
constexpr int CTX_FLOATS = CTX_ST + CTX_ROWS * PLAN_ROWS;
// the same layout for a context of ROWS rows (32: one tile; 64: the two tiles of a dual workgroup, tile A = rows 0..31, tile B = 32..63)
template <int ROWS>
struct CtxL {
    static constexpr int OBS = 0, ACT = ROWS * 22, ST = ACT + ROWS * 4, FLOATS = ST + CTX_ROWS * ROWS;
};
static_assert(CtxL<PLAN_ROWS>::ST == CTX_ST && CtxL<PLAN_ROWS>::FLOATS == CTX_FLOATS, "context layout");
// dual workgroups (round 4): TWO 32-row tiles per eight-wave workgroup — their controller calls run in lock-step on waves 0..3 / 4..7 (same
// straight-line code, shared barriers), each with its own 36 KB of controller LDS, and ONE inner FDM step serves both: the FDM device code
// runs 64-lane waves, which a single 32-row tile fills only half (lanes 32..63 shadow rows 0..31).  Dynamic LDS (102 KB), one workgroup per CU.
constexpr int PARK_LDS_FLOATS = CtxL<PLAN_ROWS>::FLOATS + NUM_CACHED * PLAN_TILE + npact::BLK * 256;   // context + cached columns + recurrent state
constexpr int DUAL_LDS_FLOATS = 2 * npact::ACTOR32_LDS_FLOATS + PLAN_COLS * PLAN_TILE + CtxL<2 * PLAN_ROWS>::FLOATS;
constexpr int PLAN_LDS_FLOATS = npact::ACTOR32_LDS_FLOATS + PLAN_COLS * PLAN_TILE + CTX_FLOATS;
static_assert(PLAN_LDS_FLOATS * sizeof(float) <= 65536, "static LDS of the persistent kernel");
static_assert(33 * PLAN_TILE <= npact::ACTOR32_HEAD_W && PLAN_ROWS * OBS_LD <= npact::ACTOR32_HEAD_W,
              "noise columns / observation tile borrow the controller's LDS below its staged head weights");

// global loads / stores of tile data: plain, or (queue schedule) agent-scope relaxed atomics = `sc1` accesses that bypass the cache
// levels which are not coherent between CUs / XCDs
template <bool COH, class T>
__device__ __forceinline__ T gld(const T *p) {
    if constexpr (COH) return __hip_atomic_load(const_cast<T *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool COH, class T>
__device__ __forceinline__ void gst(T *p, T v) {
    if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// tile -> LDS context + coefficient columns (waves 0 / STATE_WAVE); the recurrent state of (row, block) -> hm, masked (gru.py:26)
// The recurrent state of a tile, [32 rows][128] floats = 16 KB contiguous in global memory, moved coherently (queue / guest schedules):
// thread (row, block) owns 64 B of a row, so per-thread sc1 accesses put 64 lanes on 64 different lines — and, bypassing the caches, fetch
// or write each line sixteen times over, 4 bytes at a time (measured: ~30 us per export / import pair).  Instead every wave of the workgroup moves
// 8-byte pieces that are consecutive across its lanes (512 B per instruction) between global memory and a stage in the controller's idle
// bufA, and the owners read / write the stage.  Stage layout: 8-byte piece c (0..63) of row r at float offset 128 r + 2 (c ^ 2 r): the xor
// spreads a block's 32 rows over the banks.
__device__ __forceinline__ int h_stage_off(int row, int piece) { return 128 * row + 2 * ((piece ^ (2 * row)) & 63); }

// which 16 features of its row a controller thread (row, blk = 2 wave + lane half) holds, as float4 / 8-byte piece indices inside the row:
// fp32 numerics (np_actor.h): features 16 blk .. 16 blk + 15; block fixed point (np_actor_i8.h, accumulator layout): 32 wave + 8 g + 4 half + t
template <bool I8>
__device__ __forceinline__ int h_float4(int blk, int g) { return I8 ? 8 * (blk >> 1) + (blk & 1) + 2 * g : 4 * blk + g; }
template <bool I8>
__device__ __forceinline__ int h_piece(int blk, int j) { return I8 ? 16 * (blk >> 1) + 2 * (blk & 1) + 4 * (j >> 1) + (j & 1) : 8 * blk + j; }

template <int W>
__device__ __forceinline__ void h_global_to_stage(const float *src, long long i0, long long n, float *stage, unsigned tid) {
#pragma unroll
    for (int q = 0; q < 2048 / (64 * W); q++) {
        const int idx = q * 64 * W + (int)tid, row = idx >> 6, piece = idx & 63;
        const long long i = i0 + row < n ? i0 + row : n - 1;
        const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(src + i * npact::HID) + piece, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *reinterpret_cast<unsigned long long *>(stage + h_stage_off(row, piece)) = v;
    }
}
template <int W>
__device__ __forceinline__ void h_stage_to_global(float *dst, long long i0, long long n, const float *stage, unsigned tid) {
#pragma unroll
    for (int q = 0; q < 2048 / (64 * W); q++) {
        const int idx = q * 64 * W + (int)tid, row = idx >> 6, piece = idx & 63;
        if (i0 + row < n)
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(dst + (i0 + row) * npact::HID) + piece,
                               *reinterpret_cast<const unsigned long long *>(stage + h_stage_off(row, piece)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int W, bool COH, int ROWS = PLAN_ROWS, bool I8 = false>
__device__ __forceinline__ void plan_import(PlanArgsC &ap, float *lds_fdm, float *lds_act, float *ctx, long long i0, int it, unsigned tid, float (&hm)[npact::BLK],
                                            unsigned *stale_s) {
    using CX = CtxL<ROWS>;
    NP_REREAD_ARGS(ap);
    const PlanArgsC a = ap;
    const int t = (int)(tid % PLAN_TILE), r = t & (ROWS - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(tid / PLAN_TILE));
    const long long n = a->k.n;
    const long long i = i0 + r;
    const long long ic = i < n ? i : n - 1;   // rows beyond the batch shadow its last row; nothing of them is exported
    if constexpr (COH) h_global_to_stage<W>(a->rnn[it & 1], i0, n, lds_act, tid);   // the owners read the stage after the caller's barrier
    if (!COH && (wave < 4 || ROWS == 2 * PLAN_ROWS)) {   // the controller's waves (dual: waves 4..7 hold tile B): this thread's 16 features of its row
        const int blk = 2 * (wave & 3) + (t >> 5);
        const long long ih = i0 + (ROWS == 2 * PLAN_ROWS && wave >= 4 ? PLAN_ROWS : 0) + (t & (PLAN_ROWS - 1));
        const long long ihc = ih < n ? ih : n - 1;
        const float mk = a->masks[ihc];
        const float *hp = a->rnn[it & 1] + ihc * npact::HID;
        {
#pragma unroll
            for (int j = 0; j < npact::BLK / 4; j++) {
                const float4 q = reinterpret_cast<const float4 *>(hp)[h_float4<I8>(blk, j)];
                hm[4 * j] = q.x * mk;
                hm[4 * j + 1] = q.y * mk;
                hm[4 * j + 2] = q.z * mk;
                hm[4 * j + 3] = q.w * mk;
            }
        }
    }
    if (wave == 0) {  // low-level observation rows [ROWS][22]: 704 dwords, 11 per lane (dual: twice that)
        const float *src = a->ll_obs[it & 1] + i0 * 22;
        const long long avail = (n - i0) * 22;
#pragma unroll
        for (int k = 0; k < ROWS * 22 / 64; k++) {
            const int L = k * 64 + t;
            ctx[CX::OBS + L] = gld<COH>(src + (L < avail ? L : 0));
        }
    }
    if (wave == PLAN_STATE_WAVE && t < ROWS) {
        float *st = ctx + CX::ST + r;
#pragma unroll
        for (int k = 0; k < 12; k++) st[k * ROWS] = gld<COH>(a->k.s + k * a->k.ld + ic);
#pragma unroll
        for (int k = 0; k < 4; k++) st[(12 + k) * ROWS] = gld<COH>(a->k.u + k * a->k.ld + ic);
        st[CTX_SC * ROWS] = __int_as_float((int)gld<COH>(a->k.step_count + ic));
        const uint8_t *fin = a->flags[it & 1];
        unsigned fl = (gld<COH>(fin + ic) ? 1u : 0u) | (gld<COH>(fin + n + ic) ? 2u : 0u) | (gld<COH>(fin + 2 * n + ic) ? 4u : 0u);
        if (a->k.term_reasons) fl |= (unsigned)gld<COH>(a->k.term_reasons + ic) << 8;
        st[CTX_FL * ROWS] = __uint_as_float(fl);
        st[CTX_MK * ROWS] = a->masks[ic];
#pragma unroll
        for (int k = 0; k < 3; k++) st[(CTX_LT + k) * ROWS] = a->k.ll_tgt[k * a->k.ld + ic];
    }
    if (wave == PLAN_STATE_WAVE) {  // coefficient columns <- the cross-step cache (layout [row / 64][NUM_CACHE_ROWS][row % 64]); all 64 lanes
        const float *cache_blk = a->k.cache + ((ic >> 6) * NUM_CACHE_ROWS) * CACHE_TILE + (ic & (CACHE_TILE - 1));
        float *coef = lds_fdm + t;
#pragma unroll
        for (int k = 0; k < NUM_CACHED; k++) coef[cached_slot(k) * PLAN_TILE] = gld<COH>(cache_blk + k * CACHE_TILE);
        // do they belong to the state at hand?  (np_nets.h: the keys are the (alpha, beta) they were evaluated at; the caller may have edited
        // the state since.)  The workgroup re-evaluates them if not (plan_fill_cache, first iteration of a macro-step only: afterwards the
        // cache travels with the state)
        const unsigned ka = __float_as_uint(gld<COH>(cache_blk + CACHE_KEY0 * CACHE_TILE)), kb = __float_as_uint(gld<COH>(cache_blk + (CACHE_KEY0 + 1) * CACHE_TILE));
        const bool stale = ka != __float_as_uint(gld<COH>(a->k.s + 7 * a->k.ld + ic)) || kb != __float_as_uint(gld<COH>(a->k.s + 8 * a->k.ld + ic));
        const unsigned long long any = __ballot(stale);
        if (t == 0) *stale_s = any != 0ull ? 1u : 0u;
    }
}

// One inner FDM step (np_f16_step with inner_step: f16_env_kernel<TASK, 0, true, true, 64, W, true>) of the tile in the context:
// no auto-reset, flagged rows frozen, flags accumulate; leaves the controller's next observation in the context unless `last`, writes the
// task observation and the reward if `last`; `do_export`: state, counters, flags, reason bits and cached coefficients -> global.
template <int TASK, int W, bool COH, int ROWS = PLAN_ROWS>
__device__ __forceinline__ void plan_fdm_step(PlanArgsC &ap, float *lds_fdm, float *lds_act, float *ctx, long long i0, int it, bool last, bool do_export,
                                              unsigned tid) {
    using CX = CtxL<ROWS>;
    // every scalar is (re-)read from the kernel-argument segment where it is used: read through the by-value parameter the compiler
    // hoists the loads out of the iteration loop and keeps ~60 SGPRs alive across the asm phases (parked in VGPR lanes, then scratch)
    NP_REREAD_ARGS(ap);
    const PlanArgsC a = ap;
    constexpr int TILE = PLAN_TILE;
    const int t = (int)(tid % TILE), r = t & (ROWS - 1);
    const int part = __builtin_amdgcn_readfirstlane((int)(tid / TILE));
    float *coef = lds_fdm + t;
    const long long n = a->k.n;
    const long long i = i0 + r;
    const bool valid = t < ROWS && i < n;
    const long long ic = i < n ? i : n - 1;
    const bool tables = false;  // the persistent kernel serves the MLP numerics (the launcher falls back otherwise)
    const bool want_obs = last && a->final_obs != nullptr;
    const unsigned o4 = (unsigned)ic * 4u;

    float s[12], u[4], tgt[3];
    const float *st = ctx + CX::ST + r;
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = st[k * ROWS];
#pragma unroll
    for (int k = 0; k < 4; k++) u[k] = st[(12 + k) * ROWS];
#pragma unroll
    for (int k = 0; k < 3; k++) tgt[k] = at_off(a->k.tgt + k * a->k.ld, o4);   // constant during the loop
    long long sc = (long long)__float_as_int(st[CTX_SC * ROWS]);
    const unsigned fl_in = __float_as_uint(st[CTX_FL * ROWS]);
    float *nz = lds_act + t;  // 33 noise columns, in the controller's (idle) LDS
    if (want_obs && !a->k.noise && a->k.cfg.noise_scale != 0.0f) {  // this wave's share of the observation noise (f16_env_kernel, SHARED)
        const int nb = W == 8 ? part - 4 : part;
        if (nb >= 0) {
            uint32_t blk[4], k1[3], k2[3];
            rng_block(a->k.seed, a->k.call_idx + (uint64_t)it + (a->k.call_idx_base ? *a->k.call_idx_base : 0ull), a->k.row0 + ic, 2u + (uint32_t)nb, blk);
            noise_block_indices(blk, k1, k2);
            const float scale = a->k.cfg.noise_scale;
#pragma unroll
            for (int j = 0; j < 3; j++) {
                if (j < 2 || nb < 3) {
                    const int pair = j < 2 ? 2 * nb + j : 8 + nb;
                    float rs, cs, sn;
                    noise_pair(k1[j], k2[j], scale, rs, cs, sn);
                    nz[(3 * pair) * TILE] = rs;
                    nz[(3 * pair + 1) * TILE] = cs;
                    nz[(3 * pair + 2) * TILE] = sn;
                }
            }
        }
    }
    const bool frozen = (fl_in & 7u) != 0;  // planning_env.py:162-166

    // ---- F16Model.update (F16_model.py:51-67) ----
    float act[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float v = ctx[CX::ACT + r * 4 + k];
        v = v < -1.0f ? -1.0f : v;
        v = v > 1.0f ? 1.0f : v;
        act[k] = v;
    }
    control_lag(ap->k.cfg.af, act, u);
    NP_PSTAMP(4);
    {
        float k1[12];
        StateScalars sc0;
        const AeroWeights wt1 = {a->k.wt.kblob, a->k.wt.kblob_dual, a->k.wt.pwl, a->k.wt.pwl_unnorm};
        nlplant<true, AB_REST, TILE, W, true, 0>(wt1, airframe_via(ap), s, u, sc0, coef, tables, k1, part);
        NP_REREAD_ARGS(ap);
        const float dt = ap->k.cfg.dt;
#pragma unroll
        for (int k = 0; k < 12; k++) s[k] = frozen ? s[k] : s[k] + dt * k1[k];
    }
    sc += 1;  // env_base.py:102
    NP_PSTAMP(5);

    // ---- Overload evaluation at the new state (overload.py:37-42) + the 14 coefficients of the next iteration ----
    StateScalars sc1;
    float xd[12];
    {
        const AeroWeights wt2 = {ap->k.wt.kblob, ap->k.wt.kblob_dual, ap->k.wt.pwl, ap->k.wt.pwl_unnorm};
        nlplant<false, AB_FORCE, TILE, W, true, 1>(wt2, airframe_via(ap), s, u, sc1, coef, tables, xd, part);
    }
    const Trig tr = sc1.tr;
    NP_REREAD_ARGS(ap);
    NP_PSTAMP(6);
    if (part == PLAN_STATE_WAVE) {
        const unsigned fl0 = __float_as_uint(ctx[CX::ST + CTX_FL * ROWS + r]);
        float acc3[3];
        body_acceleration(s, tr, xd, acc3);
        const bool done_prev = (fl0 & 1u) != 0, bad_prev = (fl0 & 2u) != 0;
        bool done = false, bad = false;
        float reward = 0.0f, reward_task = 0.0f;
        unsigned reasons = 0;
        done_and_reward<TASK>(ap->k.cfg, s, tgt, acc3, sc, done_prev, bad_prev, done, bad, reward, reasons, reward_task);
        // as the launch-by-launch kernel: with the per-aircraft bits tracked, the bits accumulated over the earlier inner iterations are
        // OR-ed in BEFORE the counters are fed (so a condition that fired earlier in the macro-step is counted in every later iteration)
        if (ap->k.term_reasons) reasons |= (fl0 >> 8) & 0x7Fu;
        if (ap->k.term_counters) {
#pragma unroll
            for (int k = 0; k < NP_NUM_TERM_COUNTERS; k++) {
                const unsigned long long m = __ballot(valid && ((reasons >> k) & 1u));
                if (m != 0 && (tid & 63) == 0) atomicAdd(ap->k.term_counters + k, (unsigned)__popcll(m));
            }
        }
        // inner iterations: the reason bits accumulate like the flags they explain; exceed_time_limit is carried (no Timeout condition fires inside)
        const unsigned fl1 = (done ? 1u : 0u) | (bad ? 2u : 0u) | (fl0 & 4u) | (fl0 & 0x7F00u) | (reasons << 8);
        if (t < ROWS) {
            float *sw = ctx + CX::ST + r;
#pragma unroll
            for (int k = 0; k < 12; k++) sw[k * ROWS] = s[k];
#pragma unroll
            for (int k = 0; k < 4; k++) sw[(12 + k) * ROWS] = u[k];
            sw[CTX_SC * ROWS] = __int_as_float((int)sc);
            sw[CTX_FL * ROWS] = __uint_as_float(fl1);
        }
        if (valid && last) {   // every iteration overwrites these: the last one's survive (planning_env.py:153-176)
            ap->k.reward[i] = reward;
            if (ap->k.reward_task) ap->k.reward_task[i] = reward_task;
        }
        if (valid && do_export) {
#pragma unroll
            for (int k = 0; k < 12; k++) gst<COH>(ap->k.s + k * ap->k.ld + i, s[k]);
#pragma unroll
            for (int k = 0; k < 4; k++) gst<COH>(ap->k.u + k * ap->k.ld + i, u[k]);
            gst<COH>(ap->k.step_count + i, sc);
            uint8_t *fout = ap->flags[(it & 1) ^ 1];
            gst<COH>(fout + i, (uint8_t)(fl1 & 1u));
            gst<COH>(fout + n + i, (uint8_t)((fl1 >> 1) & 1u));
            gst<COH>(fout + 2 * n + i, (uint8_t)((fl1 >> 2) & 1u));
            if (ap->k.term_reasons) gst<COH>(ap->k.term_reasons + i, (unsigned char)((fl1 >> 8) & 0x7Fu));
            float *cache_w = ap->k.cache + ((i >> 6) * NUM_CACHE_ROWS) * CACHE_TILE + (i & (CACHE_TILE - 1));
#pragma unroll
            for (int k = 0; k < NUM_CACHED; k++) gst<COH>(cache_w + k * CACHE_TILE, coef[cached_slot(k) * TILE]);
            gst<COH>(cache_w + CACHE_KEY0 * CACHE_TILE, s[7]);   // the (alpha, beta) they belong to
            gst<COH>(cache_w + (CACHE_KEY0 + 1) * CACHE_TILE, s[8]);
        }
    }
    if (part == 0 && !last && t < ROWS) {
        // PlanningEnv.low_level_obs (planning_env.py:60-142) of the state just reached, for the controller's next call: straight into the context
        float o2[22], t3[3];
#pragma unroll
        for (int k = 0; k < 3; k++) t3[k] = ctx[CX::ST + (CTX_LT + k) * ROWS + r];
        observe<1, true>(ap->k.cfg, s, u, t3, tr, o2, sc1.powv);
        float2 *row = reinterpret_cast<float2 *>(ctx + CX::OBS + t * 22);
#pragma unroll
        for (int k = 0; k < 11; k++) row[k] = make_float2(o2[2 * k], o2[2 * k + 1]);
    }
    NP_PSTAMP(7);
    if (want_obs) {   // the task observation of the last iteration: [rows][22] through an LDS tile (the controller's, idle), stored coalesced
        float o[22];
        if (part == 0) {
            observe<TASK, true>(ap->k.cfg, s, u, tgt, tr, o, sc1.powv);
            if (ap->k.noise) {
#pragma unroll
                for (int k = 0; k < 22; k++) o[k] = o[k] + ap->k.noise[ic * 22 + k] * ap->k.cfg.noise_scale;
            } else if (ap->k.cfg.noise_scale != 0.0f) {
#pragma unroll
                for (int pair = 0; pair < 11; pair++) {
                    const float rs = nz[(3 * pair) * TILE], cs = nz[(3 * pair + 1) * TILE], sn = nz[(3 * pair + 2) * TILE];
                    o[2 * pair] = fmaf(rs, cs, o[2 * pair]);
                    o[2 * pair + 1] = fmaf(rs, sn, o[2 * pair + 1]);
                }
            }
        }
        __syncthreads();  // wave 0 is done reading the noise columns before the tile overwrites them
        float *obs_tile = lds_act;
        const long long rows = (n - i0) < ROWS ? (n - i0) : ROWS;
        float *dst = ap->final_obs + i0 * 22;
        if (part == 0 && t < ROWS) {
#pragma unroll
            for (int k = 0; k < 22; k++) obs_tile[t * OBS_LD + k] = o[k];
        }
        __syncthreads();
        const int total = (int)rows * 22;
#pragma nounroll
        for (int base = 0; base < 22 * ROWS; base += TILE * W) {
            const int L = base + (int)tid;
            if (L < total) {
                const unsigned rr = ((unsigned)L * 2979u) >> 16;  // L / 22 for every L < 22 * 256
                dst[L] = obs_tile[(unsigned)L + rr];
            }
        }
    }
}

// ---- pipelined schedule (eight waves per tile, static): an iteration's Overload evaluation, terminations and reward do not feed the
// controller's next call — that needs only the next low-level observation, i.e. the new state and its trigonometry.  So an inner step is
// split: the FRONT (all eight waves: integrator evaluation + Euler; waves 4..7: the new state's fp64 chains; wave 0: the next
// observation) stays on the critical path, the BACK (Overload evaluation on the four-wave plan, terminations, counters, flags) runs on
// waves 4..7 WHILE waves 0..3 run the next controller call.  The two groups meet only at workgroup barriers (gfx950 has no named
// barriers): the back's two barriers — "coefficient columns free / inputs visible" and "columns complete" — are the controller call's
// barriers 10 and 11, which bracket its six GRU layers (33 K cycles without a barrier); before and after, waves 4..7 execute the call's
// other 21 barriers back to back.  Same evaluations on the same inputs as the sequential step: bit-identical.
constexpr int ACTOR32_BARRIERS_BEFORE_GRU = 9;   // barriers of actor32_body before "h -> LDS" (np_actor.h): obs LN 1, L1 1 + LN1 3, L2 1 + LN2 3
static_assert(npact::ACTOR32_BARRIERS == ACTOR32_BARRIERS_BEFORE_GRU + 2 + 12, "barrier plan of the pipelined schedule");

template <int W, bool WIN>
__device__ __forceinline__ void plan_fdm_front(PlanArgsC &ap, float *lds_fdm, float *ctx, long long i0, int it, unsigned tid) {
    static_assert(W == 8, "the pipelined schedule needs the four helper waves");
    NP_REREAD_ARGS(ap);
    const PlanArgsC a = ap;
    constexpr int TILE = PLAN_TILE;
    const int t = (int)(tid % TILE), r = t & (PLAN_ROWS - 1);
    const int part = __builtin_amdgcn_readfirstlane((int)(tid / TILE));
    float *coef = lds_fdm + t;
    float s[12], u[4];
    const float *st = ctx + CTX_ST + r;
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = st[k * PLAN_ROWS];
#pragma unroll
    for (int k = 0; k < 4; k++) u[k] = st[(12 + k) * PLAN_ROWS];
    long long sc = (long long)__float_as_int(st[CTX_SC * PLAN_ROWS]);
    const bool frozen = (__float_as_uint(st[CTX_FL * PLAN_ROWS]) & 7u) != 0;  // planning_env.py:162-166
    float act[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float v = ctx[CTX_ACT + r * 4 + k];
        v = v < -1.0f ? -1.0f : v;
        v = v > 1.0f ? 1.0f : v;
        act[k] = v;
    }
    control_lag(ap->k.cfg.af, act, u);
    NP_PSTAMP(4);
    {
        float k1[12];
        StateScalars sc0;
        const AeroWeights wt1 = {a->k.wt.kblob, a->k.wt.kblob_dual, a->k.wt.pwl, a->k.wt.pwl_unnorm};
        // WIN (static schedule): all 36 alpha/beta-only coefficients of this state are in the columns (the previous step's back and its
        // window nets, or the import): only the six el-dependent nets — the ones that see this step's action — are evaluated on the
        // critical path, one per wave; otherwise the 22 moment-side nets are evaluated here too (the cache carries the 14 force-side ones)
        nlplant<true, WIN ? AB_EL : AB_REST, TILE, W, true, 0>(wt1, airframe_via(ap), s, u, sc0, coef, false, k1, part);
        NP_REREAD_ARGS(ap);
        const float dt = ap->k.cfg.dt;
#pragma unroll
        for (int k = 0; k < 12; k++) s[k] = frozen ? s[k] : s[k] + dt * k1[k];
    }
    sc += 1;  // env_base.py:102
    NP_PSTAMP(5);
    if (part >= 4) {   // the new state's serial chains (set 1), one per helper wave as the four-wave plan splits them
        StateScalars scx;
        float xd[12];
        const AeroWeights wt2 = {ap->k.wt.kblob, ap->k.wt.kblob_dual, ap->k.wt.pwl, ap->k.wt.pwl_unnorm};
        nlplant<false, WIN ? AB_ABALL : AB_FORCE, TILE, 4, true, 1, false, 1>(wt2, airframe_via(ap), s, u, scx, coef, false, xd, part - 4);
    }
    if (part == 4 + PLAN_STATE_WAVE && t < PLAN_ROWS) {   // the state the back (and the next front) start from
        float *sw = ctx + CTX_ST + r;
#pragma unroll
        for (int k = 0; k < 12; k++) sw[k * PLAN_ROWS] = s[k];
#pragma unroll
        for (int k = 0; k < 4; k++) sw[(12 + k) * PLAN_ROWS] = u[k];
        sw[CTX_SC * PLAN_ROWS] = __int_as_float((int)sc);
    }
    __syncthreads();   // the shared scalars of the new state are published
    NP_PSTAMP(6);
    if (part == 0 && t < PLAN_ROWS) {
        // PlanningEnv.low_level_obs (planning_env.py:60-142) of the state just reached, for the controller's next call: straight into the context
        const float *shr = coef + (NUM_LDS_SLOTS + NUM_SHARED_SCALARS * 1) * TILE;
        Trig tr;
        tr.sa = shr[0 * TILE]; tr.ca = shr[1 * TILE]; tr.sb = shr[2 * TILE]; tr.cb = shr[3 * TILE];
        tr.st = shr[4 * TILE]; tr.ct = shr[5 * TILE]; tr.sphi = shr[6 * TILE]; tr.cphi = shr[7 * TILE];
        const float powv = shr[11 * TILE];
        float o2[22], t3[3];
#pragma unroll
        for (int k = 0; k < 3; k++) t3[k] = ctx[CTX_ST + (CTX_LT + k) * PLAN_ROWS + r];
        observe<1, true>(ap->k.cfg, s, u, t3, tr, o2, powv);
        float2 *row = reinterpret_cast<float2 *>(ctx + CTX_OBS + t * 22);
#pragma unroll
        for (int k = 0; k < 11; k++) row[k] = make_float2(o2[2 * k], o2[2 * k + 1]);
    }
    NP_PSTAMP(7);
}

// the BACK of an inner step on waves 4..7 (part4 = wave - 4), during the controller call that follows it: executes exactly two workgroup barriers
template <int TASK, bool WIN>
__device__ __forceinline__ void plan_fdm_back(PlanArgsC &ap, float *lds_fdm, float *ctx, long long i0, unsigned tid, int part4) {
    NP_REREAD_ARGS(ap);
    const PlanArgsC a = ap;
    constexpr int TILE = PLAN_TILE;
    const int t = (int)(tid % TILE), r = t & (PLAN_ROWS - 1);
    float *coef = lds_fdm + t;
    const long long n = a->k.n;
    const long long i = i0 + r;
    const bool valid = t < PLAN_ROWS && i < n;
    const long long ic = i < n ? i : n - 1;
    const unsigned o4 = (unsigned)ic * 4u;
    float s[12], u[4], tgt[3];
    const float *st = ctx + CTX_ST + r;
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = st[k * PLAN_ROWS];
#pragma unroll
    for (int k = 0; k < 4; k++) u[k] = st[(12 + k) * PLAN_ROWS];
    StateScalars sc1;
    float xd[12];
    {
        const AeroWeights wt2 = {a->k.wt.kblob, a->k.wt.kblob_dual, a->k.wt.pwl, a->k.wt.pwl_unnorm};
        // force-side build-up for the Overload check; WIN: 10 of the 22 moment-side nets (what the next front finds) ride along, the other
        // 12 are evaluated in the call's three short windows (plan_window_nets)
        nlplant<false, WIN ? AB_GRU : AB_FORCE, TILE, 4, true, 1, false, 2>(wt2, airframe_via(ap), s, u, sc1, coef, false, xd, part4);   // two barriers inside (eval_nets)
    }
    NP_REREAD_ARGS(ap);
    if (part4 == PLAN_STATE_WAVE) {
#pragma unroll
        for (int k = 0; k < 3; k++) tgt[k] = at_off(ap->k.tgt + k * ap->k.ld, o4);
        const long long sc = (long long)__float_as_int(st[CTX_SC * PLAN_ROWS]);
        const unsigned fl0 = __float_as_uint(st[CTX_FL * PLAN_ROWS]);
        float acc3[3];
        body_acceleration(s, sc1.tr, xd, acc3);
        const bool done_prev = (fl0 & 1u) != 0, bad_prev = (fl0 & 2u) != 0;
        bool done = false, bad = false;
        float reward = 0.0f, reward_task = 0.0f;
        unsigned reasons = 0;
        done_and_reward<TASK>(ap->k.cfg, s, tgt, acc3, sc, done_prev, bad_prev, done, bad, reward, reasons, reward_task);
        if (ap->k.term_reasons) reasons |= (fl0 >> 8) & 0x7Fu;   // as plan_fdm_step
        if (ap->k.term_counters) {
#pragma unroll
            for (int k = 0; k < NP_NUM_TERM_COUNTERS; k++) {
                const unsigned long long m = __ballot(valid && ((reasons >> k) & 1u));
                if (m != 0 && (tid & 63) == 0) atomicAdd(ap->k.term_counters + k, (unsigned)__popcll(m));
            }
        }
        const unsigned fl1 = (done ? 1u : 0u) | (bad ? 2u : 0u) | (fl0 & 4u) | (fl0 & 0x7F00u) | (reasons << 8);
        if (t < PLAN_ROWS) ctx[CTX_ST + CTX_FL * PLAN_ROWS + r] = __uint_as_float(fl1);
    }
}

// the cached coefficients of the tile's CURRENT state when the caller's cache is not valid for the first iteration: the force-side
// evaluation the previous step would have left (same nets, same inputs, same statements as the Overload evaluation that fills the
// cache in every step) -> the coefficient columns
template <int W, int ROWS = PLAN_ROWS>
__device__ __forceinline__ void plan_fill_cache(PlanArgsC &ap, float *lds_fdm, const float *ctx, unsigned tid) {
    using CX = CtxL<ROWS>;
    NP_REREAD_ARGS(ap);
    const PlanArgsC a = ap;
    constexpr int TILE = PLAN_TILE;
    const int t = (int)(tid % TILE), r = t & (ROWS - 1);
    const int part = __builtin_amdgcn_readfirstlane((int)(tid / TILE));
    float *coef = lds_fdm + t;
    float s[12], u[4];
    const float *st = ctx + CX::ST + r;
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = st[k * ROWS];
#pragma unroll
    for (int k = 0; k < 4; k++) u[k] = st[(12 + k) * ROWS];
    StateScalars sc1;
    float xd[12];
    const AeroWeights wt1 = {a->k.wt.kblob, a->k.wt.kblob_dual, a->k.wt.pwl, a->k.wt.pwl_unnorm};
    nlplant<false, AB_FORCE, TILE, W, true, 1>(wt1, airframe_via(ap), s, u, sc1, coef, false, xd, part);
    NP_REREAD_ARGS(ap);
}

// One cheap moment-side net per helper wave in a short barrier-free window of the controller call (the L2 / A1 / A2 dense layers): no barrier
// inside — the coefficient columns it writes are read by the NEXT front, many barriers later; the state is the one the previous front left
template <const SplitPlan &P>
__device__ __forceinline__ void plan_window_nets(PlanArgsC &ap, float *lds_fdm, const float *ctx, unsigned tid, int part4) {
    NP_REREAD_ARGS(ap);
    const PlanArgsC a = ap;
    constexpr int TILE = PLAN_TILE;
    const int t = (int)(tid % TILE), r = t & (PLAN_ROWS - 1);
    float *coef = lds_fdm + t;
    const float *st = ctx + CTX_ST + r;
    const float r2d = (float)(180.0 / 3.141592653589793);
    const float alpha = st[7 * PLAN_ROWS] * r2d, beta = st[8 * PLAN_ROWS] * r2d, el = st[(12 + 1) * PLAN_ROWS];   // as nlplant forms them
    const AeroWeights wt = {a->k.wt.kblob, a->k.wt.kblob_dual, a->k.wt.pwl, a->k.wt.pwl_unnorm};
    float xn[NUM_NORM_GROUPS];
    normalise_inputs(wt, alpha, beta, el, xn);
    if (part4 == 0) eval_plan_wave<P, 0, TILE>(wt, xn, coef, false);
    else if (part4 == 1) eval_plan_wave<P, 1, TILE>(wt, xn, coef, false);
    else if (part4 == 2) eval_plan_wave<P, 2, TILE>(wt, xn, coef, false);
    else eval_plan_wave<P, 3, TILE>(wt, xn, coef, false);
    NP_REREAD_ARGS(ap);
}

// pipelined schedule: ALL 36 alpha/beta-only coefficients of an imported tile's state -> the coefficient columns (the global cache carries the
// 14 force-side ones only, and a front evaluates none of them): the back's evaluation on waves 4..7, waves 0..3 pass its two barriers
__device__ __forceinline__ void plan_fill_ab(PlanArgsC &ap, float *lds_fdm, const float *ctx, unsigned tid) {
    const int part = __builtin_amdgcn_readfirstlane((int)(tid / PLAN_TILE));
    if (part < 4) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
        return;
    }
    NP_REREAD_ARGS(ap);
    const PlanArgsC a = ap;
    constexpr int TILE = PLAN_TILE;
    const int t = (int)(tid % TILE), r = t & (PLAN_ROWS - 1);
    float *coef = lds_fdm + t;
    float s[12], u[4];
    const float *st = ctx + CTX_ST + r;
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = st[k * PLAN_ROWS];
#pragma unroll
    for (int k = 0; k < 4; k++) u[k] = st[(12 + k) * PLAN_ROWS];
    StateScalars sc1;
    float xd[12];
    const AeroWeights wt1 = {a->k.wt.kblob, a->k.wt.kblob_dual, a->k.wt.pwl, a->k.wt.pwl_unnorm};
    nlplant<false, AB_ABALL, TILE, 4, true, 1>(wt1, airframe_via(ap), s, u, sc1, coef, false, xd, part - 4);
    NP_REREAD_ARGS(ap);
}

extern __shared__ __attribute__((aligned(16))) float np_plan_dyn_lds[];   // dual workgroups only (DUAL_LDS_FLOATS; 0 bytes otherwise)

template <int TASK, int W, bool QUEUE, bool DUAL = false, bool I8 = false>
__global__ __launch_bounds__(64 * W, 2) void planning_persistent_kernel(const PlanArgs a) {
    static_assert(W == 4 || W == 8, "four or eight waves per tile");
    static_assert(!DUAL || (W == 8 && !QUEUE), "dual workgroups: eight waves = two controller calls of four, static schedule");
    static_assert(!I8 || (W == 8 && !DUAL), "block-fixed-point controller: eight-wave tiles, static / guest / queue schedules");
    static_assert(npact8::ACTOR8_LDS_FLOATS <= npact::ACTOR32_LDS_FLOATS, "the i8 controller's LDS fits the fp32 controller's region");
    constexpr bool PIPE = NP_PLAN_PIPE && W == 8 && !DUAL;   // the pipelined schedule (plan_fdm_front / plan_fdm_back) inside a tile's stay on this workgroup
    // WIN: the pipelined schedule with the 22 moment-side alpha/beta-only nets moved off the critical path, into the controller call's four
    // barrier-free windows (plan_window_nets, AB_GRU) — for the static schedule, where a tile never changes workgroup: -2 % per macro-step at
    // n <= 32 rows x CUs; a tile that moves pays an evaluation of all 36 nets per import instead of 14, which cancels the gain (guest / queue
    // schedules: +0.2 .. +0.6 %, profiles/r04_planning_moment_nets_in_call_windows.log), so the coherent kernels keep the round-4 front
    constexpr bool WIN = NP_PLAN_WIN && PIPE && (!QUEUE || NP_PLAN_WIN_QUEUE);
    constexpr int ROWS = DUAL ? 2 * PLAN_ROWS : PLAN_ROWS;    // rows of the workgroup's context
    using CX = CtxL<ROWS>;
    constexpr int ACT_FLOATS = (DUAL ? 2 : 1) * npact::ACTOR32_LDS_FLOATS;
    constexpr bool PARK = QUEUE && W == 8;   // guest schedule: a host's own tile waits in (dynamic) LDS while the guest is in (PARK_LDS_FLOATS)
    __shared__ __attribute__((aligned(16))) float lds_static[DUAL ? 4 : PLAN_LDS_FLOATS];
    __shared__ unsigned item_s, stale_s, wait_ok_s;
    float *lds_all = DUAL ? np_plan_dyn_lds : lds_static;
    // dual: waves 4..7 run tile B's controller call in the second controller region
    float *lds_act = lds_all, *lds_fdm = lds_all + ACT_FLOATS, *ctx = lds_fdm + PLAN_COLS * PLAN_TILE;
    PlanArgsC ap = (PlanArgsC)__builtin_amdgcn_kernarg_segment_ptr();  // `a` is the only kernel parameter
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / 64));
#if NPACT_PRIO
    __builtin_amdgcn_s_setprio(NPACT_PRIO);
#endif
    float h[npact::BLK];  // the recurrent state of (row, block of 16 features): masked on the way into a controller call, new on the way out
#pragma unroll
    for (int j = 0; j < npact::BLK; j++) h[j] = 0.0f;
    if constexpr (I8) {   // the block-fixed-point controller's tables (scales, biases, LayerNorm parameters, head): staged once, read by every call
        npact8::actor8_stage_tables(np_plan_dyn_lds + (PARK ? PARK_LDS_FLOATS : 0) + npact8::ACTOR8_PARK_FLOATS, ap->actor_w, threadIdx.x, 64u * W);
        __syncthreads();
    }

    // one (tile, iteration) item.  do_import: the tile is not in this workgroup's registers / LDS yet; do_export: it leaves afterwards
    // seq: run the inner step sequentially (nothing left pending) although the tile stays; no_back: no front ran in the previous iteration on
    // this workgroup although the tile is resident (both: a host parking / un-parking its own tile around a guest block)
    auto run_item = [&](long long tile, int it, bool do_import, bool do_export, bool seq, bool no_back) {
        const long long i0 = tile * ROWS;
        // per-thread indices and LDS addresses are recomputed from this opaque copy in every iteration: kept across the loop they
        // are ~25 registers the allocator parks in scratch
        unsigned tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        NP_REREAD_ARGS(ap);
        const bool last = it == ap->iterations - 1;
        NP_PSTAMP(0);
        npact::Actor32Pre pre;
        // the controller's threads: waves 0..3; dual: waves 4..7 as well, on tile B (their own LDS region, thread index within the call, rows 32..63)
        const bool ctl = W == 4 || DUAL || wave < 4;
        const unsigned ctid = DUAL ? (tid & 255u) : tid;
        float *lds_ctl = lds_act + (DUAL && wave >= 4 ? npact::ACTOR32_LDS_FLOATS : 0);
        const int row0 = DUAL && wave >= 4 ? PLAN_ROWS : 0;
        if (ctl && !I8) {
            npact::actor32_request_l1(ap->actor_w, ctid, pre);
            if (do_import) npact::actor32_stage_head(lds_ctl, ap->actor_w, ctid);  // stays staged while the workgroup lives
        }
        if (do_import) {
            plan_import<W, QUEUE, ROWS, I8>(ap, lds_fdm, lds_act, ctx, i0, it, tid, h, &stale_s);
            __syncthreads();
            if constexpr (QUEUE) {
                if (W == 4 || wave < 4) {   // the masked recurrent state (gru.py:26) of (row, block) from the stage
                    const int row = (int)(tid & 31), blk = (int)(tid >> 5) & 7;
                    const float mk = ctx[CX::ST + CTX_MK * ROWS + row];
#pragma unroll
                    for (int j = 0; j < npact::BLK / 2; j++) {
                        const float2 v = *reinterpret_cast<const float2 *>(lds_act + h_stage_off(row, h_piece<I8>(blk, j)));
                        h[2 * j] = v.x * mk;
                        h[2 * j + 1] = v.y * mk;
                    }
                }
            }
            if constexpr (WIN) plan_fill_ab(ap, lds_fdm, ctx, tid);   // every import: the fronts rely on all 36 columns
            else if (it == 0 && (!ap->cache_valid0 || stale_s != 0u)) plan_fill_cache<W, ROWS>(ap, lds_fdm, ctx, tid);
        } else if (ctl) {   // resident: h holds the previous call's new state; gru.py:26 masks it
            const float mk = ctx[CX::ST + CTX_MK * ROWS + row0 + (int)(tid & 31)];
#pragma unroll
            for (int j = 0; j < npact::BLK; j++) h[j] = h[j] * mk;
        }
        // ---- controller (ppo_actor.py:38-64): context observation, h -> context actions, h ----
        NP_PSTAMP(1);
        if (ctl) {
#if NP_PLAN_BACK_PRIO
            __builtin_amdgcn_s_setprio(NP_PLAN_BACK_PRIO);
#endif
            NP_REREAD_ARGS(ap);
            const int row = row0 + (int)(tid & 31), hi = (int)((tid >> 5) & 1), w4 = (int)(ctid >> 6);
            float xr[npact::OBS];
#pragma unroll
            for (int j = 0; j < npact::OBS; j++) xr[j] = ctx[CX::OBS + row * 22 + j];
            float hn[npact::BLK], action;
            if constexpr (I8) {   // its GRU parking area and the staged tables: dynamic LDS
                float *i8_lds = np_plan_dyn_lds + (PARK ? PARK_LDS_FLOATS : 0);
                float action1[1];
                npact8::actor8_body<1>(lds_ctl, i8_lds, i8_lds + npact8::ACTOR8_PARK_FLOATS, ap->actor_w, reinterpret_cast<const float(&)[1][npact::OBS]>(xr),
                                       reinterpret_cast<const float(&)[1][npact::BLK]>(h), reinterpret_cast<float(&)[1][npact::BLK]>(hn), action1, ctid);
                action = action1[0];
            }
            else npact::actor32_body(lds_ctl, ap->actor_w, pre, xr, h, hn, action, ctid);
            if (hi == 0) ctx[CX::ACT + row * 4 + w4] = action;
#pragma unroll
            for (int j = 0; j < npact::BLK; j++) h[j] = hn[j];
            if (!QUEUE && do_export) {   // the recurrent state leaves: rnn[(it + 1) & 1] (the coherent variants: after the inner step, below)
                NP_REREAD_ARGS(ap);
                const long long i = i0 + row;
                if (i < ap->k.n) {
                    float *hq = ap->rnn[(it & 1) ^ 1] + i * npact::HID;
#pragma unroll
                    for (int j = 0; j < npact::BLK / 4; j++) reinterpret_cast<float4 *>(hq)[h_float4<I8>(2 * w4 + hi, j)] = make_float4(h[4 * j], h[4 * j + 1], h[4 * j + 2], h[4 * j + 3]);
                }
            }
        } else if constexpr (W == 8) {
            // waves 4..7 match the call's 23 barriers; in the pipelined schedule they run the BACK of the previous inner step meanwhile
            const bool back = PIPE && !do_import && !no_back;   // a front ran in this workgroup's previous iteration
            if constexpr (WIN) {
                // the fp32 call's barriers: obs LN 1 | L1 2 | LN1 3-5 | [L2 dense] 6 | LN2 7-9 | h -> LDS 10 | [six GRU layers] 11 | 12 | LN3 13-15 | [A1 dense] 16 |
                // LN4 17-19 | [A2 dense] 20 | LN5 21-23; the block-fixed-point call's: LN1 1-2, fragments 3 | [L2] | LN2 4-5, fragments 6 | [GRU] | LN3 7-8,
                // fragments 9 | [A1] | LN4 10-11, fragments 12 | [A2] | LN5 13-14, head 15.  The back works in the four bracketed windows.
                constexpr int N0 = I8 ? 3 : 5, N1 = I8 ? 2 : 4, N2 = I8 ? 2 : 4, N3 = I8 ? 3 : 4, N4 = I8 ? 3 : 4;
                static_assert((I8 ? npact8::ACTOR8_BARRIERS : npact::ACTOR32_BARRIERS) == N0 + N1 + 2 + N2 + N3 + N4 && (I8 || ACTOR32_BARRIERS_BEFORE_GRU == 9),
                              "barrier plan of the spread back");
#pragma unroll 1
                for (int b = 0; b < N0; b++) __builtin_amdgcn_s_barrier();
                if (back) plan_window_nets<PLAN_WIN_L2>(ap, lds_fdm, ctx, tid, wave - 4);
#pragma unroll 1
                for (int b = 0; b < N1; b++) __builtin_amdgcn_s_barrier();
                if (back) {
                    plan_fdm_back<TASK, true>(ap, lds_fdm, ctx, i0, tid, wave - 4);
                } else {
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_s_barrier();
                }
#pragma unroll 1
                for (int b = 0; b < N2; b++) __builtin_amdgcn_s_barrier();
                if (back) plan_window_nets<PLAN_WIN_A1>(ap, lds_fdm, ctx, tid, wave - 4);
#pragma unroll 1
                for (int b = 0; b < N3; b++) __builtin_amdgcn_s_barrier();
                if (back) plan_window_nets<PLAN_WIN_A2>(ap, lds_fdm, ctx, tid, wave - 4);
#pragma unroll 1
                for (int b = 0; b < N4; b++) __builtin_amdgcn_s_barrier();
            } else {
                // fp32 call: 9 barriers, [GRU window = barriers 10, 11], 12 more; block-fixed-point call: 5, [GRU window = 6, 7], 8 more
                constexpr int BEFORE = I8 ? 5 : ACTOR32_BARRIERS_BEFORE_GRU, TOTALB = I8 ? npact8::ACTOR8_BARRIERS : npact::ACTOR32_BARRIERS;
#pragma unroll 1
                for (int b = 0; b < BEFORE; b++) __builtin_amdgcn_s_barrier();
                if (back) {
#if NP_PLAN_BACK_PRIO   // experiment (tools/microbench): the controller's waves at raised priority while the back runs beside them
                    __builtin_amdgcn_s_setprio(0);
#endif
                    plan_fdm_back<TASK, false>(ap, lds_fdm, ctx, i0, tid, wave - 4);
                } else {
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_s_barrier();
                }
#pragma unroll 1
                for (int b = 0; b < TOTALB - BEFORE - 2; b++) __builtin_amdgcn_s_barrier();
            }
        }
        NP_PSTAMP(2);
        __syncthreads();  // the tile's actions are in the context (and the previous step's flags, pipelined schedule)
        NP_REREAD_ARGS(ap);
        NP_PSTAMP(3);
        if (PIPE && !do_export && !seq) {   // the tile's next iteration runs here too: its controller call hides this step's back
            if constexpr (W == 8) plan_fdm_front<W, WIN>(ap, lds_fdm, ctx, i0, it, tid);
        } else {
            plan_fdm_step<TASK, W, QUEUE, ROWS>(ap, lds_fdm, lds_act, ctx, i0, it, last, do_export, tid);
        }
        __syncthreads();  // the context holds the tile's next observation / state
        NP_PSTAMP(8);
        if constexpr (QUEUE) {
            if (do_export) {   // the recurrent state leaves through the stage (the controller's LDS is idle; the last barrier covers the observation tile)
                if (W == 4 || wave < 4) {
                    const int row = (int)(tid & 31), blk = (int)(tid >> 5) & 7;
#pragma unroll
                    for (int j = 0; j < npact::BLK / 2; j++) *reinterpret_cast<float2 *>(lds_act + h_stage_off(row, h_piece<I8>(blk, j))) = make_float2(h[2 * j], h[2 * j + 1]);
                }
                __syncthreads();
                NP_REREAD_ARGS(ap);
                h_stage_to_global<W>(ap->rnn[(it & 1) ^ 1], i0, ap->k.n, lds_act, tid);
            }
        }
        if (QUEUE && do_export && !last) {   // the next observation leaves too: ll_obs[(it + 1) & 1], 704 dwords
            NP_REREAD_ARGS(ap);
            float *dst = ap->ll_obs[(it & 1) ^ 1] + i0 * 22;
            const long long avail = (ap->k.n - i0) * 22;
#pragma unroll 1
            for (int L = (int)tid; L < PLAN_ROWS * 22; L += 64 * W) {
                if (L < avail) gst<true>(dst + L, ctx[CTX_OBS + L]);
            }
        }
    };

    if (!QUEUE) {
        const long long tile = blockIdx.x;
#pragma nounroll
        for (int it = 0;; it++) {
            NP_REREAD_ARGS(ap);
            const int iters = ap->iterations;
            if (it >= iters) break;
            run_item(tile, it, it == 0, it == iters - 1, false, false);
        }
    } else {
        // SEGMENTS: (tile, iterations [it0, it1)) — the tile is imported, stays resident for the segment, is exported, and
        // queue[1 + tile] = it1 publishes how far it has come; a segment with it0 > 0 first waits for that word to reach it0.
        //   dynamic (ap->guest_blocks == 0): segment id = block index * tiles + tile from the atomic counter queue[0], blocks of
        //       ap->block iterations, handed out in order;
        //   guests  (ap->guest_blocks = B > 0): no counter.  grid = C resident workgroups, tiles = C + G.  Workgroup w owns tile w; the
        //       G guest tiles' iterations are cut into B blocks each and block j of guest g is hosted by workgroup g * B + j, which runs
        //       own [0, p) | guest [b_j, b_j+1) | own [p, iterations) with p = b_j + j * ap->block ("slack": by then block j - 1,
        //       which started `slack` earlier on its host, is through).  Every workgroup hosts at most one block: the makespan is
        //       iterations + the longest block instead of 2 x iterations, with four exports / imports per host instead of one per item.
        int seg = 0;
        bool park_after = false, unpark_before = false;
#pragma nounroll
        for (;;) {
            NP_REREAD_ARGS(ap);
            const int iters = ap->iterations;
            const int B = ap->guest_blocks;
            long long tile;
            int it0, it1;
            if (B == 0) {
                if (threadIdx.x == 0) item_s = __hip_atomic_fetch_add(ap->queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
                const unsigned id = item_s - ap->queue_base;   // the counter is never reset: it runs on from launch to launch
                const unsigned tiles = (unsigned)ap->tiles;
                const int per = ap->block;
                const unsigned nblk = (unsigned)((iters + per - 1) / per);
                if (id >= tiles * nblk) break;
                tile = (long long)(id % tiles);
                it0 = (int)(id / tiles) * per;
                it1 = it0 + per < iters ? it0 + per : iters;
            } else {
                const int w = (int)blockIdx.x, C = (int)gridDim.x;
                const int G = (int)ap->tiles - C;
                const bool host = w < G * B;
                const int j = host ? w % B : 0;
                const int b0 = host ? (int)((long long)j * iters / B) : 0, b1 = host ? (int)((long long)(j + 1) * iters / B) : 0;
                int p = host ? b0 + j * ap->block : iters;
                p = p < iters ? p : iters;
                // segments in order: own [0, p) (if any), guest [b0, b1) (if hosting), own [p, iters) (if any)
                if (seg == 0 && p == 0) seg = 1;
                if (seg == 1 && !host) seg = 2;
                if (seg == 2 && p >= iters) seg = 3;
                if (seg >= 3) break;
                tile = seg == 1 ? (long long)(C + w / B) : (long long)w;
                it0 = seg == 0 ? 0 : seg == 1 ? b0 : p;
                it1 = seg == 0 ? p : seg == 1 ? b1 : iters;
                // a host's own tile stays on the CU while the guest is in: parked in LDS (eight-wave workgroups: PARK) instead of exported
                // and imported again — context, the 14 cached coefficient columns, the recurrent state
                park_after = PARK && seg == 0 && p < iters;
                unpark_before = PARK && seg == 2 && p > 0;
                seg++;
            }
            if (unpark_before) {
                unsigned tid = threadIdx.x;
                asm volatile("" : "+v"(tid));
                float *park = np_plan_dyn_lds;
                for (int L = (int)tid; L < CX::FLOATS; L += 64 * W) ctx[L] = park[L];
                if (tid < 64) {
#pragma unroll
                    for (int k = 0; k < NUM_CACHED; k++) lds_fdm[cached_slot(k) * PLAN_TILE + tid] = park[CX::FLOATS + k * PLAN_TILE + tid];
                }
                if (tid < 256) {
#pragma unroll
                    for (int jj = 0; jj < npact::BLK; jj++) h[jj] = park[CX::FLOATS + NUM_CACHED * PLAN_TILE + jj * 256 + tid];
                }
                __syncthreads();
                if constexpr (WIN) plan_fill_ab(ap, lds_fdm, ctx, tid);   // (NP_PLAN_WIN_QUEUE builds) the parked tile left through a sequential step: its moment-side columns are one state old
            } else if (it0 > 0) {
                if (threadIdx.x == 0) {
                    // progress words carry flag_base + iterations done; what an earlier launch left is below flag_base.
                    // The wait is BOUNDED: every 64 polls thread 0 looks at the launch's error word and at the wall clock; a wait that outlasts
                    // ap->wait_ticks claims the error word (first one wins), writes the host record and the workgroup drains — as does every
                    // workgroup that finds the word set.  No progress word is raised after that, so all remaining waits end the same way.
                    unsigned ok = 1u, polls = 0u;
                    const unsigned long long t0 = wall_clock64();
                    for (;;) {
                        const int have = (int)(__hip_atomic_load(ap->queue + 1 + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - ap->flag_base);
                        if (have >= it0) break;
                        __builtin_amdgcn_s_sleep(8);
                        if ((++polls & 63u) != 0u) continue;
                        if (__hip_atomic_load(ap->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0u; break; }
                        const unsigned long long waited = wall_clock64() - t0;
                        if (waited > ap->wait_ticks) {
                            unsigned expected = 0u;
                            if (__hip_atomic_compare_exchange_strong(ap->err, &expected, 1u + blockIdx.x, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                                unsigned *eh = ap->err_host;
                                __hip_atomic_store(eh + 1, (unsigned)tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                __hip_atomic_store(eh + 2, (unsigned)it0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                __hip_atomic_store(eh + 3, (unsigned)(have < 0 ? 0 : have), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                __hip_atomic_store(eh + 4, (unsigned)(waited / 100000ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                __hip_atomic_store(eh + 0, 1u + blockIdx.x, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                            }
                            ok = 0u;
                            break;
                        }
                    }
                    wait_ok_s = ok;
                }
                __syncthreads();   // what the tile's previous owner exported was complete before it raised the flag; the imports are sc1 loads
                if (wait_ok_s == 0u) break;   // workgroup-uniform: drain (nothing of this segment was imported; nothing more is published)
            }
            asm volatile("" ::: "memory");
#pragma nounroll
            for (int it = it0; it < it1; it++) {
                const bool last_it = it == it1 - 1;
                run_item(tile, it, it == it0 && !unpark_before, last_it && !park_after, last_it && park_after, it == it0 && unpark_before);
            }
            if (park_after) {   // nothing left the CU: no progress word to raise (nobody else touches a host's own tile)
                unsigned tid = threadIdx.x;
                asm volatile("" : "+v"(tid));
                float *park = np_plan_dyn_lds;
                for (int L = (int)tid; L < CX::FLOATS; L += 64 * W) park[L] = ctx[L];
                if (tid < 64) {
#pragma unroll
                    for (int k = 0; k < NUM_CACHED; k++) park[CX::FLOATS + k * PLAN_TILE + tid] = lds_fdm[cached_slot(k) * PLAN_TILE + tid];
                }
                if (tid < 256) {
#pragma unroll
                    for (int jj = 0; jj < npact::BLK; jj++) park[CX::FLOATS + NUM_CACHED * PLAN_TILE + jj * 256 + tid] = h[jj];
                }
                __syncthreads();
                continue;
            }
            // every wave: its exports (sc1 stores) have completed; then the flag
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            NP_REREAD_ARGS(ap);
            if (threadIdx.x == 0 && !(ap->debug_stall && blockIdx.x == 0))   // (debug_stall: the fault-injection hook of tests/test_gpu_actor.py)
                __hip_atomic_store(ap->queue + 1 + tile, ap->flag_base + (unsigned)it1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

namespace {
template <int TASK, int W, bool QUEUE, bool I8>
hipError_t launch_one(const PlanArgs &args, unsigned grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    // dynamic LDS: the parking area of the guest schedule's hosts, then the block-fixed-point controller's GRU parking area
    constexpr size_t dyn = sizeof(float) * (((QUEUE && W == 8) ? PARK_LDS_FLOATS : 0) + (I8 ? npact8::ACTOR8_PARK_FLOATS + npact8::TAB_FLOATS : 0));
    const auto kernel = planning_persistent_kernel<TASK, W, QUEUE, false, I8>;
    if constexpr (dyn != 0) {
        static bool set[64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return hipGetLastError();
        if (dev < 64 && !set[dev]) {  // static + dynamic LDS above the 64 KB a kernel may use without asking
            const hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
            if (e != hipSuccess) return e;
            set[dev] = true;
        }
    }
    if (e0 && e1) hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(64 * W), dyn, st, e0, e1, 0, args);
    else hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * W), dyn, st, args);
    return hipGetLastError();
}
template <int TASK>
hipError_t launch_dual(const PlanArgs &args, unsigned grid, hipStream_t st) {
    constexpr size_t bytes = sizeof(float) * DUAL_LDS_FLOATS;
    static bool set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipGetLastError();
    if (dev < 64 && !set[dev]) {  // above the 64 KB a kernel may use without asking
        const hipError_t e = hipFuncSetAttribute((const void *)planning_persistent_kernel<TASK, 8, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
        set[dev] = true;
    }
    hipLaunchKernelGGL((planning_persistent_kernel<TASK, 8, false, true>), dim3(grid), dim3(512), bytes, st, args);
    return hipGetLastError();
}
template <int TASK, int W, bool I8>
int occupancy_of() {
    // the kernel the guest / queue schedules launch — the instantiation of the controller numerics at hand (the block-fixed-point
    // build is another kernel with more dynamic LDS) — with the dynamic LDS launch_one gives it (the hosts' parking area, the i8 controller's GRU
    // parking area and staged tables: ADVICE r4 — a query with fewer bytes would over-count once the registers allow two workgroups per CU;
    // grid = resident and B = resident / guests rest on it)
    constexpr size_t dyn = sizeof(float) * ((W == 8 ? PARK_LDS_FLOATS : 0) + (I8 ? npact8::ACTOR8_PARK_FLOATS + npact8::TAB_FLOATS : 0));
    const auto kernel = planning_persistent_kernel<TASK, W, true, false, I8>;
    if constexpr (dyn != 0) {
        if (hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
    }
    int blocks = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kernel, 64 * W, dyn) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return blocks;
}
}  // namespace

#ifndef NP_PLAN_TASKS
#define NP_PLAN_TASKS 4  // bit t: build the persistent kernels of task t.  Tracking only: the one task the reference's PlanningEnv accepts
#endif                   // (envs/planning_env.py:57-60 raises NotImplementedError otherwise); contexts of the other tasks take the launches

hipError_t launch_planning_dual(int task, const PlanArgs &args, unsigned grid, hipStream_t st) {
    if constexpr ((NP_PLAN_TASKS & 1) != 0) { if (task == 0) return launch_dual<0>(args, grid, st); }
    if constexpr ((NP_PLAN_TASKS & 2) != 0) { if (task == 1) return launch_dual<1>(args, grid, st); }
    if constexpr ((NP_PLAN_TASKS & 4) != 0) { if (task == 2) return launch_dual<2>(args, grid, st); }
    return hipErrorInvalidValue;
}

// Eight waves per tile.  (The four-wave builds of rounds 3-4 — 256 VGPRs + 36-52 B of scratch per lane, reachable only through an explicit
// np_planning_loop.waves = 4: NP_PLANNING_AUTO always took eight — were retired in round 5: VERDICT r4 item 6.)
hipError_t launch_planning_persistent(int task, int waves, bool i8, const PlanArgs &args, unsigned grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    const bool queue = args.queue != nullptr;
    if (waves != 8) return hipErrorInvalidValue;
#define NP_PLAN_CASE(T)                                                                             \
    if constexpr (((NP_PLAN_TASKS >> T) & 1) != 0) {                                                  \
        if (task == T) {                                                                             \
            if (i8) return queue ? launch_one<T, 8, true, true>(args, grid, st, e0, e1) : launch_one<T, 8, false, true>(args, grid, st, e0, e1);   \
            return queue ? launch_one<T, 8, true, false>(args, grid, st, e0, e1) : launch_one<T, 8, false, false>(args, grid, st, e0, e1); \
        }                                                                                            \
    }
    NP_PLAN_CASE(0)
    NP_PLAN_CASE(1)
    NP_PLAN_CASE(2)
#undef NP_PLAN_CASE
    return hipErrorInvalidValue;
}

bool planning_persistent_built(int task) { return task >= 0 && task <= 2 && ((NP_PLAN_TASKS >> task) & 1) != 0; }

static int workgroups_per_cu_uncached(int task, int waves, bool i8);
int planning_persistent_workgroups_per_cu(int task, int waves, bool i8) {
    static int cached[64][3][2][2] = {};   // the occupancy query costs a driver call; its answer belongs to the code object (per numerics) and the device
    if (task < 0 || task > 2) return 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return workgroups_per_cu_uncached(task, waves, i8);
    int &c = cached[dev][task][waves == 8][i8 ? 1 : 0];
    if (c == 0) c = workgroups_per_cu_uncached(task, waves, i8);
    return c;
}
static int workgroups_per_cu_uncached(int task, int waves, bool i8) {
#define NP_PLAN_CASE(T)                                                       \
    if constexpr (((NP_PLAN_TASKS >> T) & 1) != 0) {                        \
        if (task == T) return waves != 8 ? 0 : i8 ? occupancy_of<T, 8, true>() : occupancy_of<T, 8, false>(); \
    }
    NP_PLAN_CASE(0)
    NP_PLAN_CASE(1)
    NP_PLAN_CASE(2)
#undef NP_PLAN_CASE
    return 0;
}

}  // namespace npf16

#if NP_PLAN_TRACE
extern "C" int np_plan_trace_read(unsigned long long *out128, long long *actor64) {  // diagnostics builds only (tools/microbench/planning_phases.py)
    if (hipMemcpyFromSymbol(out128, HIP_SYMBOL(npf16::np_plan_trace_buf), sizeof(unsigned long long) * 128) != hipSuccess) return 1;
#if NPACT_TRACE
    if (actor64 && hipMemcpyFromSymbol(actor64, HIP_SYMBOL(npact::npact_trace), sizeof(long long) * 64) != hipSuccess) return 1;
#endif
    return 0;
}
#endif
