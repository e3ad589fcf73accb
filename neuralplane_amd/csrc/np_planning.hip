// np_planning.hip — PlanningEnv's 50 low-level iterations as ONE persistent gfx950 kernel.
//
// Reference: envs/planning_env.py:153-176 — per iteration: ego_actions = controller(low_level_obs, rnn_states, masks);
// model.update(ego_actions); step_count += 1; terminations / reward; the next low_level_obs.  np_planning_inner_loop
// (np_f16_kernels.hip) enqueues that as 2 x 50 launches (np_actor_forward + np_f16_step with inner_step); here a workgroup owns a
// tile of 32 aircraft — one 32-row tile of the controller's MFMA kernel (np_actor.h::actor_tile32) — and runs
//     controller call  ->  barrier  ->  inner FDM step (np_f16_device.h, the latency variant's device code on the tile's rows)  ->  barrier
// for every iteration inside one launch: no kernel boundary (dispatch + first loads, ~9 us of a 50 us iteration at n <= 8 192).
// The arithmetic is the launch-by-launch path's, operation by operation (same device functions, same generated statements, same
// order): results are bit-identical (tests/test_gpu_actor.py).
//
// Two schedules:
//   static : grid = tiles, workgroup b runs all iterations of tile b (tiles <= resident workgroups: n <= 8 192 at one per CU);
//   queue  : `grid` resident workgroups pull (tile, iteration) items, id = iteration * tiles + tile, from an atomic counter; an item
//            waits until its tile's previous iteration is published (release / acquire at agent scope through queue[1 + tile]).  The
//            lowest outstanding id never waits on anything unfinished, so the schedule cannot deadlock as long as the grid is
//            resident (the launcher sizes it by the occupancy query).  313 tiles on 256 CUs then take 62 rounds of items instead
//            of 2 x 50 lock-step iterations.
// The tile's data (recurrent state, low-level observation, actions, aircraft state, coefficient cache) travels through global memory
// between the two halves of an iteration exactly as between the launches it replaces — L2-resident at these sizes.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "../../include/neuralplane_amd.h"
#include "np_planning.h"
#define NPACT_NO_KERNELS 1
#include "np_actor.h"

namespace npf16 {

typedef const PlanArgs __attribute__((address_space(4))) *PlanArgsC;

constexpr int PLAN_TILE = 64;  // lanes per wave = LDS column pitch of the FDM device code; lanes 32..63 shadow rows 0..31 (stores masked)
constexpr int PLAN_STATE_WAVE = 1;  // as the latency variant: wave 1 evaluates terminations / reward and stores the state
constexpr int PLAN_NOISE_COL0 = NUM_LDS_SLOTS + 2 * NUM_SHARED_SCALARS;
constexpr int PLAN_COLS = NUM_LDS_SLOTS + 2 * NUM_SHARED_SCALARS + 33;
constexpr int PLAN_FDM_LDS = (PLAN_COLS * PLAN_TILE > PLAN_TILE * OBS_LD) ? PLAN_COLS * PLAN_TILE : PLAN_TILE * OBS_LD;

// One inner FDM step (np_f16_step with inner_step: f16_env_kernel<TASK, 0, true, true, 64, W, true>) of rows [i0, i0 + 32):
// no auto-reset, flagged rows frozen, flags accumulate; writes the controller's next observation unless `last`, the task
// observation if `last`.  fin / fout: the flag planes [3][n] read / written by this iteration.
template <int TASK, int W>
__device__ __forceinline__ void plan_fdm_step(PlanArgsC &ap, float *lds, long long i0, int it, bool last, unsigned tid) {
    // every scalar is (re-)read from the kernel-argument segment where it is used: read through the by-value parameter the compiler
    // hoists the loads out of the iteration loop and keeps ~60 SGPRs alive across the asm phases (parked in VGPR lanes, then scratch)
    NP_REREAD_ARGS(ap);
    const PlanArgsC a = ap;
    const uint8_t *fin = a->flags[it & 1];
    constexpr int TILE = PLAN_TILE;
    float *obs_tile = lds;
    const int t = (int)(tid % TILE);
    const int part = __builtin_amdgcn_readfirstlane((int)(tid / TILE));
    float *coef = lds + t;
    const long long n = a->k.n;
    const long long i = i0 + (t & (PLAN_ROWS - 1));
    const bool valid = t < PLAN_ROWS && i < n;
    const long long ic = i < n ? i : n - 1;
    const bool tables = false;  // the persistent kernel serves the MLP numerics (the launcher falls back otherwise)
    const bool want_obs = last && a->final_obs != nullptr;

    const unsigned r32 = (unsigned)ic, o4 = r32 * 4u, o8 = r32 * 8u;
    const unsigned nn = (unsigned)n;
    float s[12], u[4], tgt[3];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = at_off(a->k.s + k * a->k.ld, o4);
#pragma unroll
    for (int k = 0; k < 4; k++) u[k] = at_off(a->k.u + k * a->k.ld, o4);
#pragma unroll
    for (int k = 0; k < 3; k++) tgt[k] = at_off(a->k.tgt + k * a->k.ld, o4);
    long long sc = at_off(a->k.step_count, o8);
    const unsigned f0 = at_off(fin, r32), f1 = at_off(fin, r32 + nn), f2 = at_off(fin, r32 + 2u * nn);
    const unsigned fl_in = f0 | f1 | f2;
    if (want_obs && !a->k.noise && a->k.cfg.noise_scale != 0.0f) {  // this wave's share of the observation noise (f16_env_kernel, SHARED)
        const int nb = W == 8 ? part - 4 : part;
        if (nb >= 0) {
            uint32_t blk[4], k1[3], k2[3];
            rng_block(a->k.seed, a->k.call_idx + (uint64_t)it + (a->k.call_idx_base ? *a->k.call_idx_base : 0ull), a->k.row0 + ic, 2u + (uint32_t)nb, blk);
            noise_block_indices(blk, k1, k2);
            float *nz = coef + PLAN_NOISE_COL0 * TILE;
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
    const bool flagged = fl_in != 0;
    const bool frozen = flagged;  // planning_env.py:162-166
    const bool tmo_prev = f2 != 0;

    // coefficient columns <- the cross-step cache (layout [row / 64][NUM_CACHE_ROWS][row % 64])
    {
        const float *cache_blk = a->k.cache + ((ic >> 6) * NUM_CACHE_ROWS) * CACHE_TILE + (ic & (CACHE_TILE - 1));
#pragma unroll
        for (int k = 0; k < NUM_CACHED; k++) coef[cached_slot(k) * TILE] = cache_blk[k * CACHE_TILE];
    }

    // ---- F16Model.update (F16_model.py:51-67) ----
    float act[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float v = a->ll_act[ic * 4 + k];
        v = v < -1.0f ? -1.0f : v;
        v = v > 1.0f ? 1.0f : v;
        act[k] = v;
    }
    u[0] = 0.9f * u[0] + NP_DIVC(((0.1f * act[0]) * 0.225f) * 76300.0f, 0.3048f);
    u[1] = 0.9f * u[1] + (0.1f * act[1]) * 45.0f;
    u[2] = 0.9f * u[2] + (0.1f * act[2]) * 45.0f;
    u[3] = 0.9f * u[3] + (0.1f * act[3]) * 45.0f;
    {
        float k1[12];
        StateScalars sc0;
        const AeroWeights wt1 = {a->k.wt.kblob, a->k.wt.kblob_dual, a->k.wt.pwl, a->k.wt.pwl_unnorm};
        nlplant<true, AB_REST, TILE, W, true, 0>(wt1, s, u, sc0, coef, tables, k1, part);
        NP_REREAD_ARGS(ap);
        const float dt = ap->k.cfg.dt;
#pragma unroll
        for (int k = 0; k < 12; k++) s[k] = frozen ? s[k] : s[k] + dt * k1[k];
    }
    sc += 1;  // env_base.py:102

    // ---- Overload evaluation at the new state (overload.py:37-42) + the 14 coefficients of the next iteration ----
    StateScalars sc1;
    float xd[12];
    {
        const AeroWeights wt2 = {ap->k.wt.kblob, ap->k.wt.kblob_dual, ap->k.wt.pwl, ap->k.wt.pwl_unnorm};
        nlplant<false, AB_FORCE, TILE, W, true, 1>(wt2, s, u, sc1, coef, tables, xd, part);
    }
    const Trig tr = sc1.tr;
    NP_REREAD_ARGS(ap);
    bool done = false, bad = false;
    float reward = 0.0f;
    if (part == PLAN_STATE_WAVE) {
        float acc3[3];
        body_acceleration(s, tr, xd, acc3);
        const bool done_prev = f0 != 0, bad_prev = f1 != 0;
        unsigned reasons = 0;
        float reward_task = 0.0f;
        done_and_reward<TASK>(ap->k.cfg, s, tgt, acc3, sc, done_prev, bad_prev, done, bad, reward, reasons, reward_task);
        if (ap->k.reward_task && valid) ap->k.reward_task[i] = reward_task;
        if (ap->k.term_reasons && valid) {
            reasons |= ap->k.term_reasons[i];  // inner iterations: the bits accumulate like the flags they explain
            ap->k.term_reasons[i] = (unsigned char)reasons;
        }
        if (ap->k.term_counters) {
#pragma unroll
            for (int k = 0; k < NP_NUM_TERM_COUNTERS; k++) {
                const unsigned long long m = __ballot(valid && ((reasons >> k) & 1u));
                if (m != 0 && (tid & 63) == 0) atomicAdd(ap->k.term_counters + k, (unsigned)__popcll(m));
            }
        }
    }
    float o[22];
    if (part == 0 && want_obs) {
        observe<TASK, true>(ap->k.cfg, s, u, tgt, tr, o, sc1.powv);
        if (ap->k.noise) {
#pragma unroll
            for (int k = 0; k < 22; k++) o[k] = o[k] + ap->k.noise[ic * 22 + k] * ap->k.cfg.noise_scale;
        } else if (ap->k.cfg.noise_scale != 0.0f) {
            const float *nz = coef + PLAN_NOISE_COL0 * TILE;
#pragma unroll
            for (int pair = 0; pair < 11; pair++) {
                const float rs = nz[(3 * pair) * TILE], cs = nz[(3 * pair + 1) * TILE], sn = nz[(3 * pair + 2) * TILE];
                o[2 * pair] = fmaf(rs, cs, o[2 * pair]);
                o[2 * pair + 1] = fmaf(rs, sn, o[2 * pair + 1]);
            }
        }
    }

    if (valid && part == PLAN_STATE_WAVE) {
        unsigned iw = (unsigned)i;
        asm volatile("" : "+v"(iw));
        const unsigned w4 = iw * 4u;
#pragma unroll
        for (int k = 0; k < 12; k++) at_off(ap->k.s + k * ap->k.ld, w4) = s[k];
#pragma unroll
        for (int k = 0; k < 4; k++) at_off(ap->k.u + k * ap->k.ld, w4) = u[k];
        at_off(ap->k.step_count, iw * 8u) = sc;
        uint8_t *fout = ap->flags[(it & 1) ^ 1];
        at_off(fout, iw) = done ? 1 : 0;
        at_off(fout, iw + nn) = bad ? 1 : 0;
        at_off(fout, iw + 2u * nn) = tmo_prev ? 1 : 0;
        at_off(ap->k.reward, w4) = reward;
        float *cache_w = ap->k.cache + ((long long)(iw >> 6) * NUM_CACHE_ROWS) * CACHE_TILE + (iw & (CACHE_TILE - 1));
#pragma unroll
        for (int k = 0; k < NUM_CACHED; k++) cache_w[k * CACHE_TILE] = coef[cached_slot(k) * TILE];
    }

    // ---- [rows][22] observation rows: transpose through LDS, store coalesced ----
    auto store_rows22 = [&](float *out_base, const float (&ov)[22]) {
        __syncthreads();  // every lane is done with its coefficient column before the tile overwrites it
        const long long rows = (n - i0) < PLAN_ROWS ? (n - i0) : PLAN_ROWS;
        float *dst = out_base + i0 * 22;
        constexpr int THREADS = TILE * W;
        if (rows == PLAN_ROWS && ((uintptr_t)dst & 15) == 0) {
            if (part == 0 && t < PLAN_ROWS) {
                float2 *row = reinterpret_cast<float2 *>(obs_tile + t * 22);
#pragma unroll
                for (int k = 0; k < 11; k++) row[k] = make_float2(ov[2 * k], ov[2 * k + 1]);
            }
            __syncthreads();
            constexpr int VECS = PLAN_ROWS * 22 / 4;
            static_assert(VECS <= THREADS, "one 16-byte vector per thread");
            const float4 *src4 = reinterpret_cast<const float4 *>(obs_tile);
            float4 *dst4 = reinterpret_cast<float4 *>(dst);
            if ((int)tid < VECS) dst4[tid] = src4[tid];
        } else {
            if (part == 0 && t < PLAN_ROWS) {
#pragma unroll
                for (int k = 0; k < 22; k++) obs_tile[t * OBS_LD + k] = ov[k];
            }
            __syncthreads();
            const int total = (int)rows * 22;
#pragma nounroll
            for (int base = 0; base < 22 * PLAN_ROWS; base += THREADS) {
                const int L = base + (int)tid;
                if (L < total) {
                    const unsigned r = ((unsigned)L * 2979u) >> 16;  // L / 22 for every L < 22 * 256
                    dst[L] = obs_tile[(unsigned)L + r];
                }
            }
        }
    };
    if (want_obs) store_rows22(ap->final_obs, o);
    if (!last) {
        // PlanningEnv.low_level_obs (planning_env.py:60-142) of the state just reached, for the controller's next call
        float o2[22];
        if (part == 0) {
            float t3[3];
#pragma unroll
            for (int k = 0; k < 3; k++) t3[k] = at_off(ap->k.ll_tgt + k * ap->k.ld, o4);
            observe<1, true>(ap->k.cfg, s, u, t3, tr, o2, sc1.powv);
        }
        store_rows22(ap->ll_obs[(it & 1) ^ 1], o2);
    }
}

// the cached coefficients of the tile's CURRENT state when the caller's cache is not valid for the first iteration: the force-side
// evaluation the previous step would have left (same nets, same inputs, same statements as the Overload evaluation that fills the
// cache in every step), written to the cache rows of the tile
template <int W>
__device__ __forceinline__ void plan_fill_cache(PlanArgsC &ap, float *lds, long long i0, unsigned tid) {
    NP_REREAD_ARGS(ap);
    const PlanArgsC a = ap;
    constexpr int TILE = PLAN_TILE;
    const int t = (int)(tid % TILE);
    const int part = __builtin_amdgcn_readfirstlane((int)(tid / TILE));
    float *coef = lds + t;
    const long long n = a->k.n;
    const long long i = i0 + (t & (PLAN_ROWS - 1));
    const bool valid = t < PLAN_ROWS && i < n;
    const long long ic = i < n ? i : n - 1;
    const unsigned o4 = (unsigned)ic * 4u;
    float s[12], u[4];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = at_off(a->k.s + k * a->k.ld, o4);
#pragma unroll
    for (int k = 0; k < 4; k++) u[k] = at_off(a->k.u + k * a->k.ld, o4);
    StateScalars sc1;
    float xd[12];
    const AeroWeights wt1 = {a->k.wt.kblob, a->k.wt.kblob_dual, a->k.wt.pwl, a->k.wt.pwl_unnorm};
    nlplant<false, AB_FORCE, TILE, W, true, 1>(wt1, s, u, sc1, coef, false, xd, part);
    NP_REREAD_ARGS(ap);
    if (valid && part == PLAN_STATE_WAVE) {
        const unsigned iw = (unsigned)i;
        float *cache_w = ap->k.cache + ((long long)(iw >> 6) * NUM_CACHE_ROWS) * CACHE_TILE + (iw & (CACHE_TILE - 1));
#pragma unroll
        for (int k = 0; k < NUM_CACHED; k++) cache_w[k * CACHE_TILE] = coef[cached_slot(k) * TILE];
    }
    __syncthreads();  // the cache rows are written (same workgroup reads them back) and the columns are free
}

template <int TASK, int W, bool QUEUE>
__global__ __launch_bounds__(64 * W, 2) void planning_persistent_kernel(const PlanArgs a) {
    static_assert(W == 4 || W == 8, "four or eight waves per tile");
    __shared__ __attribute__((aligned(16))) float lds_all[npact::ACTOR32_LDS_FLOATS + PLAN_FDM_LDS];
    __shared__ unsigned item_s;
    float *lds_act = lds_all, *lds_fdm = lds_all + npact::ACTOR32_LDS_FLOATS;
    PlanArgsC ap = (PlanArgsC)__builtin_amdgcn_kernarg_segment_ptr();  // `a` is the only kernel parameter
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / 64));

    auto run_item = [&](long long tile, int it, bool first_of_tile) {
        const long long i0 = tile * PLAN_ROWS;
        // per-thread indices and LDS addresses are recomputed from this opaque copy in every iteration: kept across the loop they
        // are ~25 registers the allocator parks in scratch
        unsigned tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        NP_REREAD_ARGS(ap);
        if (first_of_tile && !ap->cache_valid0) plan_fill_cache<W>(ap, lds_fdm, i0, tid);
        // ---- controller (ppo_actor.py:38-64): ll_obs[ia], rnn[ia] -> ll_act, rnn[ib] ----
        if (W == 4 || wave < 4) {
            NP_REREAD_ARGS(ap);
            const int ia = it & 1, ib = ia ^ 1;
            npact::actor_tile32(lds_act, ap->actor_w, ap->k.n, ap->ll_obs[ia], ap->rnn[ia], ap->masks, ap->ll_act, ap->rnn[ib], tile, tid);
        } else {
#pragma unroll 1
            for (int b = 0; b < npact::ACTOR32_BARRIERS; b++) __builtin_amdgcn_s_barrier();
        }
        __syncthreads();  // the tile's actions are written
        NP_REREAD_ARGS(ap);
        plan_fdm_step<TASK, W>(ap, lds_fdm, i0, it, it == ap->iterations - 1, tid);
        __syncthreads();  // the tile's next observation / state are written
    };

    if (!QUEUE) {
        const long long tile = blockIdx.x;
#pragma nounroll
        for (int it = 0;; it++) {
            NP_REREAD_ARGS(ap);
            if (it >= ap->iterations) break;
            run_item(tile, it, it == 0);
        }
    } else {
#pragma nounroll
        for (;;) {
            NP_REREAD_ARGS(ap);
            if (threadIdx.x == 0) item_s = __hip_atomic_fetch_add(ap->queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const unsigned id = item_s;
            const unsigned tiles = (unsigned)ap->tiles;
            if (id >= tiles * (unsigned)ap->iterations) break;
            const int it = (int)(id / tiles);
            const long long tile = (long long)(id % tiles);
            if (it > 0) {
                if (threadIdx.x == 0) {
                    while (__hip_atomic_load(ap->queue + 1 + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it) __builtin_amdgcn_s_sleep(16);
                }
                __syncthreads();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // every wave: what the previous owner of the tile published is visible
            }
            run_item(tile, it, it == 0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // every wave: its stores of this item are out
            __syncthreads();
            NP_REREAD_ARGS(ap);
            if (threadIdx.x == 0) __hip_atomic_store(ap->queue + 1 + tile, (unsigned)(it + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

namespace {
template <int TASK, int W, bool QUEUE>
hipError_t launch_one(const PlanArgs &args, unsigned grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    if (e0 && e1) hipExtLaunchKernelGGL((planning_persistent_kernel<TASK, W, QUEUE>), dim3(grid), dim3(64 * W), 0, st, e0, e1, 0, args);
    else hipLaunchKernelGGL((planning_persistent_kernel<TASK, W, QUEUE>), dim3(grid), dim3(64 * W), 0, st, args);
    return hipGetLastError();
}
template <int TASK, int W>
int occupancy_of() {
    int blocks = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, planning_persistent_kernel<TASK, W, true>, 64 * W, 0) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return blocks;
}
}  // namespace

#ifndef NP_PLAN_TASKS
#define NP_PLAN_TASKS 7  // bit t: build the kernels of task t
#endif

hipError_t launch_planning_persistent(int task, int waves, const PlanArgs &args, unsigned grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    const bool queue = args.queue != nullptr;
#define NP_PLAN_CASE(T)                                                                             \
    if constexpr (((NP_PLAN_TASKS >> T) & 1) != 0) {                                                  \
        if (task == T) {                                                                             \
            if (waves == 8) return queue ? launch_one<T, 8, true>(args, grid, st, e0, e1) : launch_one<T, 8, false>(args, grid, st, e0, e1); \
            return queue ? launch_one<T, 4, true>(args, grid, st, e0, e1) : launch_one<T, 4, false>(args, grid, st, e0, e1);                \
        }                                                                                            \
    }
    NP_PLAN_CASE(0)
    NP_PLAN_CASE(1)
    NP_PLAN_CASE(2)
#undef NP_PLAN_CASE
    return hipErrorInvalidValue;
}

int planning_persistent_workgroups_per_cu(int task, int waves) {
#define NP_PLAN_CASE(T)                                                       \
    if constexpr (((NP_PLAN_TASKS >> T) & 1) != 0) {                        \
        if (task == T) return waves == 8 ? occupancy_of<T, 8>() : occupancy_of<T, 4>(); \
    }
    NP_PLAN_CASE(0)
    NP_PLAN_CASE(1)
    NP_PLAN_CASE(2)
#undef NP_PLAN_CASE
    return 0;
}

}  // namespace npf16
