// np_planning.h — the persistent PlanningEnv kernel's argument record and its launcher (np_planning.hip), as np_f16_kernels.hip's
// np_planning_inner_loop sees them.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "np_f16_kargs.h"

namespace npf16 {

// One launch = all `iterations` low-level iterations of PlanningEnv.step (reference envs/planning_env.py:153-176) for every 32-row
// tile.  `k` must stay the first member: the FDM device code re-reads its scalars from offset 0 of the kernel-argument segment.
struct PlanArgs {
    KArgs k;                 // what all iterations share (np_f16_io); fin* / fout* / call_idx are per iteration and derived below
    const float *actor_w;    // np_actor_forward's packed weights
    float *ll_obs[2];        // [n][22] ping-pong; [0] = the first iteration's input
    float *rnn[2];           // [n][128] ping-pong; [0] = the recurrent state on entry
    const float *masks;      // [n]
    float *ll_act;           // [n][4]
    uint8_t *flags[2];       // [3][n] ping-pong; [0] = the flags on entry
    float *final_obs;        // the task observation of the LAST iteration (may be null)
    int iterations;
    int cache_valid0;        // is k.cache valid for the first iteration?
    long long tiles;         // ceil(n / 32)
    int block;               // queue schedule: iterations per item (a tile stays on its workgroup for a block)
    // (tile, block of iterations) work queue (tiles > resident workgroups): item id = block index * tiles + tile, handed out in order
    unsigned *queue;         // [0] next item id + queue_base; [1 + tile] flag_base + iterations of that tile that are complete
    unsigned queue_base, flag_base;   // the words are not cleared between launches: the launcher moves the bases past what the last launch left
    int guest_blocks;        // 0: the dynamic queue above.  B > 0: the static guest schedule (np_planning.hip) — grid = resident workgroups C,
                             // the tiles - C guest tiles are cut into B blocks each, hosted by workgroups 0 .. (tiles - C) * B - 1; `block` = slack
    // bounded waits (round 5): a progress-word wait that lasts longer than wait_ticks gives up — the first such wait claims err[0] (device
    // memory, sticky for the launch: every other wait sees it and gives up too, every workgroup drains without raising another word) and
    // writes the record the launcher reads after the launch into err_host (pinned host memory)
    unsigned *err;                  // [0] 0 = none, else 1 + the workgroup whose wait expired
    unsigned *err_host;             // PLAN_ERR_WORDS words: code (= err[0]), tile, iteration waited for, iterations published, ms waited
    unsigned long long wait_ticks;  // bound of ONE wait in wall_clock64() ticks (100 MHz)
    int debug_stall;                // tests only: workgroup 0 never raises a progress word (its successors' waits expire)
};
constexpr int PLAN_ERR_WORDS = 8;

constexpr int PLAN_ROWS = 32;  // rows per tile: one 32-row controller tile (np_actor.h)

// tasks: 0 heading, 1 control, 2 tracking (PlanningEnv is a tracking env: only task 2 is built, NP_PLAN_TASKS).
// waves: 8 per workgroup (the four-wave builds were retired in round 5).  queue_mode: 0 = one workgroup per tile (grid = tiles), 1 = `grid` persistent workgroups pulling
// (tile, iteration) items.  Returns hipSuccess or the launch error.
hipError_t launch_planning_persistent(int task, int waves, bool i8, const PlanArgs &args, unsigned grid, hipStream_t stream, hipEvent_t ev_start,
                                      hipEvent_t ev_stop);   // i8: args.actor_w is an NP_ACTOR_I8_NUM_FLOATS buffer (block-fixed-point controller)
// dual workgroups: two 32-row tiles per eight-wave workgroup (np_planning.hip), static schedule, grid = ceil(tiles / 2)
hipError_t launch_planning_dual(int task, const PlanArgs &args, unsigned grid, hipStream_t stream);
// are the persistent kernels of this task (0 heading, 1 control, 2 tracking) part of the build?  (tracking: the reference's PlanningEnv task)
bool planning_persistent_built(int task);
// how many workgroups of that shape fit one CU / the device (occupancy query)
int planning_persistent_workgroups_per_cu(int task, int waves, bool i8);   // i8: the block-fixed-point controller's instantiation (its own dynamic LDS)

}  // namespace npf16
