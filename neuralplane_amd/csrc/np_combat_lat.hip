// np_combat_lat.hip — SingleCombatEnv.step for small batches (reference: envs/singlecombat_env.py:240-274; BASELINE config 5's share of one
// GPU of eight, 12 500 engagements, and of four, 25 000, are in this range): f16_combat_kernel<.., 128, WPT_DUAL8 / WPT_DUAL4> — eight
// (four) waves share a tile of 128 aircraft (64 engagements) and wave w evaluates an eighth (a quarter) of the nets of every aero
// evaluation for both 64-row halves with the two-set class bodies (np_f16_device.h::eval_nets): half the workgroups — half the CUs that pull
// the 55 KB weight stream through their scalar caches per inner step — of the four-waves-per-64-aircraft kernel for the same batch.
// A translation unit of its own: two kernels, built beside the other two.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "../../include/neuralplane_amd.h"
#include "np_f16_combat.h"

namespace npf16 {

void launch_combat_dual(const CombatArgs &a, int waves, unsigned grid, hipStream_t st, bool timed, hipEvent_t start, hipEvent_t stop) {
    const dim3 g(grid), b(64 * (unsigned)waves);
#define NP_DUAL_LAUNCH(W)                                                                                                          \
    do {                                                                                                                           \
        if (timed) hipExtLaunchKernelGGL((f16_combat_kernel<0, true, COMBAT_DUAL_TILE, W>), g, b, 0, st, start, stop, 0, a);       \
        else hipLaunchKernelGGL((f16_combat_kernel<0, true, COMBAT_DUAL_TILE, W>), g, b, 0, st, a);                                \
    } while (0)
    if (waves == 8) NP_DUAL_LAUNCH(WPT_DUAL8);
    else NP_DUAL_LAUNCH(WPT_DUAL4);
#undef NP_DUAL_LAUNCH
}

}  // namespace npf16
