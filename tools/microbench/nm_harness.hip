// The two-set (neuron-major) statements of np_mlp_asm_dual.inc on the hardware, outside the env kernel: a synthetic weight blob,
// inputs and expected outputs come from the CPU emulation of the same instruction text (python tools/emulate_dual_asm.py --dump
// tools/microbench/nm_harness.bin).  One workgroup of two waves, as in the pair kernel: wave 0 / wave 1 run the two halves of a
// phase concurrently (mode 0: ALL, 1: REST, 2: FORCE2 — completion only for 1 and 2), mode 3 runs them one after the other.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ../../neuralplane_amd/csrc nm_harness.hip -o nm_harness && ./nm_harness nm_harness.bin 0
// (History: this harness is where a record pointer handed over in VGPRs + v_readfirstlane was found to fault when both waves
//  streamed from the same buffer; the statements take the pointer as their one scalar operand since.)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "np_nets.h"
namespace npf16 {
#include "np_mlp_asm_dual.inc"
constexpr int LD = 128, COLS = NUM_LDS_SLOTS + NUM_NORM_GROUPS;
__global__ __launch_bounds__(128, 2) void harness(const float *blob, const float *xa, const float *xb, float *out, int mode) {
    __shared__ float lds[COLS * LD];
    const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t / 64);   // wave-uniform: the statements contain scalar code
    float *col = lds + t;                      // this lane's column (set A); set B = the same lane of the other wave
    float *col_b = col + 64 - 128 * wave;
    for (int g = 0; g < NUM_NORM_GROUPS; g++) col[(NUM_LIVE_NETS + g) * LD] = (wave == 0 ? xa : xb)[g];
    __syncthreads();
    const unsigned a = (unsigned)(unsigned long long)col, b = (unsigned)(unsigned long long)col_b;
    if (mode == 0 || mode == 3) {
        if (wave == 0) mlp_phase_asm_dual_ALL_0<LD * 4>(blob + MLP_PAIR_ALL_0_START, a, b);
        if (mode == 3) __syncthreads();
        if (wave == 1) mlp_phase_asm_dual_ALL_1<LD * 4>(blob + MLP_PAIR_ALL_1_START, a, b);
    } else if (mode == 1) {
        if (wave == 0) mlp_phase_asm_dual_REST_0<LD * 4>(blob + MLP_PAIR_REST_0_START, a, b);
        else mlp_phase_asm_dual_REST_1<LD * 4>(blob + MLP_PAIR_REST_1_START, a, b);
    } else {
        if (wave == 0) mlp_phase_asm_dual_FORCE2_0<LD * 4>(blob + MLP_PAIR_FORCE2_0_START, a, b);
        else mlp_phase_asm_dual_FORCE2_1<LD * 4>(blob + MLP_PAIR_FORCE2_1_START, a, b);
    }
    __syncthreads();
    if ((t & 63) == 0)      // wave 0's lanes hold the coefficients for the inputs xa, wave 1's for xb
        for (int s = 0; s < NUM_LIVE_NETS; s++) out[wave * NUM_LIVE_NETS + s] = col[s * LD];
}
}  // namespace npf16
#define CHECK(x) do { if ((x) != hipSuccess) { printf("%s failed\n", #x); return 2; } } while (0)
int main(int argc, char **argv) {
    FILE *f = fopen(argc > 1 ? argv[1] : "nm_harness.bin", "rb");
    if (!f) return 1;
    int n;
    if (fread(&n, 4, 1, f) != 1) return 1;
    std::vector<float> blob(n), xa(9), xb(9), want(84), got(84);
    if (fread(blob.data(), 4, n, f) != (size_t)n || fread(xa.data(), 4, 9, f) != 9 || fread(xb.data(), 4, 9, f) != 9 || fread(want.data(), 4, 84, f) != 84) return 1;
    const int mode = argc > 2 ? atoi(argv[2]) : 0;
    float *d_blob, *d_xa, *d_xb, *d_out;
    CHECK(hipMalloc(&d_blob, 4 * n)); CHECK(hipMalloc(&d_xa, 36)); CHECK(hipMalloc(&d_xb, 36)); CHECK(hipMalloc(&d_out, 4 * 84));
    CHECK(hipMemcpy(d_blob, blob.data(), 4 * n, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_xa, xa.data(), 36, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_xb, xb.data(), 36, hipMemcpyHostToDevice));
    int total_bad = 0;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipMemset(d_out, 0, 4 * 84));
        hipLaunchKernelGGL(npf16::harness, dim3(1), dim3(128), 0, 0, d_blob, d_xa, d_xb, d_out, mode);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(got.data(), d_out, 4 * 84, hipMemcpyDeviceToHost));
        if (mode != 0 && mode != 3) { printf("rep %d: ran to completion\n", rep); continue; }
        int bad = 0;
        for (int k = 0; k < 84; k++)
            if (!(std::fabs(got[k] - want[k]) <= 1e-4f * std::fmax(1.0f, std::fabs(want[k])))) {
                if (bad < 8) printf("  set %d slot %2d got %g want %g\n", k / 42, k % 42, got[k], want[k]);
                bad++;
            }
        printf("rep %d: %d of 84 coefficients wrong\n", rep, bad);
        total_bad += bad;
    }
    return total_bad ? 3 : 0;
}
