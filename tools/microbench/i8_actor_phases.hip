// Where the block-fixed-point controller call spends its cycles: the shipped body (np_actor_i8.h) with shader-clock stamps at its phase
// boundaries, one four-wave workgroup per CU on every CU at once (the persistent PlanningEnv kernel's situation), random weights.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -mllvm -disable-machine-licm -I../../neuralplane_amd/csrc tools/microbench/i8_actor_phases.hip -o tools/microbench/i8_actor_phases
#define NPACT_TRACE 1
#include "../../neuralplane_amd/csrc/np_actor_i8.hip"
#include <cstdio>
#include <random>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8192;
    std::mt19937 rng(3);
    std::normal_distribution<float> nd(0.0f, 0.08f);
    std::vector<float> w(npact::TOTAL), wi(npact8::TOTAL_I8), obs((size_t)n * 22), h((size_t)n * 128), m(n, 1.0f);
    for (auto &v : w) v = nd(rng);
    for (auto &v : obs) v = nd(rng) * 10.0f;
    for (auto &v : h) v = nd(rng) * 5.0f;
    if (np_actor_pack_i8(w.data(), wi.data())) return 1;
    float *dw, *dobs, *dh, *dm, *dact, *dh2;
    CHECK(hipMalloc(&dw, wi.size() * 4)); CHECK(hipMalloc(&dobs, obs.size() * 4)); CHECK(hipMalloc(&dh, h.size() * 4)); CHECK(hipMalloc(&dm, m.size() * 4));
    CHECK(hipMalloc(&dact, (size_t)n * 16)); CHECK(hipMalloc(&dh2, h.size() * 4));
    CHECK(hipMemcpy(dw, wi.data(), wi.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dobs, obs.data(), obs.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dh, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dm, m.data(), m.size() * 4, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 20; rep++) CHECK(npact8::launch_actor_i8(dw, n, dobs, dh, dm, dact, dh2, 0));
    CHECK(hipDeviceSynchronize());
    long long t[64];
    CHECK(hipMemcpyFromSymbol(t, HIP_SYMBOL(npact::npact_trace), sizeof(t)));
    const char *names[19] = {"", "obs LayerNorm + quantise", "L1 (1 k-step) + epilogue", "LN1 + quantise + barrier", "L2 + epilogue", "LN2 + 2 x quantise + barrier", "GI r", "GH r", "sigmoid r",
                             "GI z + GH z", "sigmoid z", "GI n + GH n", "tanh + blend", "LN3 + quantise + barrier", "A1 + epilogue", "LN4 + quantise + barrier", "A2 + epilogue", "LN5", "head"};
    for (int k = 1; k <= 18; k++) printf("%-32s %7lld cycles\n", names[k], t[k] - t[k - 1]);
    printf("%-32s %7lld cycles (n = %d: %d workgroups)\n", "the call", t[18] - t[0], n, (n + 31) / 32);
    return 0;
}
