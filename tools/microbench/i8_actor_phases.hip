// What the block-fixed-point controller call (np_actor_i8.h) costs and where: the shipped kernel timed with HIP events over back-to-back
// launches, one four-wave workgroup per CU (n = 8 192: the persistent PlanningEnv kernel's situation), built several times with parts
// REMOVED (timing-only switches, wrong results: -DNPACT8_EXP=1 no matrix instructions, 2 no weight stream, 16 a one-instruction epilogue,
// -DNPACT_EXP=8 free gate nonlinearities) — the differences are the parts' costs under the real overlap, which clock stamps inside the
// kernel are not (a stamp orders the memory operations around it).  tools/microbench/i8_actor_phases.sh builds and runs the set.
#include "../../neuralplane_amd/csrc/np_actor_i8.hip"
#include <cstdio>
int np_internal_fail(const char *msg) { fprintf(stderr, "%s\n", msg); return 1; }   // (the library's error string lives in np_f16_kernels.hip)
#include <random>
#include <vector>
#ifndef TILES
#define TILES 0
#endif
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
    for (int rep = 0; rep < 300; rep++) CHECK(npact8::launch_actor_i8(dw, n, dobs, dh, dm, dact, dh2, 0));
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int K = 500;
    CHECK(hipEventRecord(e0, 0));
    for (int rep = 0; rep < K; rep++) CHECK(npact8::launch_actor_i8(dw, n, dobs, dh, dm, dact, dh2, 0));
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("NPACT8_EXP=%d NPACT_EXP=%d n=%d: %.2f us per launch (%d back-to-back launches)\n", NPACT8_EXP, NPACT_EXP, n, 1e3 * ms / K, K);
    return 0;
}
