// Micro-benchmark: scalar-load (s_load) latency on gfx950 by pointer chasing through a ring whose
// footprint either fits the scalar cache or not.  hipcc --offload-arch=gfx950 -O3 smem_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void chase(const int* __restrict__ ring, int steps, long long* out, int* sink) {
    int idx = 0;
    // warm
    for (int i = 0; i < 64; i++) idx = __builtin_amdgcn_readfirstlane(ring[idx]);
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < steps; i++) {
        idx = ring[idx];                      // uniform address -> s_load_dword, dependent chain
        idx = __builtin_amdgcn_readfirstlane(idx);
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; sink[blockIdx.x] = idx; }
}

int main() {
    for (int kb : {1, 4, 8, 16, 32, 64, 256, 4096}) {
        int n = kb * 1024 / 4;
        int stride = 16;  // 64 B
        std::vector<int> h(n, 0);
        int cnt = n / stride;
        for (int i = 0; i < cnt; i++) h[i * stride] = ((i + 1) % cnt) * stride;
        int* d; long long* out; int* sink;
        CHECK(hipMalloc(&d, n * 4)); CHECK(hipMalloc(&out, 8 * 1024)); CHECK(hipMalloc(&sink, 4 * 1024));
        CHECK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
        int steps = 4096;
        for (int blocks : {1, 1024}) {
            hipLaunchKernelGGL(chase, dim3(blocks), dim3(64), 0, 0, d, steps, out, sink);
            CHECK(hipDeviceSynchronize());
            hipLaunchKernelGGL(chase, dim3(blocks), dim3(64), 0, 0, d, steps, out, sink);
            CHECK(hipDeviceSynchronize());
            long long c; CHECK(hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost));
            printf("footprint %5d KB, %4d blocks: %.1f cycles per dependent s_load (incl. readfirstlane + loop)\n", kb, blocks, (double)c / steps);
        }
        hipFree(d); hipFree(out); hipFree(sink);
    }
    return 0;
}
