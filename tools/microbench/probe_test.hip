// Does s_atc_probe (gfx9 "probe or prefetch an address into the SQC data cache") warm the scalar cache?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void chase(const int* __restrict__ ring, int steps, long long* out, int* sink) {
    int idx = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < steps; i++) {
        // the ring is sequential with stride 64 B: the line 8 entries ahead is known
        const int* ahead = ring + ((idx + 8 * 16) & (64 * 1024 / 4 * 16 - 1));
        if (MODE == 1) asm volatile("s_atc_probe 0x0, %0, 0x0" ::"s"(ahead));
        if (MODE == 2) asm volatile("s_dcache_discard %0, 0x0" ::"s"(ahead));
        // ~200 cycles of unrelated scalar work so that a prefetch has time to land
        int acc = idx;
#pragma unroll
        for (int k = 0; k < 60; k++) asm volatile("s_add_u32 %0, %0, 1\n s_nop 1" : "+s"(acc));
        idx = ring[idx + (acc - acc)];
        idx = __builtin_amdgcn_readfirstlane(idx);
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; sink[blockIdx.x] = idx; }
}

int main() {
    int kb = 1024;
    int n = kb * 1024 / 4, stride = 16, cnt = n / stride;
    std::vector<int> h(n, 0);
    for (int i = 0; i < cnt; i++) h[i * stride] = ((i + 1) % cnt) * stride;
    int* d; long long* out; int* sink;
    CHECK(hipMalloc(&d, n * 4)); CHECK(hipMalloc(&out, 64)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    int steps = 4096;
    long long c;
    hipLaunchKernelGGL(chase<0>, dim3(1), dim3(64), 0, 0, d, steps, out, sink); CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(chase<0>, dim3(1), dim3(64), 0, 0, d, steps, out, sink); CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost)); printf("no prefetch    : %.1f cycles/iter\n", (double)c / steps);
    hipLaunchKernelGGL(chase<1>, dim3(1), dim3(64), 0, 0, d, steps, out, sink); CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost)); printf("s_atc_probe    : %.1f cycles/iter\n", (double)c / steps);
    return 0;
}
