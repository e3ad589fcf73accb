// Can a wave's VALU work proceed while ANOTHER wave on the same SIMD runs a chain of fp32 MFMAs?  One workgroup of 8 waves per CU
// (waves w and w+4 share SIMD w): waves 0..3 run role A (NA dependent / independent / nop-spaced v_mfma_f32_32x32x1_2b_f32),
// waves 4..7 run role B (NB dependent v_fma_f32).  Prints the cycles each role takes alone and together.
// hipcc --offload-arch=gfx950 -O2 mfma_coissue.hip -o mfma_coissue && ./mfma_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x32 __attribute__((ext_vector_type(32)));
constexpr int NA = 2048, NB = 16384;

template <int MODE>  // 0: dependent chain, 1: two independent accumulators alternating, 2: dependent with 60 cycles of s_nop between
__device__ void role_a(float &sink) {
    f32x32 c0 = {}, c1 = {};
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
    for (int i = 0; i < NA / 2; i++) {
        if (MODE == 0) {
            asm volatile("v_mfma_f32_32x32x1_2b_f32 %0, %1, %2, %0\n\tv_mfma_f32_32x32x1_2b_f32 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
        } else if (MODE == 1) {
            asm volatile("v_mfma_f32_32x32x1_2b_f32 %0, %2, %3, %0\n\tv_mfma_f32_32x32x1_2b_f32 %1, %2, %3, %1" : "+v"(c0), "+v"(c1) : "v"(a), "v"(b));
        } else {
            asm volatile("v_mfma_f32_32x32x1_2b_f32 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 11\n\t"
                         "v_mfma_f32_32x32x1_2b_f32 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 11" : "+v"(c0) : "v"(a), "v"(b));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    sink = c0[0] + c1[5];
}
__device__ void role_b(float &sink) {
    float x = threadIdx.x * 1e-3f, y = 1.0001f, z = 0.5f;
    for (int i = 0; i < NB / 8; i++)
        asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\t"
                     "v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
    sink = x;
}
template <int MODE>
__global__ __launch_bounds__(512) void k(int run_a, int run_b, long long *cyc, float *out) {
    const int wave = threadIdx.x / 64;
    float sink = 0;
    __syncthreads();
    const long long t0 = clock64();
    if (wave < 4) { if (run_a) role_a<MODE>(sink); } else { if (run_b) role_b(sink); }
    const long long t1 = clock64();
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    if (sink == 12345.678f) out[0] = sink;
}
template <int MODE>
void run(const char *name, long long *dc, float *dout) {
    std::vector<long long> h(8 * 256);
    const int cases[3][2] = {{1, 0}, {0, 1}, {1, 1}};
    printf("%s\n", name);
    for (auto &c : cases) {
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, c[0], c[1], dc, dout);
        hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
        double a = 0, b = 0;
        for (int w = 0; w < 256; w++) for (int j = 0; j < 8; j++) (j < 4 ? a : b) += (double)h[w * 8 + j] / (256 * 4);
        printf("  A %s B %s:  A %8.0f ticks (%.1f per MFMA)   B %8.0f ticks (%.2f per v_fma)\n", c[0] ? "on " : "off", c[1] ? "on " : "off", a, a / NA, b, b / NB);
    }
}
int main() {
    long long *dc; float *dout;
    hipMalloc(&dc, 8 * 256 * 8); hipMalloc(&dout, 4);
    run<0>("dependent MFMA chain", dc, dout);
    run<1>("two independent accumulators", dc, dout);
    run<2>("dependent chain, s_nop between", dc, dout);
    printf("(clock64 ticks at the constant 100 MHz counter: 1 tick = ~24 shader cycles at 2.4 GHz)\n");
    return 0;
}
