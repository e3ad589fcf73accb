// Does a wave's VALU work proceed while ANOTHER wave on the same SIMD issues v_mfma_i32_32x32x32_i8 back to back?  (For the fp32 MFMA it does
// not: tools/microbench/mfma_coissue.hip — one pipe.)  One workgroup of 8 waves per CU (waves w and w + 4 share SIMD w): waves 0..3 run role A
// (NA i8 MFMAs on four rotating accumulators, as the controller's k-steps issue them), waves 4..7 role B (NB dependent v_fma_f32, or NB
// v_cvt_f32_i32 / v_perm_b32 — the epilogue's and the quantiser's instructions).  Cycles of each role alone and together.
// hipcc --offload-arch=gfx950 -O2 tools/microbench/mfma_i8_coissue.hip -o tools/microbench/mfma_i8_coissue && tools/microbench/mfma_i8_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
constexpr int NA = 4096, NB = 32768;

__device__ void role_a(int &sink) {
    i32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    i32x4 a = {1, 2, 3, (int)threadIdx.x}, b = {5, 6, 7, 8};
    for (int i = 0; i < NA / 4; i++) {
        asm volatile("v_mfma_i32_32x32x32_i8 %0, %4, %5, %0\n\tv_mfma_i32_32x32x32_i8 %1, %4, %5, %1\n\tv_mfma_i32_32x32x32_i8 %2, %4, %5, %2\n\tv_mfma_i32_32x32x32_i8 %3, %4, %5, %3"
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    sink = c0[0] + c1[5] + c2[7] + c3[9];
}
template <int KIND>
__device__ void role_b(float &sink) {
    float x = threadIdx.x * 1e-3f, y = 1.0001f, z = 0.5f;
    int q = threadIdx.x;
    for (int i = 0; i < NB / 8; i++) {
        if (KIND == 0)
            asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\t"
                         "v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
        else
            asm volatile("v_cvt_f32_i32 %0, %1\n\tv_perm_b32 %1, %1, %1, %2\n\tv_cvt_f32_i32 %0, %1\n\tv_perm_b32 %1, %1, %1, %2\n\t"
                         "v_cvt_f32_i32 %0, %1\n\tv_perm_b32 %1, %1, %1, %2\n\tv_cvt_f32_i32 %0, %1\n\tv_perm_b32 %1, %1, %1, %2" : "+v"(x), "+v"(q) : "v"(0x02010003));
    }
    sink = x + q;
}
template <int KIND>
__global__ __launch_bounds__(512) void k(int run_a, int run_b, long long *cyc, float *out) {
    const int wave = threadIdx.x / 64;
    float sink = 0;
    int isink = 0;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    if (wave < 4) { if (run_a) role_a(isink); } else { if (run_b) role_b<KIND>(sink); }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    if (sink == 12345.678f || isink == 123456789) out[0] = sink + isink;
}
template <int KIND>
void run(const char *name, long long *dc, float *dout) {
    std::vector<long long> h(8 * 256);
    const int cases[3][2] = {{1, 0}, {0, 1}, {1, 1}};
    printf("%s\n", name);
    for (auto &c : cases) {
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, c[0], c[1], dc, dout);
        hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
        double a = 0, b = 0;
        for (int w = 0; w < 256; w++) for (int j = 0; j < 8; j++) (j < 4 ? a : b) += (double)h[w * 8 + j] / (256 * 4);
        printf("  A %s B %s:  A %9.0f cycles (%.1f per MFMA)   B %9.0f cycles (%.2f per VALU instruction)\n", c[0] ? "on " : "off", c[1] ? "on " : "off", a, a / NA, b, b / NB);
    }
}
int main() {
    long long *dc; float *dout;
    hipMalloc(&dc, 8 * 256 * 8); hipMalloc(&dout, 4);
    run<0>("i8 MFMA (4 rotating accumulators) beside dependent v_fma_f32", dc, dout);
    run<1>("i8 MFMA beside v_cvt_f32_i32 / v_perm_b32", dc, dout);
    return 0;
}
