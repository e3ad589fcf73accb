// Issue rate of the matrix instructions a controller could use (one wave per SIMD, four independent accumulators, no memory traffic):
// cycles per instruction and operations per cycle and SIMD.   hipcc --offload-arch=gfx950 -O3 mfma_rates.hip -o mfma_rates && ./mfma_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(256) void rate(long long *cyc, float *sink, int iters) {
    const int l = threadIdx.x;
    long long t0, t1;
    if constexpr (KIND == 0) {   // v_mfma_i32_16x16x64_i8
        i32x4 a = {l, l + 1, l + 2, l + 3}, b = {l * 3, l * 5, l * 7, l * 9}, c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; i++) {
            c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
        }
        t1 = __builtin_readcyclecounter();
        sink[blockIdx.x * 256 + l] = (float)(c0[0] + c1[1] + c2[2] + c3[3]);
    } else if constexpr (KIND == 1) {   // v_mfma_f32_16x16x32_bf16
        bf16x8 a, b; for (int e = 0; e < 8; e++) { a[e] = (__bf16)(float)(l + e); b[e] = (__bf16)(float)(l - e); }
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; i++) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
        }
        t1 = __builtin_readcyclecounter();
        sink[blockIdx.x * 256 + l] = c0[0] + c1[1] + c2[2] + c3[3];
    } else if constexpr (KIND == 2) {   // v_mfma_f32_16x16x4_f32 (the fp32 pipe: 2 048 FLOP per instruction)
        float a = (float)l, b = (float)(l + 1);
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; i++) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
        }
        t1 = __builtin_readcyclecounter();
        sink[blockIdx.x * 256 + l] = c0[0] + c1[1] + c2[2] + c3[3];
    } else {   // v_mfma_i32_32x32x32_i8
        typedef int i32x16 __attribute__((ext_vector_type(16)));
        i32x4 a = {l, l + 1, l + 2, l + 3}, b = {l * 3, l * 5, l * 7, l * 9};
        i32x16 c0 = {}, c1 = {};
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; i++) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
        }
        t1 = __builtin_readcyclecounter();
        sink[blockIdx.x * 256 + l] = (float)(c0[0] + c1[1]);
    }
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    long long *dc; float *ds;
    const int blocks = 256, iters = 20000;
    CHECK(hipMalloc(&dc, blocks * 8)); CHECK(hipMalloc(&ds, blocks * 256 * 4));
    const char *names[4] = {"v_mfma_i32_16x16x64_i8", "v_mfma_f32_16x16x32_bf16", "v_mfma_f32_16x16x4_f32", "v_mfma_i32_32x32x32_i8"};
    const double ops[4] = {2.0 * 16 * 16 * 64, 2.0 * 16 * 16 * 32, 2.0 * 16 * 16 * 4, 2.0 * 32 * 32 * 32};
    for (int k = 0; k < 4; k++) {
        for (int rep = 0; rep < 2; rep++) {
            if (k == 0) hipLaunchKernelGGL(rate<0>, dim3(blocks), dim3(256), 0, 0, dc, ds, iters);
            if (k == 1) hipLaunchKernelGGL(rate<1>, dim3(blocks), dim3(256), 0, 0, dc, ds, iters);
            if (k == 2) hipLaunchKernelGGL(rate<2>, dim3(blocks), dim3(256), 0, 0, dc, ds, iters);
            if (k == 3) hipLaunchKernelGGL(rate<3>, dim3(blocks), dim3(256), 0, 0, dc, ds, iters);
            CHECK(hipDeviceSynchronize());
        }
        long long c; CHECK(hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost));
        const double per = (double)c / (4.0 * iters);
        printf("%-28s %6.1f cycles per instruction (one wave per SIMD, 4 accumulators) = %7.0f ops per cycle and SIMD = %5.1f x the fp32 matrix rate\n", names[k], per, ops[k] / per,
               ops[k] / per / 64.0);
    }
    return 0;
}
