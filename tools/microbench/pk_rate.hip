// pk_rate.hip — issue rate of v_pk_fma_f32 by operand form on gfx950 (round 3): is the 130 TFLOP/s ceiling of the net bodies' form
// (SGPR pair in src0) a property of the scalar operand, of its position, or of the op_sel broadcast?
//   A  v_pk_fma_f32 acc, s[2], x, acc                 SGPR pair in src0 (the single-set bodies)
//   B  v_pk_fma_f32 acc, x, s[2], acc                 SGPR pair in src1
//   C  v_pk_fma_f32 acc, w(VGPR pair), x, acc         all VGPR
//   D  v_pk_fma_f32 acc, s[2], x, acc op_sel_hi:[0,1,1]   one scalar broadcast to both halves (the two-set, neuron-major bodies)
//   E  v_fma_f32    acc, s, x, acc                    unpacked, SGPR src0
// 16 independent accumulators per wave, W waves per SIMD; prints TFLOP/s for the whole chip and cycles per instruction.
// Build: hipcc --offload-arch=gfx950 -O3 -o pk_rate pk_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 4000, NACC = 16;

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, const float *wsrc, long long *cyc) {
    f32x2 acc[NACC], x = {out[threadIdx.x], out[threadIdx.x + 1]}, wv = {wsrc[threadIdx.x & 3], wsrc[4]};
#pragma unroll
    for (int j = 0; j < NACC; j++) acc[j] = f32x2{(float)j, x[0]};
    const int i0 = __builtin_amdgcn_readfirstlane(__float_as_int(wsrc[5])), i1 = __builtin_amdgcn_readfirstlane(__float_as_int(wsrc[6]));
    const long long sp = ((long long)i1 << 32) | (unsigned)i0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int j = 0; j < NACC; j++) {
            if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "s"(sp), "v"(x));
            else if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %2, %1, %0" : "+v"(acc[j]) : "s"(sp), "v"(x));
            else if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(wv), "v"(x));
            else if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[j]) : "s"(sp), "v"(x));
            else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[j][0]) : "s"(i0), "v"(x[0]));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < NACC; j++) s += acc[j][0] + acc[j][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, float *d_out, float *d_w, long long *d_cyc) {
    for (int wps = 1; wps <= 3; wps++) {  // waves per SIMD: 256 CUs x wps workgroups of 4 waves
        const int grid = 256 * wps;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d_out, d_w, d_cyc);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d_out, d_w, d_cyc);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        long long c;
        CHECK(hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost));
        const double flop = (double)grid * 256 * ITERS * NACC * (MODE == 4 ? 2.0 : 4.0);
        printf("%-58s %d wave(s)/SIMD: %7.1f TFLOP/s, %.2f cycles per instruction (wave 0)\n", name, wps, flop / (ms * 1e-3) / 1e12, (double)c / (ITERS * NACC));
    }
}

int main() {
    float *d_out, *d_w;
    long long *d_cyc;
    CHECK(hipMalloc(&d_out, 4 * 256 * 1024 + 64));
    CHECK(hipMalloc(&d_w, 64));
    CHECK(hipMalloc(&d_cyc, 8 * 1024));
    CHECK(hipMemset(d_out, 0, 4 * 256 * 1024 + 64));
    float w[16] = {0.5f, 0.25f, 0.125f, 1.0f, 0.75f, 0.999f, 1.001f, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    CHECK(hipMemcpy(d_w, w, 64, hipMemcpyHostToDevice));
    run<0>("A v_pk_fma_f32 acc, s[2], x, acc", d_out, d_w, d_cyc);
    run<1>("B v_pk_fma_f32 acc, x, s[2], acc", d_out, d_w, d_cyc);
    run<2>("C v_pk_fma_f32 acc, v[2], x, acc", d_out, d_w, d_cyc);
    run<3>("D v_pk_fma_f32 acc, s[2], x, acc op_sel_hi:[0,1,1]", d_out, d_w, d_cyc);
    run<4>("E v_fma_f32 acc, s, x, acc", d_out, d_w, d_cyc);
    return 0;
}
