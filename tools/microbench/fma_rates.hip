// Micro-benchmark: issue rate of the three ways to feed a wave-uniform weight into a per-lane FMA
// on gfx950.  hipcc --offload-arch=gfx950 -O3 fma_rates.hip -o fma_rates && ./fma_rates
//   A: v_fmac_f32 v, s, v           (weight in an SGPR)
//   B: v_pk_fma_f32 v[2], s[2], v[2] (two weights in an SGPR pair, two accumulators packed)
//   C: v_fmac_f32_dpp v, v, v row_newbcast:n (16 weights per VGPR, lane n of each row broadcast)
//   D: v_pk_fma_f32 v[2], s, v[2] op_sel (one SGPR weight applied to two packed values = 2 aircraft/lane)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 2000;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* wsrc, long long* cyc) {
    float x = out[threadIdx.x];
    float acc[10];
#pragma unroll
    for (int j = 0; j < 10; j++) acc[j] = x * j;
    float s0 = wsrc[blockIdx.x & 1], s1 = wsrc[2], s2 = wsrc[3], s3 = wsrc[4];
    s0 = __builtin_amdgcn_readfirstlane(__float_as_int(s0)) ? s0 : s0;
    int i0 = __builtin_amdgcn_readfirstlane(__float_as_int(wsrc[5]));
    int i1 = __builtin_amdgcn_readfirstlane(__float_as_int(wsrc[6]));
    float w = wsrc[threadIdx.x & 15];
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
#pragma unroll
                for (int j = 0; j < 10; j++) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[j]) : "s"(i0), "v"(x));
            }
        } else if (MODE == 1) {
            f32x2* a2 = (f32x2*)acc;
            f32x2 xx = {x, x};
            long long sp = ((long long)i1 << 32) | (unsigned)i0;
#pragma unroll
            for (int r = 0; r < 4; r++) {
#pragma unroll
                for (int j = 0; j < 5; j++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a2[j]) : "s"(sp), "v"(xx));
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
#pragma unroll
                for (int j = 0; j < 10; j++) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[j]) : "v"(w), "v"(x));
            }
        } else if (MODE == 3) {
            f32x2* a2 = (f32x2*)acc;
            f32x2 xx = {x, x * 2};
            long long sp = ((long long)i1 << 32) | (unsigned)i0;
#pragma unroll
            for (int r = 0; r < 4; r++) {
#pragma unroll
                for (int j = 0; j < 5; j++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a2[j]) : "s"(sp), "v"(xx));
            }
        } else if (MODE == 4) {  // plain VGPR-VGPR fmac for reference
#pragma unroll
            for (int r = 0; r < 4; r++) {
#pragma unroll
                for (int j = 0; j < 10; j++) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[j]) : "v"(w), "v"(x));
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float sum = 0;
#pragma unroll
    for (int j = 0; j < 10; j++) sum += acc[j];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
int run(const char* name, int fma_per_inst, int inst_per_iter, int blocks_per_cu) {
    int nblk = 256 * blocks_per_cu;
    float* out; float* w; long long* cyc;
    CHECK(hipMalloc(&out, sizeof(float) * 256 * nblk));
    CHECK(hipMalloc(&w, 64 * 4));
    CHECK(hipMalloc(&cyc, 8));
    std::vector<float> hw(64, 1.0001f);
    CHECK(hipMemcpy(w, hw.data(), 64 * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(out, 0, sizeof(float) * 256 * nblk));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(256), 0, 0, out, w, cyc);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(256), 0, 0, out, w, cyc);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    long long c; CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    double insts = (double)ITERS * inst_per_iter;
    double flops = 2.0 * fma_per_inst * 64.0 * insts * 4.0 * nblk;  // 4 waves per block
    printf("%-44s blocks/CU=%d: %.3f ms, %.1f TFLOP/s, wave0 %.2f cyc/inst (s_memtime ticks)\n", name, blocks_per_cu, ms,
           flops / ms / 1e9, (double)c / insts);
    hipFree(out); hipFree(w); hipFree(cyc);
    return 0;
}

int main() {
    for (int b : {1, 2, 4}) {
        run<0>("A v_fmac_f32 v,s,v", 1, 40, b);
        run<1>("B v_pk_fma_f32 v2,s2,v2", 2, 20, b);
        run<2>("C v_fmac_f32_dpp row_newbcast", 1, 40, b);
        run<3>("D v_pk_fma_f32 v2,s(bcast),v2", 2, 20, b);
        run<4>("E v_fmac_f32 v,v,v", 1, 40, b);
    }
    return 0;
}
