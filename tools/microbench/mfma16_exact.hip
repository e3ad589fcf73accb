// v_mfma_f32_16x16x1_4b_f32 (K = 1, four 16 x 16 blocks): is a chain of them the scalar fmaf chain bit for bit, with the layout
//   A: lane l -> block l/16, i = l%16;  B: lane l -> block l/16, j = l%16;
//   D: 16 VGPRs, block b = r/4, lane l -> i = 4*(l/16) + r%4, j = l%16 ?
// Also: the bias as one more MFMA (A = bias, B = 1.0, C = 0), and the issue rate of one dependent chain vs two interleaved chains.
// hipcc --offload-arch=gfx950 -O2 -ffp-contract=off mfma16_exact.hip -o mfma16_exact && ./mfma16_exact
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int K = 128;
__global__ void chain(const float *A, const float *B, const float *C, float *D) {
    const int l = threadIdx.x;
    f32x16 acc;
    for (int r = 0; r < 16; r++) acc[r] = C[r * 64 + l];
    for (int k = 0; k < K; k++) acc = __builtin_amdgcn_mfma_f32_16x16x1f32(A[k * 64 + l], B[k * 64 + l], acc, 0, 0, 0);
    for (int r = 0; r < 16; r++) D[r * 64 + l] = acc[r];
}
__global__ void bias_chain(const float *A, const float *B, const float *bias, float *D) {  // bias[l]: the A operand of the bias step
    const int l = threadIdx.x;
    f32x16 acc;
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    acc = __builtin_amdgcn_mfma_f32_16x16x1f32(bias[l], 1.0f, acc, 0, 0, 0);
    for (int k = 0; k < K; k++) acc = __builtin_amdgcn_mfma_f32_16x16x1f32(A[k * 64 + l], B[k * 64 + l], acc, 0, 0, 0);
    for (int r = 0; r < 16; r++) D[r * 64 + l] = acc[r];
}
template <int CHAINS>
__global__ void rate(float *out, long long *cycles, int iters) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; c++)
        for (int r = 0; r < 16; r++) acc[c][r] = 0.0f;
    const float a = 1.0f + threadIdx.x * 1e-7f, b = 0.5f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int c = 0; c < CHAINS; c++) acc[c] = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, acc[c], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int c = 0; c < CHAINS; c++)
        for (int r = 0; r < 16; r++) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}
static float rnd(int mode) {
    const float u = (float)rand() / (float)RAND_MAX * 2.0f - 1.0f;
    switch (mode) {
        case 0: return u;
        case 1: return u * 1e-20f;
        case 2: return u * 1e18f;
        default: return (rand() & 7) == 0 ? u * 1e-30f : u * (float)(1 << (rand() % 20));
    }
}
int main() {
    std::vector<float> A(K * 64), B(K * 64), C(16 * 64), D(16 * 64), bias(64);
    float *dA, *dB, *dC, *dD, *dbias, *dout; long long *dcyc;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMalloc(&dbias, 256); hipMalloc(&dout, 1024 * 256 * 4); hipMalloc(&dcyc, 8);
    int bad_total = 0;
    for (int mode = 0; mode < 4; mode++) {
        srand(17 + mode);
        for (auto &x : A) x = rnd(mode);
        for (auto &x : B) x = rnd(mode == 2 ? 0 : mode);
        for (auto &x : C) x = rnd(mode);
        for (auto &x : bias) x = rnd(mode);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dbias, bias.data(), 256, hipMemcpyHostToDevice);
        for (int pass = 0; pass < 2; pass++) {
            if (pass == 0) hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
            else hipLaunchKernelGGL(bias_chain, dim3(1), dim3(64), 0, 0, dA, dB, dbias, dD);
            hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int r = 0; r < 16; r++)
                for (int l = 0; l < 64; l++) {
                    const int b = r / 4, i = 4 * (l / 16) + r % 4, j = l % 16;
                    float acc = pass == 0 ? C[r * 64 + l] : bias[16 * b + i];
                    for (int k = 0; k < K; k++) acc = fmaf(A[k * 64 + 16 * b + i], B[k * 64 + 16 * b + j], acc);
                    uint32_t x, y; memcpy(&x, &acc, 4); memcpy(&y, &D[r * 64 + l], 4);
                    if (x != y && !(std::isnan(acc) && std::isnan(D[r * 64 + l]))) { if (bad < 3) printf("mode %d pass %d r %d l %d: host %a gpu %a\n", mode, pass, r, l, acc, D[r * 64 + l]); bad++; }
                }
            printf("mode %d %s: %d / 1024 elements differ\n", mode, pass ? "bias step as an MFMA" : "accumulator preloaded", bad);
            bad_total += bad;
        }
    }
    printf(bad_total ? "MFMA chain != fmaf chain\n" : "v_mfma_f32_16x16x1_4b_f32 K=1 chain == fmaf chain bit for bit (layout confirmed)\n");
    const int iters = 2000;
    for (int waves = 1; waves <= 2; waves++) {
        long long c1 = 0, c2 = 0, c4 = 0;
        hipLaunchKernelGGL(rate<1>, dim3(256), dim3(256 * waves), 0, 0, dout, dcyc, iters); hipMemcpy(&c1, dcyc, 8, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(rate<2>, dim3(256), dim3(256 * waves), 0, 0, dout, dcyc, iters); hipMemcpy(&c2, dcyc, 8, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(rate<4>, dim3(256), dim3(256 * waves), 0, 0, dout, dcyc, iters); hipMemcpy(&c4, dcyc, 8, hipMemcpyDeviceToHost);
        printf("%d wave(s) per SIMD: cycles per MFMA: 1 dependent chain %.1f, 2 interleaved chains %.1f, 4 chains %.1f\n", waves,
               (double)c1 / (iters * 8), (double)c2 / (iters * 16), (double)c4 / (iters * 32));
    }
    return bad_total != 0;
}
