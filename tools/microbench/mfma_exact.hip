// Is v_mfma_f32_32x32x1_2b_f32 (K = 1 per instruction) an exact IEEE fused multiply-add per element, so that a chain of them
// over k reproduces acc = fmaf(a[k], b[k], acc), k ascending, bit for bit?  Also pins the operand / accumulator layout:
//   A: lane l -> block l/32, row i = l%32;  B: lane l -> block l/32, column j = l%32;
//   D: 32 VGPRs, block b = r/16, r' = r%16, lane l -> i = 8*(r'/4) + 4*(l/32) + r'%4, j = l%32.
// hipcc --offload-arch=gfx950 -O2 -ffp-contract=off mfma_exact.hip -o mfma_exact && ./mfma_exact
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x32 __attribute__((ext_vector_type(32)));
constexpr int K = 128;
__global__ void chain(const float *A, const float *B, const float *C, float *D) {
    const int l = threadIdx.x;
    f32x32 acc;
    for (int r = 0; r < 32; r++) acc[r] = C[r * 64 + l];
    for (int k = 0; k < K; k++) acc = __builtin_amdgcn_mfma_f32_32x32x1f32(A[k * 64 + l], B[k * 64 + l], acc, 0, 0, 0);
    for (int r = 0; r < 32; r++) D[r * 64 + l] = acc[r];
}
static float rnd(int mode) {
    const float u = (float)rand() / (float)RAND_MAX * 2.0f - 1.0f;
    switch (mode) {
        case 0: return u;
        case 1: return u * 1e-20f;                      // products are denormal / underflow
        case 2: return u * 1e18f;                       // products near overflow
        default: return (rand() & 7) == 0 ? u * 1e-30f : u * (float)(1 << (rand() % 20));
    }
}
int main() {
    std::vector<float> A(K * 64), B(K * 64), C(32 * 64), D(32 * 64);
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
    int bad_total = 0;
    for (int mode = 0; mode < 4; mode++) {
        srand(17 + mode);
        for (auto &x : A) x = rnd(mode);
        for (auto &x : B) x = rnd(mode == 2 ? 0 : mode);
        for (auto &x : C) x = rnd(mode);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int r = 0; r < 32; r++)
            for (int l = 0; l < 64; l++) {
                const int b = r / 16, rr = r % 16, i = 8 * (rr / 4) + 4 * (l / 32) + rr % 4, j = l % 32;
                float acc = C[r * 64 + l];
                for (int k = 0; k < K; k++) acc = fmaf(A[k * 64 + 32 * b + i], B[k * 64 + 32 * b + j], acc);
                uint32_t x, y; memcpy(&x, &acc, 4); memcpy(&y, &D[r * 64 + l], 4);
                if (x != y && !(std::isnan(acc) && std::isnan(D[r * 64 + l]))) { if (bad < 3) printf("mode %d r %d l %d: host %a gpu %a\n", mode, r, l, acc, D[r * 64 + l]); bad++; }
            }
        printf("mode %d: %d / 2048 elements differ\n", mode, bad);
        bad_total += bad;
    }
    printf(bad_total ? "MFMA chain != fmaf chain\n" : "MFMA K=1 chain == fmaf chain bit for bit (layout confirmed)\n");
    return bad_total != 0;
}
