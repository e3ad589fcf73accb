// mfma4_net.hip — research microbenchmark (round 3): can the aero nets be evaluated WITHOUT the scalar weight stream
// for the small and mid-size batches, where one wave per SIMD waits ~300 cycles per 48-float weight group?
//
// v_mfma_f32_4x4x1_16b_f32 (16 blocks of a 4x4 outer product, K = 1): with block b = lanes 4b..4b+3,
//     A: lane l holds a[l % 4] of its block        B: lane l holds b[l % 4] of its block        D: VGPR r, lane l = a[r] * b[l % 4] + C
// so with B = "lane l holds the activation x_k of aircraft l" and A = "lane l holds W[4g + l % 4][k]" (the same four weights in every
// block), D[r][l] += W[4g + r][k] * x_k[l]: FOUR neurons of the layer for all 64 aircraft per instruction, results in the natural
// "one lane per aircraft, one VGPR per neuron" layout — the arithmetic of two v_pk_fma_f32, at the same pipe rate, but the weights
// arrive as a VGPR loaded from LDS (in-order ds_read, ~100-cycle latency, deep pipelining) instead of an SGPR pair loaded through the
// 16 KB scalar cache (out-of-order s_load, only lgkmcnt(0) usable, ~300-cycle round trip on a miss).
//
// This file: (1) checks that a K = 1 chain of this instruction equals the scalar fmaf chain of the numerics spec bit for bit
// (bias as a first step with B = 1: bias * 1 + 0 is exact), (2) times the two hidden layers of NETS 1-20-10 nets per wave with the
// weights in LDS, for 1 wave per CU and 1 wave per SIMD, and prints cycles per net.  Compare: the scalar-stream bodies need 138 VALU
// instructions per such net = 552 cycles at full issue, and measure ~1 700 cycles per net on a lone wave (docs/DESIGN_HISTORY.md §3:
// 13 cycles per instruction in the four-wave latency kernel).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma4_net mfma4_net.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int H1 = 20, H2 = 10, G1 = 5, G2 = 3;  // neuron groups of four per layer (10 -> 12: two padding neurons)
constexpr int NETS = 12;
// LDS record of one net, A-operand order: layer 1: [g][4 neurons][2] = (bias, W[n][0]); layer 2: [g][4 neurons][21] padded to 24 =
// (bias, W[n][0..19]) — lane l reads the 16 bytes at (group base) + (l % 4) * pitch + 4 * k: four consecutive K-steps of its neuron
constexpr int L1_PITCH = 4, L2_PITCH = 24;
constexpr int REC = G1 * 4 * L1_PITCH + G2 * 4 * L2_PITCH;  // floats per net in LDS: 80 + 288 = 368

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float relu1(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

// the two hidden layers of one 1-20-10 net for this lane's aircraft; weights from LDS
__device__ __forceinline__ void net_mfma(const float *rec, int l4, float x, float (&h2)[12]) {
    f32x4 h1[G1];
#pragma unroll
    for (int g = 0; g < G1; g++) {
        const f32x4 w = *reinterpret_cast<const f32x4 *>(rec + (g * 4 + l4) * L1_PITCH);  // (bias, W0, -, -)
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        acc = mfma4(w[0], 1.0f, acc);  // acc = bias
        acc = mfma4(w[1], x, acc);     // acc = fma(W[n][0], x, acc)
#pragma unroll
        for (int r = 0; r < 4; r++) acc[r] = relu1(acc[r]);
        h1[g] = acc;
    }
    const float *l2 = rec + G1 * 4 * L1_PITCH;
    // layer 2: the three neuron groups are three INDEPENDENT accumulator chains — issued interleaved (a dependent K = 1 MFMA cannot
    // issue back to back: ~20 cycles from one to the next on the same accumulator)
    f32x4 acc[G2], w[G2];
#pragma unroll
    for (int g = 0; g < G2; g++) {
        acc[g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        w[g] = *reinterpret_cast<const f32x4 *>(l2 + (g * 4 + l4) * L2_PITCH);
    }
#pragma unroll
    for (int g = 0; g < G2; g++) acc[g] = mfma4(w[g][0], 1.0f, acc[g]);
#pragma unroll
    for (int v = 0; v < 6; v++) {
        if (v > 0) {
#pragma unroll
            for (int g = 0; g < G2; g++) w[g] = *reinterpret_cast<const f32x4 *>(l2 + (g * 4 + l4) * L2_PITCH + 4 * v);
        }
#pragma unroll
        for (int j = (v == 0 ? 1 : 0); j < 4; j++) {
            const int k = 4 * v + j - 1;
            if (k < H1) {
#pragma unroll
                for (int g = 0; g < G2; g++) acc[g] = mfma4(w[g][j], h1[k / 4][k % 4], acc[g]);
            }
        }
    }
#pragma unroll
    for (int g = 0; g < G2; g++)
#pragma unroll
        for (int r = 0; r < 4; r++) h2[4 * g + r] = relu1(acc[g][r]);
}

// three nets at once: nine independent layer-2 chains (3 nets x 3 neuron groups) issued round-robin
__device__ __forceinline__ void net_mfma3(const float *rec, int l4, float x, float (&h2)[3][12]) {
    f32x4 h1[3][G1];
#pragma unroll
    for (int g = 0; g < G1; g++) {
        f32x4 w[3], acc[3];
#pragma unroll
        for (int n = 0; n < 3; n++) {
            w[n] = *reinterpret_cast<const f32x4 *>(rec + n * REC + (g * 4 + l4) * L1_PITCH);
            acc[n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int n = 0; n < 3; n++) acc[n] = mfma4(w[n][0], 1.0f, acc[n]);
#pragma unroll
        for (int n = 0; n < 3; n++) acc[n] = mfma4(w[n][1], x, acc[n]);
#pragma unroll
        for (int n = 0; n < 3; n++) {
#pragma unroll
            for (int r = 0; r < 4; r++) acc[n][r] = relu1(acc[n][r]);
            h1[n][g] = acc[n];
        }
    }
    f32x4 acc[3][G2], w[3][G2];
#pragma unroll
    for (int n = 0; n < 3; n++)
#pragma unroll
        for (int g = 0; g < G2; g++) {
            acc[n][g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            w[n][g] = *reinterpret_cast<const f32x4 *>(rec + n * REC + G1 * 4 * L1_PITCH + (g * 4 + l4) * L2_PITCH);
        }
#pragma unroll
    for (int n = 0; n < 3; n++)
#pragma unroll
        for (int g = 0; g < G2; g++) acc[n][g] = mfma4(w[n][g][0], 1.0f, acc[n][g]);
#pragma unroll
    for (int v = 0; v < 6; v++) {
        if (v > 0) {
#pragma unroll
            for (int n = 0; n < 3; n++)
#pragma unroll
                for (int g = 0; g < G2; g++) w[n][g] = *reinterpret_cast<const f32x4 *>(rec + n * REC + G1 * 4 * L1_PITCH + (g * 4 + l4) * L2_PITCH + 4 * v);
        }
#pragma unroll
        for (int j = (v == 0 ? 1 : 0); j < 4; j++) {
            const int k = 4 * v + j - 1;
            if (k < H1) {
#pragma unroll
                for (int n = 0; n < 3; n++)
#pragma unroll
                    for (int g = 0; g < G2; g++) acc[n][g] = mfma4(w[n][g][j], h1[n][k / 4][k % 4], acc[n][g]);
            }
        }
    }
#pragma unroll
    for (int n = 0; n < 3; n++)
#pragma unroll
        for (int g = 0; g < G2; g++)
#pragma unroll
            for (int r = 0; r < 4; r++) h2[n][4 * g + r] = relu1(acc[n][g][r]);
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void bench3_kernel(const float *__restrict__ recs, const float *__restrict__ xin, float *__restrict__ out,
                                                            long long *__restrict__ cycles, int reps) {
    __shared__ __attribute__((aligned(16))) float lds[NETS * REC];
    for (int i = threadIdx.x; i < NETS * REC; i += blockDim.x) lds[i] = recs[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, l4 = lane & 3;
    const float x = xin[(long long)blockIdx.x * blockDim.x + threadIdx.x];
    float sum = 0.0f;
    const long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < reps; rep++) {
#pragma unroll 1
        for (int n = 0; n < NETS; n += 3) {
            float h2[3][12];
            net_mfma3(lds + n * REC, l4, x + (float)rep * 1e-3f, h2);
#pragma unroll
            for (int q = 0; q < 3; q++)
#pragma unroll
                for (int j = 0; j < H2; j++) sum += h2[q][j];
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[(long long)blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void bench_kernel(const float *__restrict__ recs, const float *__restrict__ xin, float *__restrict__ out,
                                                           long long *__restrict__ cycles, int reps) {
    __shared__ __attribute__((aligned(16))) float lds[NETS * REC];
    for (int i = threadIdx.x; i < NETS * REC; i += blockDim.x) lds[i] = recs[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, l4 = lane & 3;
    const float x = xin[(long long)blockIdx.x * blockDim.x + threadIdx.x];
    float sum = 0.0f;
    const long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < reps; rep++) {
#pragma unroll 1
        for (int n = 0; n < NETS; n++) {
            float h2[12];
            net_mfma(lds + n * REC, l4, x + (float)rep * 1e-3f, h2);
#pragma unroll
            for (int j = 0; j < H2; j++) sum += h2[j];
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[(long long)blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// h2 of one net for every lane: the exactness check
__global__ void check_kernel(const float *__restrict__ recs, const float *__restrict__ xin, float *__restrict__ h2out) {
    __shared__ __attribute__((aligned(16))) float lds[REC];
    for (int i = threadIdx.x; i < REC; i += blockDim.x) lds[i] = recs[i];
    __syncthreads();
    float h2[12];
    net_mfma(lds, threadIdx.x & 3, xin[threadIdx.x], h2);
    for (int j = 0; j < 12; j++) h2out[threadIdx.x * 12 + j] = h2[j];
}

int main() {
    std::vector<float> recs(NETS * REC, 0.0f), x(1 << 16);
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.0f - 1.0f; };
    std::vector<float> W1(NETS * H1), B1(NETS * H1), W2(NETS * H2 * H1), B2(NETS * H2);
    for (auto &v : W1) v = rnd() * 0.8f;
    for (auto &v : B1) v = rnd() * 0.3f;
    for (auto &v : W2) v = rnd() * 0.5f;
    for (auto &v : B2) v = rnd() * 0.2f;
    for (auto &v : x) v = rnd() * 1.5f;
    for (int n = 0; n < NETS; n++) {
        float *r = recs.data() + n * REC;
        for (int j = 0; j < H1; j++) {
            r[j * L1_PITCH + 0] = B1[n * H1 + j];
            r[j * L1_PITCH + 1] = W1[n * H1 + j];
        }
        float *l2 = r + G1 * 4 * L1_PITCH;
        for (int j = 0; j < H2; j++) {
            l2[j * L2_PITCH] = B2[n * H2 + j];
            for (int k = 0; k < H1; k++) l2[j * L2_PITCH + 1 + k] = W2[(n * H2 + j) * H1 + k];
        }
    }
    float *d_recs, *d_x, *d_out, *d_h2;
    long long *d_cyc;
    CHECK(hipMalloc(&d_recs, recs.size() * 4));
    CHECK(hipMalloc(&d_x, x.size() * 4));
    CHECK(hipMalloc(&d_out, x.size() * 4));
    CHECK(hipMalloc(&d_h2, 64 * 12 * 4));
    CHECK(hipMalloc(&d_cyc, 4096 * 8));
    CHECK(hipMemcpy(d_recs, recs.data(), recs.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_x, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    // (1) exactness against the scalar rule: acc = bias; acc = fmaf(W[j][k], x[k], acc), k ascending; relu = clamp to [0, 1]
    hipLaunchKernelGGL(check_kernel, dim3(1), dim3(64), 0, 0, d_recs, d_x, d_h2);
    std::vector<float> h2(64 * 12);
    CHECK(hipMemcpy(h2.data(), d_h2, h2.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; l++) {
        float h1[H1];
        for (int j = 0; j < H1; j++) {
            float acc = B1[j];
            acc = fmaf(W1[j], x[l], acc);
            h1[j] = fminf(fmaxf(acc, 0.0f), 1.0f);
        }
        for (int j = 0; j < H2; j++) {
            float acc = B2[j];
            for (int k = 0; k < H1; k++) acc = fmaf(W2[j * H1 + k], h1[k], acc);
            const float want = fminf(fmaxf(acc, 0.0f), 1.0f);
            if (memcmp(&want, &h2[l * 12 + j], 4) != 0) bad++;
        }
    }
    printf("exactness: %d of %d hidden-layer-2 outputs differ from the scalar fmaf chain\n", bad, 64 * H2);
    // (2) timing
    const int reps = 20;
    for (int cfg = 0; cfg < 2; cfg++) {
        const int waves = cfg == 0 ? 1 : cfg == 1 ? 4 : 8;   // per workgroup = per CU (one workgroup per CU: grid 256)
        const int grid = 256;
        for (int it = 0; it < 3; it++) {
            if (waves == 1) hipLaunchKernelGGL(bench_kernel<1>, dim3(grid), dim3(64), 0, 0, d_recs, d_x, d_out, d_cyc, reps);
            else if (waves == 4) hipLaunchKernelGGL(bench_kernel<4>, dim3(grid), dim3(256), 0, 0, d_recs, d_x, d_out, d_cyc, reps);
            else hipLaunchKernelGGL(bench_kernel<8>, dim3(grid), dim3(512), 0, 0, d_recs, d_x, d_out, d_cyc, reps);
            CHECK(hipDeviceSynchronize());
        }
        std::vector<long long> cyc(grid);
        CHECK(hipMemcpy(cyc.data(), d_cyc, grid * 8, hipMemcpyDeviceToHost));
        double mean = 0;
        for (auto c : cyc) mean += (double)c;
        mean /= grid;
        printf("%d wave(s) per CU: %.0f cycles per 1-20-10 net (two hidden layers, 73 MFMAs + 30 clamps) per wave\n", waves, mean / (reps * NETS));
    }
    for (int cfg = 0; cfg < 2; cfg++) {
        const int waves = cfg == 0 ? 1 : 4, grid = 256;
        for (int it = 0; it < 3; it++) {
            if (waves == 1) hipLaunchKernelGGL(bench3_kernel<1>, dim3(grid), dim3(64), 0, 0, d_recs, d_x, d_out, d_cyc, reps);
            else hipLaunchKernelGGL(bench3_kernel<4>, dim3(grid), dim3(256), 0, 0, d_recs, d_x, d_out, d_cyc, reps);
            CHECK(hipDeviceSynchronize());
        }
        std::vector<long long> cyc(grid);
        CHECK(hipMemcpy(cyc.data(), d_cyc, grid * 8, hipMemcpyDeviceToHost));
        double mean = 0;
        for (auto c : cyc) mean += (double)c;
        mean /= grid;
        printf("three nets interleaved (nine chains), %d wave(s) per CU: %.0f cycles per net per wave\n", waves, mean / (reps * NETS));
    }
    return 0;
}
