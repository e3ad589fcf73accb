// DESIGN.md section 14 (b), one step further than i8_dense_layer.hip: a STACK of Linear(128, 128) + ReLU layers in block fixed point (sign + 22 bits:
// the top balanced limb of a 24-bit value can reach +128, which an int8 does not hold — found by this probe)
// with everything a layer needs on the device — the quantiser (row maximum -> exponent, round to nearest even, three balanced signed 8-bit
// limbs into LDS), the six limb products on v_mfma_i32_32x32x32_i8, the epilogue (64-bit combine, ONE rounding to fp32, scale, bias, ReLU,
// activations back to LDS) — checked bit for bit against the same arithmetic on the host and timed per layer.  One workgroup of four waves per
// 32-row tile (the controller's tiling); LayerNorms, gates and the recurrent state are not part of this probe.
// hipcc --offload-arch=gfx950 -O3 i8_mlp_stack.hip -o i8_mlp_stack && ./i8_mlp_stack [layers]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
constexpr int ROWS = 32, K = 128, N = 128, LIMBS = 3, KP = K + 16, XP = K + 4, SETS = 8;

// ---- the numerics, once for both sides -------------------------------------------------------------------------------------------
// exponent of a row: the smallest e with max|x| < 2^e (from the fp32 exponent field; 0 -> -126)
__host__ __device__ inline int row_exponent(float m) {
    uint32_t u; memcpy(&u, &m, 4);
    return (int)((u >> 23) & 255) - 126;
}
__host__ __device__ inline float pow2f(int e) {   // 2^e for -126 <= e <= 127
    uint32_t u = (uint32_t)(e + 127) << 23; float f; memcpy(&f, &u, 4); return f;
}
__host__ __device__ inline int quantise(float x, int e) {   // rint(x * 2^(22 - e)): |q| <= 2^22 because |x| < 2^e; two exact scalings keep 2^(22 - e) in range
    float s = x * pow2f(22 - e > 127 ? 127 : 22 - e);
    if (22 - e > 127) s *= pow2f(22 - e - 127);
#ifdef __HIP_DEVICE_COMPILE__
    int q = (int)__builtin_rintf(s);
#else
    int q = (int)nearbyintf(s);
#endif
    return q;
}
__host__ __device__ inline void limbs_of(int v, int &l0, int &l1, int &l2) {   // v = l2 * 65536 + l1 * 256 + l0, balanced signed digits = sign-extended bytes
    l0 = (int)(int8_t)(v & 255); v = (v - l0) >> 8;
    l1 = (int)(int8_t)(v & 255); v = (v - l1) >> 8;
    l2 = v;   // |l2| <= 64
}
// the six limb products of weight >= 2^16, then ONE rounding to fp32, the scale 2^(ex + ew - 46) (two exact factors), bias, ReLU
__host__ __device__ inline float finish(int c0, int c1, int c2, int ex, int ew, float bias) {
    // S = (c0 2^8 + c1) 2^8 + c2 (times 2^16) as TWO fused multiply-adds on the exactly converted class sums (|c| < 2^23): two roundings, the
    // same two on any machine that has fmaf — the 64-bit / double-precision single-rounding combine costs 3.5 x the issue slots (quarter-rate
    // conversions) for one rounding less
    const float u = fmaf((float)c0, 256.f, (float)c1);
    float y = fmaf(u, 256.f, (float)c2);
    const int e = ex + ew - 44 + 16;
    y = y * pow2f(e < -126 ? -126 : e > 127 ? 127 : e);   // results stay far from the denormal range for the operands of this probe
    y = y + bias;
    return y > 0.f ? y : 0.f;
}

__global__ __launch_bounds__(256) void mlp_stack(const float *__restrict__ x0, const int8_t *__restrict__ wb, const int *__restrict__ ew,
                                                 const float *__restrict__ bias, float *__restrict__ out, long long *cycles, int layers) {
    __shared__ __attribute__((aligned(16))) int8_t xa[LIMBS][ROWS][KP];
    __shared__ __attribute__((aligned(16))) float X[ROWS][XP];
    __shared__ int ex_s[ROWS];
    const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63;
    for (int i = tid; i < ROWS * K; i += 256) X[i / K][i % K] = x0[i];
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    i32x4 b[2][4][LIMBS];
    auto fetch_b = [&](int buf, int layer) {
        const int8_t *w = wb + (size_t)(layer % SETS) * LIMBS * N * K;
#pragma unroll
        for (int ks = 0; ks < 4; ks++)
#pragma unroll
            for (int li = 0; li < LIMBS; li++) b[buf][ks][li] = *reinterpret_cast<const i32x4 *>(&w[((((size_t)li * 4 + wave) * 4 + ks) * 64 + l) * 16]);
    };
    fetch_b(0, 0);
    auto one_layer = [&](int layer, int bb) {
#ifndef SKIP_QUANT
        // 1. quantiser: thread t holds 4 x 4 consecutive features of row t / 8 (float4 reads, one packed dword per limb and group out)
        {
            const int r = tid >> 3, f0 = 4 * (tid & 7);
            float4 v[4];
            float m = 0.f;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                v[j] = *reinterpret_cast<const float4 *>(&X[r][f0 + 32 * j]);
                m = fmaxf(fmaxf(m, fmaxf(fabsf(v[j].x), fabsf(v[j].y))), fmaxf(fabsf(v[j].z), fabsf(v[j].w)));
            }
            m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4));
            const int e = row_exponent(m);
            if ((tid & 7) == 0) ex_s[r] = e;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float xs[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
                unsigned p0 = 0, p1 = 0, p2 = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    int a0, a1, a2;
                    limbs_of(quantise(xs[i], e), a0, a1, a2);
                    p0 |= (unsigned)(a0 & 255) << (8 * i); p1 |= (unsigned)(a1 & 255) << (8 * i); p2 |= (unsigned)(a2 & 255) << (8 * i);
                }
                *reinterpret_cast<unsigned *>(&xa[0][r][f0 + 32 * j]) = p0;
                *reinterpret_cast<unsigned *>(&xa[1][r][f0 + 32 * j]) = p1;
                *reinterpret_cast<unsigned *>(&xa[2][r][f0 + 32 * j]) = p2;
            }
        }
#endif
        __syncthreads();
        // 2. the six limb products (B of this layer was requested a layer ago; the next layer's is requested now)
        fetch_b(1 - bb, layer + 1);
        i32x16 acc[3];
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[c][r] = 0;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            i32x4 a[LIMBS];
#pragma unroll
            for (int li = 0; li < LIMBS; li++) a[li] = *reinterpret_cast<const i32x4 *>(&xa[li][l & 31][32 * ks + 16 * (l >> 5)]);
            acc[2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[2], b[bb][ks][0], acc[2], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[2], b[bb][ks][1], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[1], b[bb][ks][1], acc[2], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[2], b[bb][ks][2], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[1], b[bb][ks][2], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0], b[bb][ks][2], acc[2], 0, 0, 0);
        }
        // 3. epilogue: C/D of the 32x32 shapes: feature = 32 wave + (l & 31), row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)
        const int f = 32 * wave + (l & 31);
        const int ewf = ew[(layer % SETS) * N + f];
        const float bf = bias[(layer % SETS) * N + f];
#ifdef SKIP_EPI
        if (acc[0][0] == 0x7fffffff) X[0][f] = (float)acc[1][3] + (float)acc[2][5];
#else
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
            X[row][f] = finish(acc[0][r], acc[1][r], acc[2][r], ex_s[row], ewf, bf);
        }
#endif
        __syncthreads();
    };
    for (int layer = 0; layer < layers; layer += 2) {
        one_layer(layer, 0);
        if (layer + 1 < layers) one_layer(layer + 1, 1);
    }
    const long long t1 = __builtin_readcyclecounter();
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
    if (blockIdx.x == 0)
        for (int i = tid; i < ROWS * N; i += 256) out[i] = X[i / N][i % N];
}

int main(int argc, char **argv) {
    const int layers = argc > 1 ? atoi(argv[1]) : 4;
    std::mt19937 rng(11);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> x0(ROWS * K), W((size_t)SETS * N * K), bias(SETS * N);
    for (auto &v : x0) v = nd(rng);
    for (auto &v : W) v = nd(rng) * 0.09f;      // ~ 1 / sqrt(128): activations keep their scale from layer to layer
    for (auto &v : bias) v = nd(rng) * 0.1f;
    // weights: per output feature exponent, 24-bit integers, limbs in fragment order
    std::vector<int> wq((size_t)SETS * N * K), ew(SETS * N);
    std::vector<int8_t> wb((size_t)SETS * LIMBS * N * K);
    for (int s = 0; s < SETS; s++)
        for (int n = 0; n < N; n++) {
            float m = 0.f;
            for (int k = 0; k < K; k++) m = fmaxf(m, fabsf(W[((size_t)s * N + n) * K + k]));
            const int e = row_exponent(m);
            ew[s * N + n] = e;
            for (int k = 0; k < K; k++) {
                const int q = quantise(W[((size_t)s * N + n) * K + k], e);
                wq[((size_t)s * N + n) * K + k] = q;
                int l0, l1, l2; limbs_of(q, l0, l1, l2);
                const int lv[3] = {l0, l1, l2};
                const int wave = n / 32, lane = (n & 31) + 32 * ((k & 31) / 16), ks = k / 32, e16 = k & 15;
                for (int li = 0; li < 3; li++) wb[((((((size_t)s * LIMBS + li) * 4 + wave) * 4 + ks) * 64 + lane) * 16) + e16] = (int8_t)lv[li];
            }
        }
    // host restatement
    std::vector<float> X(x0), Y(ROWS * N);
    for (int layer = 0; layer < layers; layer++) {
        const int s = layer % SETS;
        for (int r = 0; r < ROWS; r++) {
            float m = 0.f;
            for (int k = 0; k < K; k++) m = fmaxf(m, fabsf(X[r * K + k]));
            const int ex = row_exponent(m);
            int x0l[K], x1l[K], x2l[K];
            for (int k = 0; k < K; k++) limbs_of(quantise(X[r * K + k], ex), x0l[k], x1l[k], x2l[k]);
            for (int n = 0; n < N; n++) {
                int c0 = 0, c1 = 0, c2 = 0;
                for (int k = 0; k < K; k++) {
                    int w0, w1, w2; limbs_of(wq[((size_t)s * N + n) * K + k], w0, w1, w2);
                    c0 += x2l[k] * w2; c1 += x2l[k] * w1 + x1l[k] * w2; c2 += x2l[k] * w0 + x1l[k] * w1 + x0l[k] * w2;
                }
                Y[r * N + n] = finish(c0, c1, c2, ex, ew[s * N + n], bias[s * N + n]);
            }
        }
        X = Y;
    }
    float *dx, *dbias, *dout; int8_t *dw; int *dew; long long *dcyc;
    const int blocks = 256;
    CHECK(hipMalloc(&dx, x0.size() * 4)); CHECK(hipMalloc(&dbias, bias.size() * 4)); CHECK(hipMalloc(&dout, ROWS * N * 4));
    CHECK(hipMalloc(&dw, wb.size())); CHECK(hipMalloc(&dew, ew.size() * 4)); CHECK(hipMalloc(&dcyc, blocks * 8));
    CHECK(hipMemcpy(dx, x0.data(), x0.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dbias, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dw, wb.data(), wb.size(), hipMemcpyHostToDevice)); CHECK(hipMemcpy(dew, ew.data(), ew.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mlp_stack, dim3(1), dim3(256), 0, 0, dx, dw, dew, dbias, dout, dcyc, layers);
    CHECK(hipDeviceSynchronize());
    std::vector<float> got(ROWS * N);
    CHECK(hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0; double ref_max = 0;
    for (int i = 0; i < ROWS * N; i++) { if (memcmp(&got[i], &X[i], 4)) bad++; ref_max = fmax(ref_max, fabs(X[i])); }
    // the same stack in double precision on the unquantised weights: how far the fixed point is from real arithmetic
    std::vector<double> Xd(x0.begin(), x0.end()), Yd(ROWS * N);
    for (int layer = 0; layer < layers; layer++) {
        const int s = layer % SETS;
        for (int r = 0; r < ROWS; r++)
            for (int n = 0; n < N; n++) {
                double a = bias[s * N + n];
                for (int k = 0; k < K; k++) a += Xd[r * K + k] * (double)W[((size_t)s * N + n) * K + k];
                Yd[r * N + n] = a > 0 ? a : 0;
            }
        Xd = Yd;
    }
    double worst = 0; for (int i = 0; i < ROWS * N; i++) worst = fmax(worst, fabs((double)got[i] - Xd[i]));
    printf("%d layers of Linear(128, 128) + ReLU, 32 rows: %d of %d outputs differ from the host restatement of the same integer arithmetic; max |fixed point - float64| %.3g (outputs up to %.3g)\n",
           layers, bad, ROWS * N, worst, ref_max);
    const int timed = 360;
    for (int rep = 0; rep < 3; rep++) { hipLaunchKernelGGL(mlp_stack, dim3(blocks), dim3(256), 0, 0, dx, dw, dew, dbias, dout, dcyc, timed); CHECK(hipDeviceSynchronize()); }
    std::vector<long long> cyc(blocks);
    CHECK(hipMemcpy(cyc.data(), dcyc, blocks * 8, hipMemcpyDeviceToHost));
    long long mx = 0, mn = 1ll << 62; for (auto c : cyc) { mx = c > mx ? c : mx; mn = c < mn ? c : mn; }
    printf("%d workgroups x %d layers: %.0f .. %.0f shader cycles per layer INCLUDING quantiser and epilogue (matrix pipe 768; the shipped fp32 layer 5 160)\n", blocks, timed,
           (double)mn / timed, (double)mx / timed);
    return bad != 0;
}
