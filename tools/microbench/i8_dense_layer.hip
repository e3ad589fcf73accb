// Block-fixed-point dense layer on the i8 matrix pipe: Y[32 rows][128] = X[32][128] . W[128][128] with X and W as
// 24-bit integers cut into three signed 8-bit limbs, the six limb products of weight >= 2^16 as v_mfma_i32_16x16x64_i8 into i32 accumulators
// (three shift classes), combined in 64-bit integers.  One workgroup of four waves per 32-row tile, each wave 32 output features — the
// controller's tiling (np_actor.h::actor32_body).  Checks the result against int64 arithmetic on the host (exact by construction) and
// times LAYERS back-to-back layers per workgroup (A limbs in LDS, B limbs streamed from global memory / L2, compiler-scheduled).
// hipcc --offload-arch=gfx950 -O3 i8_dense_layer.hip -o i8_dense_layer && ./i8_dense_layer
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
constexpr int ROWS = 32, K = 128, N = 128, LIMBS = 3;

// A limbs in LDS: xa[limb][row][k] int8.  B limbs in global, in FRAGMENT order: wb[set][limb][wave][k-step][lane][16 bytes].
// out[row][n][class] i32 (class 0: l2.w2, 1: l2.w1 + l1.w2, 2: l2.w0 + l1.w1 + l0.w2)
__global__ __launch_bounds__(256) void dense_i8(const int8_t *__restrict__ xa_g, const int8_t *__restrict__ wb, int *__restrict__ out, long long *cycles, int layers) {
    constexpr int KP = K + 16;   // row pitch 144 B: a pitch of 128 B puts the 32 rows of a ds_read_b128 on the same four banks (measured: 3 072 cycles per layer, all of it bank conflicts)
    __shared__ __attribute__((aligned(16))) int8_t xa[LIMBS][ROWS][KP];
    const int tid = threadIdx.x, wave = tid >> 6, l = tid & 63;
    for (int i = tid; i < LIMBS * ROWS * K / 16; i += 256)
        *reinterpret_cast<int4 *>(&xa[i / (ROWS * K / 16)][(i / (K / 16)) % ROWS][16 * (i % (K / 16))]) = reinterpret_cast<const int4 *>(xa_g)[i];
    __syncthreads();
    // v_mfma_i32_32x32x32_i8 (the 16x16x64 shape issues at half this rate from one wave per SIMD: tools/microbench/mfma_rates.hip): a wave's
    // 32 rows x 32 features are ONE block; lane l holds row (A) / feature (B) l & 31 and the 16 k-values 16 (l >> 5) .. + 15 of a 32-wide k-step
    i32x16 acc[3];
    i32x4 a[2][LIMBS], b[2][4][LIMBS];   // A: [buffer][limb], one k-step ahead (LDS); B: [buffer][k-step][limb], one LAYER ahead (L2 latency ~ a layer's MFMAs)
    auto fetch_b = [&](int buf, int layer) {
        const int8_t *w = wb + (size_t)(layer & 7) * LIMBS * N * K;   // eight different weight sets: the stream comes from L2, not from L1
#pragma unroll
        for (int ks = 0; ks < 4; ks++)
#pragma unroll
            for (int li = 0; li < LIMBS; li++)
                b[buf][ks][li] = *reinterpret_cast<const i32x4 *>(&w[((((size_t)li * 4 + wave) * 4 + ks) * 64 + l) * 16]);   // fragment order: one instruction = 1 KB of consecutive bytes
    };
    auto fetch_a = [&](int buf, int ks) {
#pragma unroll
        for (int li = 0; li < LIMBS; li++) a[buf][li] = *reinterpret_cast<const i32x4 *>(&xa[li][l & 31][32 * ks + 16 * (l >> 5)]);
    };
    auto mfmas = [&](int ab, int bb, int ks) {   // one accumulator is touched by every second instruction at most
        acc[2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ab][2], b[bb][ks][0], acc[2], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ab][2], b[bb][ks][1], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ab][1], b[bb][ks][1], acc[2], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ab][2], b[bb][ks][2], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ab][1], b[bb][ks][2], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ab][0], b[bb][ks][2], acc[2], 0, 0, 0);
    };
    const long long t0 = __builtin_readcyclecounter();
    fetch_b(0, 0);
    fetch_a(0, 0);
    for (int layer = 0; layer < layers; layer += 2) {   // two layers per trip: the buffer indices stay compile-time constants
#pragma unroll
        for (int half = 0; half < 2; half++) {
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[c][r] = 0;
            fetch_b(1 - half, layer + half + 1);
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                fetch_a((ks + 1) & 1, (ks + 1) & 3);
                mfmas(ks & 1, half, ks);
            }
            // a real layer converts, adds the bias, applies the activation and re-quantises here; the probe keeps the accumulators alive
            asm volatile("" : "+v"(acc[0]), "+v"(acc[2]));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
    if (blockIdx.x == 0) {
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int r = 0; r < 16; r++)   // C/D of the 32x32 shapes: col = l & 31, row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)
                out[(((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * N + 32 * wave + (l & 31)) * 3 + c] = acc[c][r];
    }
}

static void limbs_of(int v, int8_t *l0, int8_t *l1, int8_t *l2) {   // v = l2 * 65536 + l1 * 256 + l0, balanced signed digits
    int d0 = ((v + 128) & 255) - 128; v = (v - d0) >> 8;
    int d1 = ((v + 128) & 255) - 128; v = (v - d1) >> 8;
    *l0 = (int8_t)d0; *l1 = (int8_t)d1; *l2 = (int8_t)v;
}

int main(int argc, char **argv) {
    const int layers = argc > 1 ? atoi(argv[1]) : 9;
    std::mt19937 rng(7);
    std::vector<int> X(ROWS * K), W((size_t)8 * K * N);
    for (auto &v : X) v = (int)(rng() % (1 << 23)) - (1 << 22);
    for (auto &v : W) v = (int)(rng() % (1 << 23)) - (1 << 22);
    std::vector<int8_t> xa((size_t)LIMBS * ROWS * K), wb((size_t)8 * LIMBS * N * K);
    for (int r = 0; r < ROWS; r++)
        for (int k = 0; k < K; k++) limbs_of(X[r * K + k], &xa[(0 * ROWS + r) * K + k], &xa[(1 * ROWS + r) * K + k], &xa[(2 * ROWS + r) * K + k]);
    for (int s = 0; s < 8; s++)
        for (int k = 0; k < K; k++)
            for (int n = 0; n < N; n++) {
                int8_t l[3];
                limbs_of(W[((size_t)s * K + k) * N + n], &l[0], &l[1], &l[2]);
                // fragment order (a load instruction reads 64 lanes x 16 consecutive bytes; [n][k] order put every lane on its own 128-byte line: 3 072 cycles per layer)
                for (int li = 0; li < 3; li++) {
                    const int wave = n / 32, lane = (n & 31) + 32 * ((k & 31) / 16), ks = k / 32, e = k & 15;
                    wb[((((((size_t)s * LIMBS + li) * 4 + wave) * 4 + ks) * 64 + lane) * 16) + e] = l[li];
                }
            }
    int8_t *dx, *dw; int *dout; long long *dcyc;
    const int blocks = 256;
    CHECK(hipMalloc(&dx, xa.size())); CHECK(hipMalloc(&dw, wb.size())); CHECK(hipMalloc(&dout, ROWS * N * 3 * 4)); CHECK(hipMalloc(&dcyc, blocks * 8));
    CHECK(hipMemcpy(dx, xa.data(), xa.size(), hipMemcpyHostToDevice)); CHECK(hipMemcpy(dw, wb.data(), wb.size(), hipMemcpyHostToDevice));
    // correctness: one layer, weight set 0
    hipLaunchKernelGGL(dense_i8, dim3(1), dim3(256), 0, 0, dx, dw, dout, dcyc, 2);   // two layers (the loop's trip): the second one, weight set 1, is what is left in the accumulators
    CHECK(hipDeviceSynchronize());
    std::vector<int> out(ROWS * N * 3);
    CHECK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
    long long bad = 0; double worst = 0;
    for (int r = 0; r < ROWS; r++)
        for (int n = 0; n < N; n++) {
            long long c0 = 0, c1 = 0, c2 = 0; __int128 full = 0;
            for (int k = 0; k < K; k++) {
                int8_t x0, x1, x2, w0, w1, w2;
                limbs_of(X[r * K + k], &x0, &x1, &x2); limbs_of(W[((size_t)1 * K + k) * N + n], &w0, &w1, &w2);
                c0 += x2 * w2; c1 += x2 * w1 + x1 * w2; c2 += x2 * w0 + x1 * w1 + x0 * w2;
                full += (__int128)X[r * K + k] * W[((size_t)1 * K + k) * N + n];
            }
            if (out[(r * N + n) * 3 + 0] != c0 || out[(r * N + n) * 3 + 1] != c1 || out[(r * N + n) * 3 + 2] != c2) bad++;
            const double six = (double)c0 * 4294967296.0 + (double)c1 * 16777216.0 + (double)c2 * 65536.0;
            const double rel = fabs(six - (double)full) / (128.0 * 4194304.0 * 4194304.0);   // against K x max|x| x max|w|
            if (rel > worst) worst = rel;
        }
    printf("one layer, 32 x 128 x 128: %lld of %d outputs differ from int64 arithmetic; six-of-nine limb products vs all nine: worst %.3g of K * max|x| * max|w| (2^-24 = %.3g)\n",
           bad, ROWS * N, worst, 1.0 / 16777216.0);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(dense_i8, dim3(blocks), dim3(256), 0, 0, dx, dw, dout, dcyc, layers * 50);
        CHECK(hipDeviceSynchronize());
    }
    std::vector<long long> cyc(blocks);
    CHECK(hipMemcpy(cyc.data(), dcyc, blocks * 8, hipMemcpyDeviceToHost));
    long long mx = 0, mn = 1ll << 62; for (auto c : cyc) { mx = c > mx ? c : mx; mn = c < mn ? c : mn; }
    printf("%d workgroups x %d layers: %.0f .. %.0f shader cycles per layer (24 x v_mfma_i32_32x32x32_i8 per wave and layer: 768 cycles of matrix pipe); the fp32 layer of the shipped controller: 5 160\n",
           blocks, layers * 50, (double)mn / (layers * 50), (double)mx / (layers * 50));
    return bad != 0;
}
