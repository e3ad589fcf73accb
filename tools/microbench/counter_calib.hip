// counter_calib.hip — known-byte-count kernels in the env kernel's own access patterns, to calibrate rocprofv3's FETCH_SIZE /
// WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, HBM section: FETCH_SIZE reads 1/2 of a 16 B/lane streaming read; every other
// width and WRITE_SIZE are uncalibrated).  Each kernel moves an exactly known number of bytes per launch:
//
//   rd_dword   R rows of [n] f32, one dword per lane and row (SoA: s / u / tgt / coefficient-cache rows)      R * 4 * n read
//   wr_dword   W rows of [n] f32, one dword per lane and row                                                  W * 4 * n written
//   rd_u8/wr_u8 3 rows of [n] bytes (the done / bad / timeout flags)                                          3 * n
//   rd_i64/wr_i64 one [n] int64 row (step_count)                                                              8 * n
//   rd_act     [n][4] f32 row-major, four dword loads per lane at a 16-byte pitch (the action rows)           16 * n read
//   wr_obs     [n][22] f32 row-major written as 16-byte vectors (the observation tile)                        88 * n written
//   rd_vec16   16 B per lane streaming read (the guide's calibration point, for comparison)                   16 * n read
//   mix        the env.step kernel's whole per-aircraft pattern (163 B read + 227 B written incl. the cache)  390 * n
//
// Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in separate passes (tools/calibrate_counters.sh);
// tools/summarize_calibration.py turns the two CSVs into profiles/r03_counter_calibration.json.
// Build: hipcc --offload-arch=gfx950 -O3 -o counter_calib counter_calib.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

constexpr int BLOCK = 128;

template <int R>
__global__ __launch_bounds__(BLOCK) void rd_dword(const float *__restrict__ src, long long ld, long long n, float *__restrict__ sink) {
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < R; k++) acc += src[k * ld + i];
    if (acc == 123456.0f) sink[i] = acc;  // never true for the fill pattern: the loads stay, nothing is written
}
template <int W>
__global__ __launch_bounds__(BLOCK) void wr_dword(float *__restrict__ dst, long long ld, long long n, float v) {
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int k = 0; k < W; k++) dst[k * ld + i] = v + (float)k;
}
__global__ __launch_bounds__(BLOCK) void rd_u8(const uint8_t *__restrict__ src, long long ld, long long n, uint8_t *__restrict__ sink) {
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const unsigned v = src[i] | src[ld + i] | src[2 * ld + i];
    if (v == 77u) sink[i] = (uint8_t)v;
}
__global__ __launch_bounds__(BLOCK) void wr_u8(uint8_t *__restrict__ dst, long long ld, long long n, int v) {
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    dst[i] = (uint8_t)v;
    dst[ld + i] = (uint8_t)(v + 1);
    dst[2 * ld + i] = (uint8_t)(v + 2);
}
__global__ __launch_bounds__(BLOCK) void rd_i64(const long long *__restrict__ src, long long n, long long *__restrict__ sink) {
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const long long v = src[i];
    if (v == 0x7777777777ll) sink[i] = v;
}
__global__ __launch_bounds__(BLOCK) void wr_i64(long long *__restrict__ dst, long long n, long long v) {
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    dst[i] = v + i;
}
__global__ __launch_bounds__(BLOCK) void rd_act(const float *__restrict__ src, long long n, float *__restrict__ sink) {
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; k++) acc += src[i * 4 + k];
    if (acc == 123456.0f) sink[i] = acc;
}
// [n][22] rows leave a 128-row tile as 704 float4s, the env kernel's vectorised observation store
__global__ __launch_bounds__(BLOCK) void wr_obs(float *__restrict__ dst, long long n, float v) {
    const long long i0 = (long long)blockIdx.x * BLOCK;
    if (i0 + BLOCK > n) return;  // full tiles only (n is a multiple of 128 here)
    float4 *dst4 = reinterpret_cast<float4 *>(dst + i0 * 22);
    constexpr int VECS = BLOCK * 22 / 4;
#pragma unroll
    for (int it = 0; it < (VECS + BLOCK - 1) / BLOCK; it++) {
        const int L = it * BLOCK + (int)threadIdx.x;
        if (L < VECS) dst4[L] = make_float4(v, v + 1.0f, v + 2.0f, v + (float)L);
    }
}
__global__ __launch_bounds__(BLOCK) void rd_vec16(const float4 *__restrict__ src, long long n, float4 *__restrict__ sink) {
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const float4 v = src[i];
    if (v.x + v.y + v.z + v.w == 123456.0f) sink[i] = v;
}
// the whole per-aircraft pattern of f16_env_kernel<.., CACHED>: reads s 48 + u 16 + tgt 12 + step_count 8 + flags 3
// + action 16 + cache 56 = 159 (+4: u row 4 is never touched) ; writes s 48 + u 16 + tgt 12 + step_count 8 + flags 3 + reward 4 +
// obs 88 + cache 56 = 235.  (tgt is re-written by the kernel although DESIGN's 278 B figure does not count it: 12 B.)
__global__ __launch_bounds__(BLOCK) void mix(float *s, float *u, float *tgt, long long *sc, const uint8_t *fin, uint8_t *fout,
                                             const float *act, float *obs, float *rew, float *cache, long long ld, long long n) {
    __shared__ float tile[BLOCK * 22];
    const long long i0 = (long long)blockIdx.x * BLOCK, i = i0 + threadIdx.x;
    if (i0 + BLOCK > n) return;
    float acc = 0.0f;
    float sv[12], uv[4], tv[3], cv[14];
#pragma unroll
    for (int k = 0; k < 12; k++) sv[k] = s[k * ld + i];
#pragma unroll
    for (int k = 0; k < 4; k++) uv[k] = u[k * ld + i];
#pragma unroll
    for (int k = 0; k < 3; k++) tv[k] = tgt[k * ld + i];
    long long c = sc[i];
    const unsigned f = fin[i] | fin[ld + i] | fin[2 * ld + i];
#pragma unroll
    for (int k = 0; k < 4; k++) acc += act[i * 4 + k];
    float *cb = cache + ((i >> 6) * 14) * 64 + (i & 63);
#pragma unroll
    for (int k = 0; k < 14; k++) cv[k] = cb[k * 64];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k * ld + i] = sv[k] + acc;
#pragma unroll
    for (int k = 0; k < 4; k++) u[k * ld + i] = uv[k] + acc;
#pragma unroll
    for (int k = 0; k < 3; k++) tgt[k * ld + i] = tv[k] + acc;
    sc[i] = c + 1;
    fout[i] = (uint8_t)f;
    fout[ld + i] = (uint8_t)(f + 1);
    fout[2 * ld + i] = (uint8_t)(f + 2);
    rew[i] = acc;
#pragma unroll
    for (int k = 0; k < 14; k++) cb[k * 64] = cv[k] + acc;
#pragma unroll
    for (int k = 0; k < 22; k++) tile[threadIdx.x * 22 + k] = sv[k % 12] + (float)k;
    __syncthreads();
    const float4 *src4 = reinterpret_cast<const float4 *>(tile);
    float4 *dst4 = reinterpret_cast<float4 *>(obs + i0 * 22);
    constexpr int VECS = BLOCK * 22 / 4;
#pragma unroll
    for (int it = 0; it < (VECS + BLOCK - 1) / BLOCK; it++) {
        const int L = it * BLOCK + (int)threadIdx.x;
        if (L < VECS) dst4[L] = src4[L];
    }
}

int main(int argc, char **argv) {
    const long long n = argc > 1 ? atoll(argv[1]) : 1048576;  // a multiple of 128
    const int reps = argc > 2 ? atoi(argv[2]) : 5;
    if (n % 128) {
        fprintf(stderr, "n must be a multiple of 128\n");
        return 1;
    }
    const long long ld = n;
    float *f32 = nullptr, *obs = nullptr, *sink = nullptr, *cache = nullptr, *act = nullptr;
    uint8_t *b0 = nullptr, *b1 = nullptr;
    long long *i64 = nullptr;
    CHECK(hipMalloc(&f32, sizeof(float) * 20 * n));
    CHECK(hipMalloc(&obs, sizeof(float) * 22 * n));
    CHECK(hipMalloc(&sink, sizeof(float) * 4 * n));
    CHECK(hipMalloc(&cache, sizeof(float) * 14 * n));
    CHECK(hipMalloc(&act, sizeof(float) * 4 * n));
    CHECK(hipMalloc(&b0, 3 * n));
    CHECK(hipMalloc(&b1, 3 * n));
    CHECK(hipMalloc(&i64, 8 * n));
    CHECK(hipMemset(f32, 0, sizeof(float) * 20 * n));
    CHECK(hipMemset(obs, 0, sizeof(float) * 22 * n));
    CHECK(hipMemset(cache, 0, sizeof(float) * 14 * n));
    CHECK(hipMemset(act, 0, sizeof(float) * 4 * n));
    CHECK(hipMemset(b0, 0, 3 * n));
    CHECK(hipMemset(b1, 0, 3 * n));
    CHECK(hipMemset(i64, 0, 8 * n));
    const dim3 grid((unsigned)(n / BLOCK)), block(BLOCK);
    printf("n=%lld reps=%d\n", n, reps);
    printf("expected bytes per launch: rd_dword<19> R=%lld  wr_dword<19> W=%lld  rd_dword<14> R=%lld  wr_dword<14> W=%lld  rd_u8 R=%lld  wr_u8 W=%lld  "
           "rd_i64 R=%lld  wr_i64 W=%lld  rd_act R=%lld  wr_obs W=%lld  rd_vec16 R=%lld  mix R=%lld W=%lld\n",
           19 * 4 * n, 19 * 4 * n, 14 * 4 * n, 14 * 4 * n, 3 * n, 3 * n, 8 * n, 8 * n, 16 * n, 88 * n, 16 * n, 159 * n, 235 * n);
    for (int r = 0; r < reps; r++) {
        hipLaunchKernelGGL(rd_dword<19>, grid, block, 0, 0, f32, ld, n, sink);
        hipLaunchKernelGGL(wr_dword<19>, grid, block, 0, 0, f32, ld, n, 1.0f);
        hipLaunchKernelGGL(rd_dword<14>, grid, block, 0, 0, cache, ld, n, sink);
        hipLaunchKernelGGL(wr_dword<14>, grid, block, 0, 0, cache, ld, n, 2.0f);
        hipLaunchKernelGGL(rd_u8, grid, block, 0, 0, b0, ld, n, b1);
        hipLaunchKernelGGL(wr_u8, grid, block, 0, 0, b1, ld, n, 1);
        hipLaunchKernelGGL(rd_i64, grid, block, 0, 0, i64, n, (long long *)sink);
        hipLaunchKernelGGL(wr_i64, grid, block, 0, 0, i64, n, 5ll);
        hipLaunchKernelGGL(rd_act, grid, block, 0, 0, act, n, sink);
        hipLaunchKernelGGL(wr_obs, grid, block, 0, 0, obs, n, 3.0f);
        hipLaunchKernelGGL(rd_vec16, grid, block, 0, 0, (const float4 *)obs, n, (float4 *)sink);
        hipLaunchKernelGGL(mix, grid, block, 0, 0, f32, f32 + 12 * n, f32 + 16 * n, i64, b0, b1, act, obs, sink, cache, ld, n);
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipGetLastError());
    printf("done\n");
    return 0;
}
