// How long does one round trip of K back-to-back s_load_dwordx16 + s_waitcnt lgkmcnt(0) take on gfx950, when the stream is cold
// in the scalar cache (footprint 256 KB, L2-resident) or warm (footprint 4 KB)?  One wave per workgroup, `blocks` workgroups
// (1 = a lone wave on the chip, 256 = one per CU, 2048 = two per SIMD), each with its own region.  Answers whether a deeper
// request (more chunks per wait) amortises the latency or whether misses are served one line after the other.
// hipcc --offload-arch=gfx950 -O3 smem_stream.hip -o smem_stream && ./smem_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int K>
__global__ void stream(const float *base, int region_bytes, int trips, long long *out, float *sink, int misalign_floats) {
    const float *p0 = base + (size_t)blockIdx.x * (region_bytes / 4) + misalign_floats;
    const float *p = p0;
    const float *end = p0 + region_bytes / 4 - 16;
    float acc = 0.f;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < trips; i++) {
        float v;
        if constexpr (K == 1)
            asm volatile("s_load_dwordx16 s[4:19], %1, 0x0\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32 %0, s4" : "=v"(v) : "s"(p)
                         : "s4","s5","s6","s7","s8","s9","s10","s11","s12","s13","s14","s15","s16","s17","s18","s19","memory");
        else if constexpr (K == 2)
            asm volatile("s_load_dwordx16 s[4:19], %1, 0x0\n\ts_load_dwordx16 s[20:35], %1, 0x40\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32 %0, s20" : "=v"(v) : "s"(p)
                         : "s4","s5","s6","s7","s8","s9","s10","s11","s12","s13","s14","s15","s16","s17","s18","s19",
                           "s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","s30","s31","s32","s33","s34","s35","memory");
        else if constexpr (K == 3)
            asm volatile("s_load_dwordx16 s[4:19], %1, 0x0\n\ts_load_dwordx16 s[20:35], %1, 0x40\n\ts_load_dwordx16 s[36:51], %1, 0x80\n\t"
                         "s_waitcnt lgkmcnt(0)\n\tv_mov_b32 %0, s36" : "=v"(v) : "s"(p)
                         : "s4","s5","s6","s7","s8","s9","s10","s11","s12","s13","s14","s15","s16","s17","s18","s19",
                           "s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","s30","s31","s32","s33","s34","s35",
                           "s36","s37","s38","s39","s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","memory");
        else
            asm volatile("s_load_dwordx16 s[4:19], %1, 0x0\n\ts_load_dwordx16 s[20:35], %1, 0x40\n\ts_load_dwordx16 s[36:51], %1, 0x80\n\t"
                         "s_load_dwordx16 s[52:67], %1, 0xc0\n\ts_load_dwordx16 s[68:83], %1, 0x100\n\ts_load_dwordx16 s[84:99], %1, 0x140\n\t"
                         "s_waitcnt lgkmcnt(0)\n\tv_mov_b32 %0, s84" : "=v"(v) : "s"(p)
                         : "s4","s5","s6","s7","s8","s9","s10","s11","s12","s13","s14","s15","s16","s17","s18","s19",
                           "s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","s30","s31","s32","s33","s34","s35",
                           "s36","s37","s38","s39","s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51",
                           "s52","s53","s54","s55","s56","s57","s58","s59","s60","s61","s62","s63","s64","s65","s66","s67",
                           "s68","s69","s70","s71","s72","s73","s74","s75","s76","s77","s78","s79","s80","s81","s82","s83",
                           "s84","s85","s86","s87","s88","s89","s90","s91","s92","s93","s94","s95","s96","s97","s98","s99","memory");
        acc += v;
        p += 16 * K;
        if (p + 16 * K > end) p = p0;
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; sink[blockIdx.x] = acc; }
}

template <int K>
int run(const float *d, long long *out, float *sink, int region, int blocks, int mis = 0) {
    const int trips = 2048;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(stream<K>, dim3(blocks), dim3(64), 0, 0, d, region, trips, out, sink, mis);
        CHECK(hipDeviceSynchronize());
    }
    std::vector<long long> h(blocks);
    CHECK(hipMemcpy(h.data(), out, 8 * blocks, hipMemcpyDeviceToHost));
    double s = 0;
    for (auto c : h) s += (double)c;
    printf("  K=%d chunks per wait: %7.1f cycles per round trip (%5.1f per 64-byte chunk)\n", K, s / blocks / trips, s / blocks / trips / K);
    return 0;
}

int main() {
    const int max_blocks = 2048;
    const size_t bytes = (size_t)max_blocks * 262144;
    float *d; long long *out; float *sink;
    CHECK(hipMalloc(&d, bytes)); CHECK(hipMemset(d, 0, bytes));
    CHECK(hipMalloc(&out, 8 * max_blocks)); CHECK(hipMalloc(&sink, 4 * max_blocks));
    for (int region : {4096, 262144})
        for (int blocks : {1, 256, 2048}) {
            printf("region %d KB per wave (%s), %d waves:\n", region / 1024, region <= 8192 ? "warm in the scalar cache" : "cold, L2 / HBM", blocks);
            if (run<1>(d, out, sink, region, blocks) || run<2>(d, out, sink, region, blocks) || run<3>(d, out, sink, region, blocks) || run<6>(d, out, sink, region, blocks)) return 1;
        }
    // the same with every load starting 48 bytes into a 64-byte line (a record stream behind a 112-byte header)
    for (int region : {4096, 262144})
        for (int blocks : {1, 2048}) {
            printf("MISALIGNED by 48 bytes: region %d KB per wave, %d waves:\n", region / 1024, blocks);
            if (run<1>(d, out, sink, region, blocks, 12) || run<3>(d, out, sink, region, blocks, 12) || run<6>(d, out, sink, region, blocks, 12)) return 1;
        }
    CHECK(hipDeviceSynchronize());
    return 0;
}
