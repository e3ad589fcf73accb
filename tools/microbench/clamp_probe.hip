// What does the `clamp` output modifier of v_pk_fma_f32 return (kernel default MODE: IEEE = 1, DX10_CLAMP = 1)?
// Expected: min(max(x, 0), 1) with clamp(NaN) = +0, clamp(-0) = +0 — i.e. a ReLU that saturates at 1.
// hipcc --offload-arch=gfx950 -O2 clamp_probe.hip -o clamp_probe && ./clamp_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
__global__ void k(const float *x, float *y) {
    const int l = threadIdx.x;
    float a = x[l], b = x[l], one = 1.0f, zero = 0.0f;
    float r0, r1;
    // (a * 1 + 0) clamped, both halves
    asm volatile("v_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\tv_mov_b32 v12, %4\n\tv_mov_b32 v13, %4\n\tv_mov_b32 v14, %5\n\tv_mov_b32 v15, %5\n\t"
                 "v_pk_fma_f32 v[16:17], v[10:11], v[12:13], v[14:15] clamp\n\tv_mov_b32 %0, v16\n\tv_mov_b32 %1, v17"
                 : "=v"(r0), "=v"(r1) : "v"(a), "v"(b), "v"(one), "v"(zero) : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17");
    y[2 * l] = r0;
    y[2 * l + 1] = r1;
}
int main() {
    float h[64];
    const float vals[] = {0.5f, -0.5f, 1.0f, 1.5f, 0.0f, -0.0f, INFINITY, -INFINITY, NAN, -NAN, 1e-40f, -1e-40f, 3e38f, 0.99999994f, 1.0000001f, 1.17549435e-38f};
    for (int i = 0; i < 64; i++) h[i] = vals[i % 16];
    float *dx, *dy, out[128];
    hipMalloc(&dx, 256); hipMalloc(&dy, 512);
    hipMemcpy(dx, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dy);
    hipMemcpy(out, dy, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; i++) {
        const float x = vals[i];
        float e = x > 0.0f ? x : 0.0f;          // the reference's ReLU (NaN -> 0, -0 -> +0) ...
        e = e > 1.0f ? 1.0f : e;                // ... saturating at 1
        unsigned ue, u0, u1; memcpy(&ue, &e, 4); memcpy(&u0, &out[2 * i], 4); memcpy(&u1, &out[2 * i + 1], 4);
        printf("x = %-14g clamp -> %-14g %-14g (bits %08x %08x) expected %08x %s\n", x, out[2 * i], out[2 * i + 1], u0, u1, ue, (u0 == ue && u1 == ue) ? "" : "DIFFERENT");
        bad += !(u0 == ue && u1 == ue);
    }
    printf(bad ? "%d differ\n" : "clamp == saturating ReLU on every probe\n", bad);
    return bad != 0;
}
