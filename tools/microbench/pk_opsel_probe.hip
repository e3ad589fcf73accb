// What v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 really do on gfx950 when ONE SGPR pair feeds two operands with different op_sel
// (the first FMA of the neuron-major bodies: acc = fma(w, x, bias) with (w, bias) in one pair), and when op_sel picks the odd SGPR.
// hipcc --offload-arch=gfx950 -O2 pk_opsel_probe.hip -o pk_opsel_probe && ./pk_opsel_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(float *out) {
    float r[12];
    asm volatile(
        "s_mov_b32 s4, 2.0\n\t"
        "s_mov_b32 s5, 0x41200000\n\t"      // 10.0
        "v_mov_b32 v2, 0x40400000\n\t"      // 3.0  (set A)
        "v_mov_b32 v3, 0x40a00000\n\t"      // 5.0  (set B)
        "s_nop 4\n\t"
        "v_pk_fma_f32 v[10:11], s[4:5], v[2:3], s[4:5] op_sel:[0,0,1] op_sel_hi:[0,1,1]\n\t"     // expect 16, 20
        "v_pk_fma_f32 v[12:13], s[4:5], v[2:3], v[10:11] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"  // expect 46, 70
        "v_pk_fma_f32 v[14:15], s[4:5], v[2:3], 0 op_sel:[1,0,0] op_sel_hi:[1,1,0]\n\t"          // expect 30, 50
        "v_pk_mul_f32 v[16:17], s[4:5], v[2:3] op_sel:[1,0] op_sel_hi:[1,1]\n\t"                 // expect 30, 50
        "v_pk_add_f32 v[18:19], s[4:5], v[2:3] op_sel:[0,0] op_sel_hi:[0,1]\n\t"                 // expect 5, 7
        "v_pk_fma_f32 v[20:21], s[4:5], v[2:3], v[10:11] op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"  // expect 22, 30
        "v_mov_b32 %0, v10\n\tv_mov_b32 %1, v11\n\tv_mov_b32 %2, v12\n\tv_mov_b32 %3, v13\n\tv_mov_b32 %4, v14\n\tv_mov_b32 %5, v15\n\t"
        "v_mov_b32 %6, v16\n\tv_mov_b32 %7, v17\n\tv_mov_b32 %8, v18\n\tv_mov_b32 %9, v19\n\tv_mov_b32 %10, v20\n\tv_mov_b32 %11, v21\n\t"
        : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]), "=v"(r[8]), "=v"(r[9]), "=v"(r[10]), "=v"(r[11])
        :
        : "s4", "s5", "v2", "v3", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21");
    if (threadIdx.x == 0)
        for (int k = 0; k < 12; k++) out[k] = r[k];
}
int main() {
    float *d, h[12];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const float want[12] = {16, 20, 46, 70, 30, 50, 30, 50, 5, 7, 22, 30};
    const char *what[6] = {"fma(w, x, b) one pair twice", "fma(odd sgpr, x, acc)", "fma(odd sgpr, x, 0)", "mul(odd sgpr, x)", "add(even sgpr, x)", "fma(even sgpr, x, acc)"};
    for (int k = 0; k < 6; k++)
        printf("%-32s got (%g, %g) want (%g, %g) %s\n", what[k], h[2 * k], h[2 * k + 1], want[2 * k], want[2 * k + 1],
               (h[2 * k] == want[2 * k] && h[2 * k + 1] == want[2 * k + 1]) ? "ok" : "MISMATCH");
    return 0;
}
