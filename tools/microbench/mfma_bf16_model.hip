// What arithmetic does v_mfma_f32_16x16x32_bf16 perform on gfx950?  (Round 4 probe for the next lever of PlanningEnv's controller: its
// 1 184 K = 1 fp32 MFMA steps per call sit on the fp32 matrix roof; a bf16 x 3 split of fp32 operands would run on the 16 x faster bf16
// pipe — usable under this repo's "HIP == oracle bit for bit" rule only if the instruction's accumulation can be restated exactly in C.)
// The kernel runs ONE instruction per trial on operands chosen by the host and dumps A, B, C, D bit patterns; the models are compared on
// the CPU with exact rational arithmetic (tools/microbench/mfma_bf16_model.py).
// hipcc --offload-arch=gfx950 -O3 mfma_bf16_model.hip -o mfma_bf16_model && ./mfma_bf16_model out.bin [trials]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// per trial: A[16][32] u16, B[32][16] u16, C[16][16] f32 -> D[16][16] f32
__global__ void one_mfma(const uint16_t *A, const uint16_t *B, const float *Cm, float *D, int trials) {
    const int l = threadIdx.x;
    for (int t = blockIdx.x; t < trials; t += gridDim.x) {
        const uint16_t *a_ = A + (size_t)t * 512, *b_ = B + (size_t)t * 512;
        const float *c_ = Cm + (size_t)t * 256;
        union { bf16x8 v; uint16_t u[8]; } a, b;
        for (int e = 0; e < 8; e++) {
            a.u[e] = a_[(l & 15) * 32 + 8 * (l >> 4) + e];          // A[row = l & 15][k = 8 (l >> 4) + e]
            b.u[e] = b_[(8 * (l >> 4) + e) * 16 + (l & 15)];        // B[k][col = l & 15]
        }
        f32x4 c;
        for (int r = 0; r < 4; r++) c[r] = c_[((l >> 4) * 4 + r) * 16 + (l & 15)];   // C[row = 4 (l >> 4) + r][col = l & 15]
        f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c, 0, 0, 0);
        for (int r = 0; r < 4; r++) D[(size_t)t * 256 + ((l >> 4) * 4 + r) * 16 + (l & 15)] = d[r];
    }
}

static uint16_t bf16_of(float f) {  // round to nearest even
    uint32_t u; memcpy(&u, &f, 4);
    u += 0x7fff + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}

int main(int argc, char **argv) {
    // mfma_bf16_model out.bin [trials]            random operand classes (the first survey)
    // mfma_bf16_model --in in.bin out.bin         operands from a file written by mfma_bf16_probe_gen.py: int32 trials, A, B, C as below
    if (argc > 3 && !strcmp(argv[1], "--in")) {
        FILE *fi = fopen(argv[2], "rb");
        if (!fi) { printf("cannot read %s\n", argv[2]); return 1; }
        int32_t trials = 0;
        if (fread(&trials, 4, 1, fi) != 1 || trials <= 0) return 1;
        std::vector<uint16_t> A((size_t)trials * 512), B((size_t)trials * 512);
        std::vector<float> Cm((size_t)trials * 256), D((size_t)trials * 256);
        if (fread(A.data(), 2, A.size(), fi) != A.size() || fread(B.data(), 2, B.size(), fi) != B.size() || fread(Cm.data(), 4, Cm.size(), fi) != Cm.size()) return 1;
        fclose(fi);
        uint16_t *dA, *dB; float *dC, *dD;
        CHECK(hipMalloc(&dA, A.size() * 2)); CHECK(hipMalloc(&dB, B.size() * 2)); CHECK(hipMalloc(&dC, Cm.size() * 4)); CHECK(hipMalloc(&dD, D.size() * 4));
        CHECK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dC, Cm.data(), Cm.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(one_mfma, dim3(64), dim3(64), 0, 0, dA, dB, dC, dD, trials);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
        FILE *f = fopen(argv[3], "wb");
        if (!f) { printf("cannot write %s\n", argv[3]); return 1; }
        int32_t hdr[2] = {trials, 0};
        fwrite(hdr, 4, 2, f);
        fwrite(A.data(), 2, A.size(), f); fwrite(B.data(), 2, B.size(), f); fwrite(Cm.data(), 4, Cm.size(), f); fwrite(D.data(), 4, D.size(), f);
        fclose(f);
        printf("wrote %s: %d trials\n", argv[3], trials);
        return 0;
    }
    const char *out = argc > 1 ? argv[1] : "mfma_bf16_model.bin";
    const int trials = argc > 2 ? atoi(argv[2]) : 96;
    std::vector<uint16_t> A((size_t)trials * 512), B((size_t)trials * 512);
    std::vector<float> Cm((size_t)trials * 256), D((size_t)trials * 256);
    std::mt19937 rng(12345);
    std::uniform_real_distribution<float> uni(-1.f, 1.f);
    for (int t = 0; t < trials; t++) {
        const int kind = t % 6;
        for (int i = 0; i < 512; i++) {
            float a = uni(rng), b = uni(rng);
            if (kind == 0) { a = (float)((int)(rng() % 9) - 4); b = (float)((int)(rng() % 9) - 4); }          // small integers: every model agrees (layout check)
            else if (kind == 2 || kind == 3) { a = ldexpf(a, (int)(rng() % 17) - 8); b = ldexpf(b, (int)(rng() % 17) - 8); }   // wide exponent spread
            else if (kind == 4) { a = ldexpf(a, (int)(rng() % 41) - 20); b = ldexpf(b, (int)(rng() % 41) - 20); }             // very wide
            else if (kind == 5) { a = ldexpf(a, -60 - (int)(rng() % 8)); b = ldexpf(b, -60 - (int)(rng() % 8)); }             // products below 2^-126: denormal handling
            A[(size_t)t * 512 + i] = bf16_of(a);
            B[(size_t)t * 512 + i] = bf16_of(b);
        }
        for (int i = 0; i < 256; i++) {
            float c = uni(rng);
            if (kind == 0) c = (float)((int)(rng() % 17) - 8);
            else if (kind == 3) c = ldexpf(c, 10);        // accumulator dominates: products lose low bits
            else if (kind == 4) c = ldexpf(c, (int)(rng() % 41) - 20);
            else if (kind == 5) c = ldexpf(c, -122 - (int)(rng() % 6));
            Cm[(size_t)t * 256 + i] = c;
        }
    }
    uint16_t *dA, *dB; float *dC, *dD;
    CHECK(hipMalloc(&dA, A.size() * 2)); CHECK(hipMalloc(&dB, B.size() * 2)); CHECK(hipMalloc(&dC, Cm.size() * 4)); CHECK(hipMalloc(&dD, D.size() * 4));
    CHECK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dC, Cm.data(), Cm.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(one_mfma, dim3(8), dim3(64), 0, 0, dA, dB, dC, dD, trials);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
    FILE *f = fopen(out, "wb");
    if (!f) { printf("cannot write %s\n", out); return 1; }
    int32_t hdr[2] = {trials, 6};
    fwrite(hdr, 4, 2, f);
    fwrite(A.data(), 2, A.size(), f); fwrite(B.data(), 2, B.size(), f); fwrite(Cm.data(), 4, Cm.size(), f); fwrite(D.data(), 4, D.size(), f);
    fclose(f);
    printf("wrote %s: %d trials\n", out, trials);
    return 0;
}
