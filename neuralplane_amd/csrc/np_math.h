// np_math.h — device-side elementary functions of the numerics spec.
//
// The reference evaluates sin/cos/tan/pow through ATen (SLEEF / MKL-VML on CPU, CUDA libdevice on
// GPU): implementation-defined to ~1 ulp.  To make trajectories reproducible bit-for-bit across
// devices and against the CPU oracle, this framework fixes them as explicit fp64 operation
// sequences (fdlibm kernels), rounded ONCE to fp32.  IEEE-754 fma/div/sqrt/rint are correctly
// rounded on gfx950 and on the host, so the sequences give identical bits everywhere.
// Compile with -ffp-contract=off: every fused operation is spelled out here.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace npf16 {

// fdlibm k_sin.c / k_cos.c / e_rem_pio2.c / e_log.c constants (Sun Microsystems, freely redistributable)
#define NPM_S1 (-1.66666666666666324348e-01)
#define NPM_S2 (8.33333333332248946124e-03)
#define NPM_S3 (-1.98412698298579493134e-04)
#define NPM_S4 (2.75573137070700676789e-06)
#define NPM_S5 (-2.50507602534068634195e-08)
#define NPM_S6 (1.58969099521155010221e-10)
#define NPM_C1 (4.16666666666666019037e-02)
#define NPM_C2 (-1.38888888888741095749e-03)
#define NPM_C3 (2.48015872894767294178e-05)
#define NPM_C4 (-2.75573143513906633035e-07)
#define NPM_C5 (2.08757232129817482790e-09)
#define NPM_C6 (-1.13596475577881948265e-11)
#define NPM_INVPIO2 (6.36619772367581382433e-01)
#define NPM_PIO2_1 (1.57079632673412561417e+00)
#define NPM_PIO2_2 (6.07710050630396597660e-11)
#define NPM_PIO2_3 (2.02226624871116645580e-21)
#define NPM_PIO2_3T (8.47842766036889956997e-32)
#define NPM_TWO_PI (6.283185307179586476925)

// ---------------------------------------------------------------------------------------------
// Division by a constant c (a literal of the reference, or a per-context constant such as a normalisation sigma):
//     q = x * rc;  r = fma(-q, c, x);  q' = fma(r, rc, q)        with rc = RN(1 / c) prepared once
// Markstein's correction step: q' is the correctly rounded IEEE quotient x / c.  That is a property of the constant (it can
// fail for a few significands of x for unlucky c): the CPU test suite proves it for every constant on the path over ALL
// 2^24 significands (tests/test_oracle_golden.py::test_constant_divisors_*), exponent-independent as long as nothing under- or
// overflows: |x| >= 2^-100 and q normal.  Zero, infinite, NaN and denormal q are returned as q = x * rc, which is the exact
// IEEE result for 0 / inf / NaN; denormal quotients and |x| < 2^-100 (unreachable for physical quantities) may differ from IEEE
// in the last place — the oracle evaluates the same sequence, so HIP == oracle holds there too (numerics spec).
// 5 VALU instructions instead of the 12 of the IEEE division sequence; ~50 such divisions per aircraft-step.
// ---------------------------------------------------------------------------------------------
#ifndef NPF16_DIVC
#define NPF16_DIVC 1  // 0: plain IEEE division (A/B reference); 9: x * rc, TIMING EXPERIMENT ONLY (wrong last bits)
#endif
__device__ __forceinline__ float np_divc(float x, float c, float rc) {
#if NPF16_DIVC == 0
    return x / c;
#elif NPF16_DIVC == 9
    return x * rc;
#else
    const float q = x * rc;
    const float r = fmaf(-q, c, x);
    const float f = fmaf(r, rc, q);
    return __builtin_isnormal(q) ? f : q;
#endif
}
#define NP_RCP_CONST(c) ((float)(1.0 / (double)(c)))
#define NP_DIVC(x, c) np_divc((x), (c), NP_RCP_CONST(c))

__device__ __forceinline__ void sincos_d(double x, double &sn, double &cs) {
    if (!(fabs(x) < 1073741824.0)) x = fmod(x, NPM_TWO_PI);  // exact remainder; inf/NaN -> NaN (cold path)
    const double k = rint(x * NPM_INVPIO2);
    double r = fma(-k, NPM_PIO2_1, x);
    r = fma(-k, NPM_PIO2_2, r);
    r = fma(-k, NPM_PIO2_3, r);
    r = fma(-k, NPM_PIO2_3T, r);
    const double z = r * r;
    double p = fma(z, NPM_S6, NPM_S5);
    p = fma(z, p, NPM_S4);
    p = fma(z, p, NPM_S3);
    p = fma(z, p, NPM_S2);
    p = fma(z, p, NPM_S1);
    const double sr = fma(z * r, p, r);
    double q = fma(z, NPM_C6, NPM_C5);
    q = fma(z, q, NPM_C4);
    q = fma(z, q, NPM_C3);
    q = fma(z, q, NPM_C2);
    q = fma(z, q, NPM_C1);
    const double cr = fma(z * z, q, fma(z, -0.5, 1.0));
    int n = (k == k) ? ((int)k & 3) : 0;
    const bool swap = n & 1;
    double s_ = swap ? cr : sr;
    double c_ = swap ? sr : cr;
    // n: 0 (s,c) 1 (c,-s) 2 (-s,-c) 3 (-c,s)
    sn = (n & 2) ? -s_ : s_;
    cs = ((n + 1) & 2) ? -c_ : c_;
}

__device__ __forceinline__ void np_sincos(float x, float &s, float &c) {
#if defined(NPF16_EXP) && (NPF16_EXP & 8)  // timing experiment only: hardware approximations instead of the fp64 sequences
    s = __sinf(x);
    c = __cosf(x);
    return;
#endif
    double sd, cd;
    sincos_d((double)x, sd, cd);
    s = (float)sd;
    c = (float)cd;
}

// sin, cos and tan of the same angle (tan = sin/cos in fp64, rounded once)
__device__ __forceinline__ void np_sincostan(float x, float &s, float &c, float &t) {
#if defined(NPF16_EXP) && (NPF16_EXP & 8)
    s = __sinf(x);
    c = __cosf(x);
    t = s / c;
    return;
#endif
    double sd, cd;
    sincos_d((double)x, sd, cd);
    s = (float)sd;
    c = (float)cd;
    t = (float)(sd / cd);
}

#define NPM_LG1 (6.666666666666735130e-01)
#define NPM_LG2 (3.999999999940941908e-01)
#define NPM_LG3 (2.857142874366239149e-01)
#define NPM_LG4 (2.222219843214978396e-01)
#define NPM_LG5 (1.818357216161805012e-01)
#define NPM_LG6 (1.531383769920937332e-01)
#define NPM_LG7 (1.479819860511658591e-01)
#define NPM_INV_LN2 (1.44269504088896338700e+00)
#define NPM_LN2 (6.93147180559945286227e-01)

// log2(x) in fp64, fixed operation sequence (fdlibm e_log.c kernel on the mantissa)
__device__ __forceinline__ double log2_d(double x) {
    if (x != x || x < 0.0) return __builtin_nan("");
    if (x == 0.0) return -__builtin_inf();
    if (x == __builtin_inf()) return __builtin_inf();
    uint64_t bits = (uint64_t)__double_as_longlong(x);
    int e = (int)((bits >> 52) & 0x7FF) - 1023;
    bits = (bits & 0x000FFFFFFFFFFFFFULL) | 0x3FF0000000000000ULL;
    double mant = __longlong_as_double((long long)bits);
    if (mant > 1.4142135623730951) {
        mant *= 0.5;
        e += 1;
    }
    const double f = mant - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * fma(w, fma(w, NPM_LG6, NPM_LG4), NPM_LG2);
    const double t2 = z * fma(w, fma(w, fma(w, NPM_LG7, NPM_LG5), NPM_LG3), NPM_LG1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double lnm = f - (hfsq - s * (hfsq + R));
    return fma(lnm, NPM_INV_LN2, (double)e);
}

// 2^P in fp64, fixed operation sequence
__device__ __forceinline__ double exp2_d(double P) {
    if (P != P) return __builtin_nan("");
    if (P > 2000.0) P = 2000.0;
    if (P < -2000.0) P = -2000.0;
    const double k = rint(P);
    const double t = (P - k) * NPM_LN2;
    double q = 1.0 / 6227020800.0;
    q = fma(q, t, 1.0 / 479001600.0);
    q = fma(q, t, 1.0 / 39916800.0);
    q = fma(q, t, 1.0 / 3628800.0);
    q = fma(q, t, 1.0 / 362880.0);
    q = fma(q, t, 1.0 / 40320.0);
    q = fma(q, t, 1.0 / 5040.0);
    q = fma(q, t, 1.0 / 720.0);
    q = fma(q, t, 1.0 / 120.0);
    q = fma(q, t, 1.0 / 24.0);
    q = fma(q, t, 1.0 / 6.0);
    q = fma(q, t, 0.5);
    q = fma(q, t, 1.0);
    q = fma(q, t, 1.0);
    return ldexp(q, (int)k);
}

// x^y = exp2(y*log2 x) in fp64 (atmosphere model's tfac^4.14)
__device__ __forceinline__ float np_pow(float xf, float yf) {
#if defined(NPF16_EXP) && (NPF16_EXP & 8)
    return __powf(xf, yf);
#endif
    const double x = (double)xf, y = (double)yf;
    if (x != x || y != y) return __builtin_nanf("");
    if (x < 0.0) return __builtin_nanf("");
    if (x == 0.0) return y > 0.0 ? 0.0f : __builtin_inff();
    if (x == (double)__builtin_inff()) return y > 0.0 ? __builtin_inff() : 0.0f;
    return (float)exp2_d(y * log2_d(x));
}

// the same for an exponent known to be positive and finite (the atmosphere's 4.14: np_f16_airframe.atm_exp, checked on the host), handed over as the
// double the sequence works in — np_pow(x, (float)y) without the run-time tests on y and without its conversion
__device__ __forceinline__ float np_pow_posexp(float xf, double y) {
#if defined(NPF16_EXP) && (NPF16_EXP & 8)
    return __powf(xf, (float)y);
#endif
    const double x = (double)xf;
    if (x != x) return __builtin_nanf("");
    if (x < 0.0) return __builtin_nanf("");
    if (x == 0.0) return 0.0f;
    if (x == (double)__builtin_inff()) return __builtin_inff();
    return (float)exp2_d(y * log2_d(x));
}

// ---- pairwise geometry / reward functions of the combat envs (envs/utils/utils.py:156-249) ----
#define NPM_PS0 (1.66666666666666657415e-01)
#define NPM_PS1 (-3.25565818622400915405e-01)
#define NPM_PS2 (2.01212532134862925881e-01)
#define NPM_PS3 (-4.00555345006794114027e-02)
#define NPM_PS4 (7.91534994289814532176e-04)
#define NPM_PS5 (3.47933107596021167570e-05)
#define NPM_QS1 (-2.40339491173441421878e+00)
#define NPM_QS2 (2.02094576023350569471e+00)
#define NPM_QS3 (-6.88283971605453293030e-01)
#define NPM_QS4 (7.70381505559019352791e-02)
#define NPM_PIO2_D (1.57079632679489655800e+00)
#define NPM_PI_D (3.14159265358979311600e+00)

// acos on [-1, 1] in fp64 (fdlibm e_acos.c rational kernel), rounded once
__device__ __forceinline__ float np_acos(float xf) {
    const double x = (double)xf;
    if (x != x) return __builtin_nanf("");
    const double ax = fabs(x);
    const bool small = ax < 0.5;
    const double z = small ? x * x : (1.0 - ax) * 0.5;
    double p = fma(z, NPM_PS5, NPM_PS4);
    p = fma(z, p, NPM_PS3);
    p = fma(z, p, NPM_PS2);
    p = fma(z, p, NPM_PS1);
    p = fma(z, p, NPM_PS0);
    p = z * p;
    double q = fma(z, NPM_QS4, NPM_QS3);
    q = fma(z, q, NPM_QS2);
    q = fma(z, q, NPM_QS1);
    q = fma(z, q, 1.0);
    const double r = p / q;
    if (small) return (float)(NPM_PIO2_D - fma(x, r, x));
    const double s = sqrt(z);
    const double t = 2.0 * fma(s, r, s);
    return (float)(x > 0.0 ? t : NPM_PI_D - t);
}

// atanh(x) = ln((1+x)/(1-x)) / 2 for x in (-1, 1)
__device__ __forceinline__ float np_atanh(float xf) {
    const double x = (double)xf;
    return (float)(0.5 * (log2_d((1.0 + x) / (1.0 - x)) * NPM_LN2));
}

__device__ __forceinline__ float np_exp(float xf) { return (float)exp2_d((double)xf * NPM_INV_LN2); }

// torch.linalg.norm / torch.sum over the 3 components of a row: fp64, rounded once (numerics spec)
__device__ __forceinline__ float np_norm3(float a, float b, float c) {
    const double ad = a, bd = b, cd = c;
    return (float)sqrt((ad * ad + bd * bd) + cd * cd);
}
__device__ __forceinline__ float np_dot3(float a0, float a1, float a2, float b0, float b1, float b2) {
    const float p0 = a0 * b0, p1 = a1 * b1, p2 = a2 * b2;
    return (float)(((double)p0 + (double)p1) + (double)p2);
}

// envs/utils/utils.py:144-154 wrap_PI: torch.remainder (exact fmod, then +divisor on sign mismatch),
// `res += 2*pi*(res<0)`, `res -= 2*pi*(res>pi)`; 2*pi and pi rounded to fp32 at use.
__device__ __forceinline__ float np_wrap_pi(float x) {
    const float TWO_PI_F = 6.28318530717958647692f;
    const float PI_F = 3.14159265358979323846f;
    float res = fmodf(x, TWO_PI_F);
    if (res != 0.0f && res < 0.0f) res = res + TWO_PI_F;
    res = res + ((res < 0.0f) ? TWO_PI_F : 0.0f);
    res = res - ((res > PI_F) ? TWO_PI_F : 0.0f);
    return res;
}

// ---------------------------------------------------------------------------------------------
// Counter-based RNG of the spec: Philox4x32-10, key = seed, counter = (row, call_idx | block<<56)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&out)[4]) {
#ifndef NPF16_PHILOX_ROUNDS
#define NPF16_PHILOX_ROUNDS 10   // the spec's ten rounds; other values are TIMING EXPERIMENTS only (tools/microbench/ab_libs.py: HIP != oracle)
#endif
#pragma unroll
    for (int r = 0; r < NPF16_PHILOX_ROUNDS; r++) {
        // one v_mad_u64_u32 per 32x32->64 product (separate mul_hi / mul_lo would be two quarter-rate instructions)
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        // three-input xor = ONE v_bitop3_b32 (gfx950; truth table 0x96) — left to itself the compiler emits two v_xor_b32 for most rounds
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32(hi1, c1, k0, 0x96), n2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96);
        c0 = n0;
        c1 = lo1;
        c2 = n2;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0;
    out[1] = c1;
    out[2] = c2;
    out[3] = c3;
}

__device__ __forceinline__ void rng_block(uint64_t seed, uint64_t call_idx, int64_t row, uint32_t blk,
                                          uint32_t (&out)[4]) {
    philox4x32_10((uint32_t)(uint64_t)row, (uint32_t)((uint64_t)row >> 32), (uint32_t)call_idx,
                  (((uint32_t)(call_idx >> 32)) & 0x00FFFFFFu) | (blk << 24), (uint32_t)seed, (uint32_t)(seed >> 32),
                  out);
}

// Observation noise of the numerics spec: explicit fp32 sequences — fma polynomials accurate to ~1e-7
// relative — so that host and device agree bit for bit.  They only feed the Box-Muller transform of add_rng_noise.
// -2 ln(u), u in (0, 1) normal: exponent + degree-7 polynomial on [sqrt(1/2), sqrt(2)) (no division)
__device__ __forceinline__ float neg2ln_spec(float u) {
    const uint32_t ix = __float_as_uint(u) + 0x004AFB0Du;  // 0x3F800000 - 0x3F3504F3
    const int e = (int)(ix >> 23) - 127;
    const float m = __uint_as_float((ix & 0x007FFFFFu) + 0x3F3504F3u);
    const float f = m - 1.0f;
    float p = fmaf(f, 0.2026811391115188f, -0.3246837854385376f);  // -2 x the coefficients of ln(1 + f) / f
    p = fmaf(f, p, 0.34494027495384216f);
    p = fmaf(f, p, -0.3979713022708893f);
    p = fmaf(f, p, 0.49940142035484314f);
    p = fmaf(f, p, -0.6667022705078125f);
    p = fmaf(f, p, 1.0000072717666626f);
    p = fmaf(f, p, -1.9999998807907104f);
    return fmaf((float)e, -1.3862943649291992f, f * p);
}

// sqrt(w), w in [1e-7, 40]: w * rsqrt(w), the reciprocal square root by two Newton steps from an exponent-halving seed
// (8e-7 relative; the IEEE sqrt sequence costs 17 instructions on gfx950, this one 11)
__device__ __forceinline__ float sqrt_spec(float w) {
    float y = __uint_as_float(0x5F1FFFF9u - (__float_as_uint(w) >> 1));
    float t = w * y;
    t = fmaf(-t, y, 2.38924456f);
    y = y * (0.703952253f * t);
    const float h = 0.5f * w;
    t = h * y;
    t = fmaf(-t, y, 1.5f);
    y = y * t;
    return w * y;
}

// uniformly distributed unit vector from 21 random bits: 18 bits of angle inside one octant, 3 bits of symmetry
__device__ __forceinline__ void unit_vector_spec(uint32_t k2, float &cs, float &sn) {
    const float th = fmaf((float)(k2 & 0x3FFFFu), 2.9960562e-06f, 1.4980281e-06f);  // (k + 0.5) * (pi / 4) / 2^18
    const float z = th * th;
    float p = fmaf(z, -0.00019587951828725636f, 0.008332748897373676f);
    p = fmaf(z, p, -0.166666641831398f);
    p = z * p;
    const float s = fmaf(th, p, th);
    float q = fmaf(z, 2.4463830413878895e-05f, -0.001388759003020823f);
    q = fmaf(z, q, 0.04166664928197861f);
    q = fmaf(z, q, -0.5f);
    const float c = fmaf(z, q, 1.0f);
    const bool swap = (k2 & 0x40000u) != 0;
    const float a = swap ? s : c, b = swap ? c : s;
    // a, b > 0: the sign bits come straight from bits 19 / 20 of k2 (v_bfi_b32)
    cs = __uint_as_float((__float_as_uint(a) & 0x7FFFFFFFu) | ((k2 << 12) & 0x80000000u));
    sn = __uint_as_float((__float_as_uint(b) & 0x7FFFFFFFu) | ((k2 << 11) & 0x80000000u));
}

// ---- the same three sequences for TWO Box-Muller pairs at once (element k of every vector = pair k): every float operation is a
// packed instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 perform the IEEE operation per half, so each element gets exactly
// the bits of the scalar sequence above); the integer parts stay scalar per element.
typedef float np_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ np_f32x2 np_fma2(np_f32x2 a, np_f32x2 b, np_f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ np_f32x2 np_fma2(np_f32x2 a, float b, float c) { return __builtin_elementwise_fma(a, (np_f32x2)(b), (np_f32x2)(c)); }
__device__ __forceinline__ np_f32x2 np_fma2(np_f32x2 a, np_f32x2 b, float c) { return __builtin_elementwise_fma(a, b, (np_f32x2)(c)); }

__device__ __forceinline__ np_f32x2 neg2ln_spec2(np_f32x2 u) {
    np_f32x2 m, ef;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const uint32_t ix = __float_as_uint(u[k]) + 0x004AFB0Du;
        ef[k] = (float)((int)(ix >> 23) - 127);
        m[k] = __uint_as_float((ix & 0x007FFFFFu) + 0x3F3504F3u);
    }
    const np_f32x2 f = m - (np_f32x2)(1.0f);
    np_f32x2 p = np_fma2(f, 0.2026811391115188f, -0.3246837854385376f);
    p = np_fma2(f, p, 0.34494027495384216f);
    p = np_fma2(f, p, -0.3979713022708893f);
    p = np_fma2(f, p, 0.49940142035484314f);
    p = np_fma2(f, p, -0.6667022705078125f);
    p = np_fma2(f, p, 1.0000072717666626f);
    p = np_fma2(f, p, -1.9999998807907104f);
    return np_fma2(ef, (np_f32x2)(-1.3862943649291992f), f * p);
}

__device__ __forceinline__ np_f32x2 sqrt_spec2(np_f32x2 w) {
    np_f32x2 y;
#pragma unroll
    for (int k = 0; k < 2; k++) y[k] = __uint_as_float(0x5F1FFFF9u - (__float_as_uint(w[k]) >> 1));
    np_f32x2 t = w * y;
    t = np_fma2(-t, y, 2.38924456f);
    y = y * ((np_f32x2)(0.703952253f) * t);
    const np_f32x2 h = (np_f32x2)(0.5f) * w;
    t = h * y;
    t = np_fma2(-t, y, 1.5f);
    y = y * t;
    return w * y;
}

__device__ __forceinline__ void unit_vector_spec2(const uint32_t (&k2)[2], np_f32x2 &cs, np_f32x2 &sn) {
    np_f32x2 kf;
#pragma unroll
    for (int k = 0; k < 2; k++) kf[k] = (float)(k2[k] & 0x3FFFFu);
    const np_f32x2 th = np_fma2(kf, 2.9960562e-06f, 1.4980281e-06f);
    const np_f32x2 z = th * th;
    np_f32x2 p = np_fma2(z, -0.00019587951828725636f, 0.008332748897373676f);
    p = np_fma2(z, p, -0.166666641831398f);
    p = z * p;
    const np_f32x2 s = np_fma2(th, p, th);
    np_f32x2 q = np_fma2(z, 2.4463830413878895e-05f, -0.001388759003020823f);
    q = np_fma2(z, q, 0.04166664928197861f);
    q = np_fma2(z, q, -0.5f);
    const np_f32x2 c = np_fma2(z, q, 1.0f);
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const bool swap = (k2[k] & 0x40000u) != 0;
        const float a = swap ? s[k] : c[k], b = swap ? c[k] : s[k];
        cs[k] = __uint_as_float((__float_as_uint(a) & 0x7FFFFFFFu) | ((k2[k] << 12) & 0x80000000u));
        sn[k] = __uint_as_float((__float_as_uint(b) & 0x7FFFFFFFu) | ((k2[k] << 11) & 0x80000000u));
    }
}

}  // namespace npf16
