/*
 * oracle/f16_oracle.c — CPU restatement (plain C, scalar fp32) of the NeuralPlane F-16 env.step
 * hot path.  TEST INFRASTRUCTURE ONLY — see f16_oracle.h.  Nothing under neuralplane_amd/ links,
 * loads or calls this file.
 *
 * Every function cites the reference file:line it restates (paths relative to the reference
 * root, xuecy22/NeuralPlane @ 2024-12-18).  The reference is eager PyTorch: each Python-level
 * arithmetic operator is ONE fp32 rounding, Python float/int scalars are rounded to fp32 at the
 * point of use, `c / tensor` is `tensor.reciprocal() * c` (torch/_tensor.py __rtruediv__), and
 * `tensor ** 2` is `tensor * tensor`.  This file keeps that operation order literally — compile
 * with -ffp-contract=off (the Makefile does) so the compiler adds no fused operations of its own.
 *
 * Where the reference's result depends on a library implementation rather than on IEEE-754
 * (nn.Linear accumulation order; sin/cos/tan/pow), the oracle follows the numerics spec of
 * DESIGN.md §Numerics, which the HIP kernel implements as well:
 *   - Linear: acc = bias; acc = fmaf(W[j][k], x[k], acc) for k ascending.
 *   - sin/cos/tan/pow: evaluated in fp64 by the fixed operation sequences below (fdlibm
 *     kernels), rounded once to fp32.
 * Parity of the whole against the reference is pinned by tests/golden (see the header).
 */
#include "f16_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* model blob                                                                                  */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
    char name[24];
    uint32_t input_mask, n_linear;
    uint32_t dims[6];
    double in_mean[3], in_std[3];
    double out_mean, out_std;
    uint32_t param_offset, n_params;
} blob_rec; /* 128 bytes, little endian, see tools/export_weights.py */

typedef struct {
    int n_in, n_linear;
    int dims[6];
    int sel[3];           /* which of (alpha, beta, el) feeds input slot i */
    float in_mean[3], in_std[3]; /* per input slot, rounded double->float like torch does */
    float in_rstd[3];            /* RN(1 / in_std): the normalisation divides by a per-model constant (divc_rc) */
    float out_mean, out_std;
    const float *w[4];
    const float *b[4];
    const float *ws[4]; /* the parameters as the fp32 evaluation uses them: rescaled by powers of two (ACT_SHIFT below) */
    const float *bs[4];
    const float *pwl; /* NULL, or t[64], a[64], x0[64], c[64] (blob PWL section) */
} net_t;

/* The airframe as data (f16o_airframe; device twin: csrc/np_f16_device.h::Airframe, built by np_f16_kernels.hip::make_airframe): the
 * reference's literals (F16_dynamics.py:22-35,61-76,114-116; F16_model.py:52-62) rounded to fp32 where the literal expressions round —
 * derived constants folded in double first.  r_* = RN(1 / x) of a divisor. */
typedef struct af_t {
    float g, mass, r_mass, B, S, cbar, Heng;
    float Jy, r_Jy, Jxz, Jz, Jx;
    float xc, cbar_over_B, c1, c2, c3, c4, denom, r_denom;
    float ail_ref, r_ail_ref, rud_ref, r_rud_ref;
    float atm_lapse, atm_exp, rho0;
    float lag_keep, lag_new, thrust_frac, thrust_max, thrust_unit, r_thrust_unit, surf_max[3];
} af_t;

struct f16o_model {
    net_t net[F16O_NUM_NETS];
    float *params;
    float *scaled;
    af_t af; /* the F-16 unless f16o_model_set_airframe said otherwise */
};

/* Numerics spec, "ReLU": hidden activations are carried divided by 2^ACT_SHIFT and the ReLU saturates at 1 there, i.e. at
 * 2^40 in the reference's terms (its own ReLU does not saturate; trained nets on physical inputs stay below 1e4).  The first
 * layer's weights and every hidden bias are multiplied by 2^-40, the output layer's weights by 2^+40: powers of two commute
 * with every fp32 rounding, so below the bound — and above 1.3e-26 — each intermediate is the reference-order value times
 * 2^-40 exactly and the net's output is unchanged.  The load fails if a parameter cannot be rescaled exactly. */
#define ACT_SHIFT 40

static int g_mode = 0;
void f16o_set_mode(int mode) { g_mode = mode; }
int f16o_get_mode(void) { return g_mode; }

/* Division by a constant (numerics spec; device twin: csrc/np_math.h::np_divc).  The reference writes
 * `x / c`; for a constant c with rc = RN(1/c) the sequence q = x*rc, r = fma(-q, c, x), q' = fma(r, rc, q) (Markstein's
 * correction) IS that IEEE quotient — a property of c that f16o_divc_check proves over all significands for every constant
 * used below.  Non-normal q (0, inf, NaN, denormal) is returned as q: exact for 0 / inf / NaN. */
static inline float divc_rc(float x, float c, float rc) {
    if (g_mode & F16O_MODE_DIV_IEEE) return x / c;
    const float q = x * rc;
    const float r = fmaf(-q, c, x);
    const float f = fmaf(r, rc, q);
    return isnormal(q) ? f : q;
}
#define DIVC(x, c) divc_rc((x), (c), (float)(1.0 / (double)(c)))
float f16o_divc(float x, float c) { return DIVC(x, c); }

void f16o_airframe_default(f16o_airframe *a) {
    memset(a, 0, sizeof(*a));
    a->g = 32.17; a->mass = 636.94; a->B = 30.0; a->S = 300.0; a->cbar = 11.32; a->xcgr = 0.35; a->xcg = 0.30; a->Heng = 0.0; /* :61-68 */
    a->Jy = 55814.0; a->Jxz = 982.0; a->Jz = 63100.0; a->Jx = 9496.0;                                                   /* :71-74 */
    a->ail_ref = 21.5; a->rud_ref = 30.0;                                                                                /* :114-115 */
    a->atm_lapse = 0.703e-5; a->atm_exp = 4.14; a->rho0 = 2.377e-3;                                                       /* :22-35 */
    a->lag_keep = 0.9; a->lag_new = 0.1; a->thrust_frac = 0.225; a->thrust_max = 76300.0; a->thrust_unit = 0.3048;        /* F16_model.py:53 */
    a->surf_max[0] = a->surf_max[1] = a->surf_max[2] = 45.0;                                                              /* :54-56 */
}

static float rcp_f(float c) { return (float)(1.0 / (double)c); }

void f16o_model_set_airframe(f16o_model *m, const f16o_airframe *in) {
    f16o_airframe a;
    int zero = 1;
    if (in)
        for (size_t k = 0; k < sizeof(*in); k++)
            if (((const unsigned char *)in)[k]) zero = 0;
    if (zero) f16o_airframe_default(&a);
    else a = *in;
    af_t *d = &m->af;
    d->g = (float)a.g; d->mass = (float)a.mass; d->r_mass = rcp_f(d->mass); d->B = (float)a.B; d->S = (float)a.S; d->cbar = (float)a.cbar;
    d->Heng = (float)a.Heng;
    d->Jy = (float)a.Jy; d->r_Jy = rcp_f(d->Jy); d->Jxz = (float)a.Jxz; d->Jz = (float)a.Jz; d->Jx = (float)a.Jx;
    d->xc = (float)(a.xcgr - a.xcg);                               /* (xcgr - xcg)      :204 */
    d->cbar_over_B = (float)(a.cbar / a.B);                        /* (cbar / B)        :212 */
    d->c1 = (float)(a.Jz * (a.Jz - a.Jy) + a.Jxz * a.Jxz);         /* Jz*(Jz-Jy)+Jxz^2  :225 */
    d->c2 = (float)(a.Jxz * (a.Jx - a.Jy + a.Jz));                 /* Jxz*(Jx-Jy+Jz)         */
    d->c3 = (float)(a.Jz - a.Jx);                                  /* (Jz - Jx)         :226 */
    d->c4 = (float)(a.Jx * (a.Jx - a.Jy) + a.Jxz * a.Jxz);         /* Jx*(Jx-Jy)+Jxz^2  :227 */
    d->denom = (float)(a.Jx * a.Jz - a.Jxz * a.Jxz);               /* :224 */
    d->r_denom = rcp_f(d->denom);
    d->ail_ref = (float)a.ail_ref; d->r_ail_ref = rcp_f(d->ail_ref); d->rud_ref = (float)a.rud_ref; d->r_rud_ref = rcp_f(d->rud_ref);
    d->atm_lapse = (float)a.atm_lapse; d->atm_exp = (float)a.atm_exp; d->rho0 = (float)a.rho0;
    d->lag_keep = (float)a.lag_keep; d->lag_new = (float)a.lag_new; d->thrust_frac = (float)a.thrust_frac; d->thrust_max = (float)a.thrust_max;
    d->thrust_unit = (float)a.thrust_unit; d->r_thrust_unit = rcp_f(d->thrust_unit);
    for (int k = 0; k < 3; k++) d->surf_max[k] = (float)a.surf_max[k];
}

long f16o_divc_check(float c) {
    const float rc = (float)(1.0 / (double)c);
    long bad = 0;
    for (int binade = 0; binade < 2; binade++) {
        for (uint32_t m = 0; m < (1u << 23); m++) {
            uint32_t bits = ((127u + (uint32_t)binade) << 23) | m;
            float x;
            memcpy(&x, &bits, 4);
            const float q = x * rc, r = fmaf(-q, c, x), f = fmaf(r, rc, q), want = x / c;
            if (memcmp(&f, &want, 4) != 0) bad++;
        }
    }
    return bad;
}

f16o_model *f16o_model_load(const void *blob, size_t nbytes) {
    const unsigned char *p = (const unsigned char *)blob;
    if (nbytes < 16 || memcmp(p, "NPF16MLP", 8) != 0) return NULL;
    uint32_t ver, nn;
    memcpy(&ver, p + 8, 4);
    memcpy(&nn, p + 12, 4);
    if ((ver != 1 && ver != 2) || nn != F16O_NUM_NETS) return NULL;
    size_t hdr = 16 + (size_t)nn * sizeof(blob_rec);
    if (nbytes < hdr) return NULL;
    f16o_model *m = (f16o_model *)calloc(1, sizeof(*m));
    size_t pbytes = nbytes - hdr, n_par_total = 0;
    m->params = (float *)malloc(pbytes);
    memcpy(m->params, p + hdr, pbytes);
    for (uint32_t i = 0; i < nn; i++) {
        blob_rec r;
        memcpy(&r, p + 16 + (size_t)i * sizeof(blob_rec), sizeof(r));
        net_t *t = &m->net[i];
        t->n_linear = (int)r.n_linear;
        if (t->n_linear < 2 || t->n_linear > 4) goto bad;
        for (int k = 0; k < 6; k++) t->dims[k] = (int)r.dims[k];
        t->n_in = t->dims[0];
        int slot = 0;
        for (int k = 0; k < 3; k++)
            if (r.input_mask & (1u << k)) {
                t->sel[slot] = k;
                t->in_mean[slot] = (float)r.in_mean[k];
                t->in_std[slot] = (float)r.in_std[k];
                t->in_rstd[slot] = (float)(1.0 / (double)t->in_std[slot]);
                slot++;
            }
        if (slot != t->n_in) goto bad;
        t->out_mean = (float)r.out_mean;
        t->out_std = (float)r.out_std;
        size_t off = r.param_offset, used = 0;
        for (int l = 0; l < t->n_linear; l++) {
            int in = t->dims[l], out = t->dims[l + 1];
            if (in < 1 || in > 20 || out < 1 || out > 20) goto bad;
            t->w[l] = m->params + off + used;
            used += (size_t)in * out;
            t->b[l] = m->params + off + used;
            used += (size_t)out;
        }
        if (used != r.n_params || (off + used) * 4 > pbytes) goto bad;
        if (off + used > n_par_total) n_par_total = off + used;
    }
    m->scaled = (float *)malloc(n_par_total * sizeof(float));
    for (uint32_t i = 0; i < nn; i++) {
        net_t *t = &m->net[i];
        for (int l = 0; l < t->n_linear; l++) {
            int in = t->dims[l], out = t->dims[l + 1];
            const int last = (l + 1 == t->n_linear);
            const int ew = l == 0 ? -ACT_SHIFT : (last ? ACT_SHIFT : 0), eb = last ? 0 : -ACT_SHIFT;
            float *ws = m->scaled + (t->w[l] - m->params), *bs = m->scaled + (t->b[l] - m->params);
            for (int q = 0; q < in * out; q++) {
                ws[q] = ldexpf(t->w[l][q], ew);
                if (t->w[l][q] != 0.0f && !(isnormal(ws[q]) && ldexpf(ws[q], -ew) == t->w[l][q])) goto bad;
            }
            for (int q = 0; q < out; q++) {
                bs[q] = ldexpf(t->b[l][q], eb);
                if (t->b[l][q] != 0.0f && !(isnormal(bs[q]) && ldexpf(bs[q], -eb) == t->b[l][q])) goto bad;
            }
            t->ws[l] = ws;
            t->bs[l] = bs;
        }
    }
    if (ver >= 2) { /* PWL section: "PWL1", n_tables, seg_cap, then {net_index, n_segments, t, a, x0, c} */
        const unsigned char *q = p + hdr + n_par_total * 4;
        const unsigned char *end = p + nbytes;
        uint32_t nt, cap;
        if (q + 12 > end || memcmp(q, "PWL1", 4) != 0) goto bad;
        memcpy(&nt, q + 4, 4);
        memcpy(&cap, q + 8, 4);
        if (cap != 64) goto bad;
        q += 12;
        for (uint32_t k = 0; k < nt; k++) {
            uint32_t idx, nseg;
            if (q + 8 + 4 * 64 * 4 > end) goto bad;
            memcpy(&idx, q, 4);
            memcpy(&nseg, q + 4, 4);
            if (idx >= nn || m->net[idx].n_in != 1 || nseg < 1 || nseg > 64) goto bad;
            m->net[idx].pwl = m->params + (q + 8 - (p + hdr)) / 4;
            q += 8 + 4 * 64 * 4;
        }
    }
    f16o_model_set_airframe(m, NULL); /* the F-16 literals */
    return m;
bad:
    f16o_model_free(m);
    return NULL;
}

void f16o_model_free(f16o_model *m) {
    if (!m) return;
    free(m->params);
    free(m->scaled);
    free(m);
}

/* ------------------------------------------------------------------------------------------ */
/* numerics spec: elementary functions                                   */
/* ------------------------------------------------------------------------------------------ */

/* fdlibm (Sun Microsystems, freely redistributable) k_sin.c / k_cos.c / e_rem_pio2.c / e_log.c
 * constants. */
static const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                    S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                    S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
static const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                    C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                    C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
static const double INVPIO2 = 6.36619772367581382433e-01, PIO2_1 = 1.57079632673412561417e+00,
                    PIO2_2 = 6.07710050630396597660e-11, PIO2_3 = 2.02226624871116645580e-21,
                    PIO2_3T = 8.47842766036889956997e-32;
static const double TWO_PI_D = 6.283185307179586476925;

static void sincos_d(double x, double *sn, double *cs) {
    if (!(fabs(x) < 1073741824.0)) x = fmod(x, TWO_PI_D); /* exact; inf/NaN -> NaN */
    double k = rint(x * INVPIO2);
    double r = fma(-k, PIO2_1, x);
    r = fma(-k, PIO2_2, r);
    r = fma(-k, PIO2_3, r);
    r = fma(-k, PIO2_3T, r);
    double z = r * r;
    double p = fma(z, S6, S5);
    p = fma(z, p, S4);
    p = fma(z, p, S3);
    p = fma(z, p, S2);
    p = fma(z, p, S1);
    double sr = fma(z * r, p, r);
    double q = fma(z, C6, C5);
    q = fma(z, q, C4);
    q = fma(z, q, C3);
    q = fma(z, q, C2);
    q = fma(z, q, C1);
    double cr = fma(z * z, q, fma(z, -0.5, 1.0));
    int n = (int)k & 3; /* |k| < 2^30 here (NaN converts to an unspecified int; result is NaN anyway) */
    if (k != k) n = 0;
    double s_, c_;
    switch (n) {
    case 0: s_ = sr; c_ = cr; break;
    case 1: s_ = cr; c_ = -sr; break;
    case 2: s_ = -sr; c_ = -cr; break;
    default: s_ = -cr; c_ = sr; break;
    }
    *sn = s_;
    *cs = c_;
}

void f16o_sincos(float x, float *s, float *c) {
    if (g_mode & F16O_MODE_LIBM) {
        *s = sinf(x);
        *c = cosf(x);
        return;
    }
    double sd, cd;
    sincos_d((double)x, &sd, &cd);
    *s = (float)sd;
    *c = (float)cd;
}

float f16o_tan(float x) {
    if (g_mode & F16O_MODE_LIBM) return tanf(x);
    double sd, cd;
    sincos_d((double)x, &sd, &cd);
    return (float)(sd / cd);
}

static const double LG1 = 6.666666666666735130e-01, LG2 = 3.999999999940941908e-01,
                    LG3 = 2.857142874366239149e-01, LG4 = 2.222219843214978396e-01,
                    LG5 = 1.818357216161805012e-01, LG6 = 1.531383769920937332e-01,
                    LG7 = 1.479819860511658591e-01;
static const double INV_LN2 = 1.44269504088896338700e+00, LN2 = 6.93147180559945286227e-01;

/* log2(x) in fp64, fixed operation sequence (fdlibm e_log.c kernel on the mantissa). */
static double log2_d(double x) {
    if (x != x || x < 0.0) return NAN;
    if (x == 0.0) return -INFINITY;
    if (x == INFINITY) return INFINITY;
    /* x is a positive finite double (from float arithmetic: always a normal double) */
    uint64_t bits;
    memcpy(&bits, &x, 8);
    int e = (int)((bits >> 52) & 0x7FF) - 1023;
    bits = (bits & 0x000FFFFFFFFFFFFFULL) | 0x3FF0000000000000ULL;
    double mant;
    memcpy(&mant, &bits, 8);
    if (mant > 1.4142135623730951) {
        mant *= 0.5;
        e += 1;
    }
    double f = mant - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * fma(w, fma(w, LG6, LG4), LG2);
    double t2 = z * fma(w, fma(w, fma(w, LG7, LG5), LG3), LG1);
    double R = t2 + t1;
    double hfsq = 0.5 * f * f;
    double lnm = f - (hfsq - s * (hfsq + R));
    return fma(lnm, INV_LN2, (double)e);
}

/* 2^P in fp64, fixed operation sequence. */
static double exp2_d(double P) {
    if (P != P) return NAN;
    if (P > 2000.0) P = 2000.0;
    if (P < -2000.0) P = -2000.0;
    double k = rint(P);
    double t = (P - k) * LN2;
    /* exp(t), |t| <= 0.347: Taylor to degree 13, Horner */
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

/* x^y for the atmosphere model: exp2(y*log2(x)) in fp64. */
static double pow_d(double x, double y) {
    if (x != x || y != y) return NAN;
    if (x < 0.0) return NAN; /* non-integer exponent */
    if (x == 0.0) return y > 0.0 ? 0.0 : INFINITY;
    if (x == INFINITY) return y > 0.0 ? INFINITY : 0.0;
    return exp2_d(y * log2_d(x));
}

float f16o_pow(float x, float y) {
    if (g_mode & F16O_MODE_LIBM) return powf(x, y);
    return (float)pow_d((double)x, (double)y);
}

/* envs/utils/utils.py:144-154  wrap_2PI / wrap_PI.  `angle % (2*pi)` is torch.remainder:
 * fmod (exact) then `+ divisor` when the signs differ. */
static const float TWO_PI_F = (float)(2.0 * 3.141592653589793);
static const float PI_F = (float)3.141592653589793;

float f16o_wrap_pi(float x) {
    float res = fmodf(x, TWO_PI_F);
    if (res != 0.0f && (res < 0.0f)) res = res + TWO_PI_F;
    res = res + ((res < 0.0f) ? TWO_PI_F : 0.0f); /* res += 2*pi*mask1           :146-147 */
    res = res - ((res > PI_F) ? TWO_PI_F : 0.0f); /* res -= 2*pi*mask1           :152-153 */
    return res;
}

/* ------------------------------------------------------------------------------------------ */
/* counter-based RNG of the numerics spec (production mode; the reference uses torch's global  */
/* generator, whose stream is device-specific — parity tests inject the reference's draws).    */
/* ------------------------------------------------------------------------------------------ */

void f16o_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static void rng_block(uint64_t seed, uint64_t call_idx, int64_t row, uint32_t blk, uint32_t out[4]) {
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t ctr[4] = {(uint32_t)(uint64_t)row, (uint32_t)((uint64_t)row >> 32), (uint32_t)call_idx,
                       (((uint32_t)(call_idx >> 32)) & 0x00FFFFFFu) | (blk << 24)};
    f16o_philox4x32(ctr, key, out);
}

void f16o_rng_uniforms(uint64_t seed, uint64_t call_idx, int64_t row, float u8[8]) {
    uint32_t w[4];
    for (uint32_t b = 0; b < 2; b++) {
        rng_block(seed, call_idx, row, b, w);
        for (int i = 0; i < 4; i++) u8[4 * b + i] = (float)(w[i] >> 8) * 5.9604644775390625e-08f; /* 2^-24 */
    }
}

/* Observation noise (numerics spec; device twin: csrc/np_f16_device.h::add_rng_noise).  The reference
 * adds torch.randn_like(obs) * noise_scale (heading_task.py:150-152) from torch's device-specific global generator; the
 * spec's stream is its own: 22 standard normals per (seed, call, global row) from FOUR Philox4x32-10 blocks (counter blocks
 * 2..5) by Box-Muller on 11 pairs.  Every pair takes a 21-bit radius index K1 and a 21-bit direction index K2: pairs 0..7 the
 * top 21 bits of two words, pairs 8..10 are assembled from the low 11 bits of the 16 words (all 512 bits are independent).
 *   u  = (K1 + 0.5) * 2^-21                          in (0, 1): radius^2 = -2 ln u, at most 5.5 sigma
 *   th = ((K2 & 0x3FFFF) + 0.5) * (pi/4) * 2^-18     in (0, pi/4); bits 18, 19, 20 of K2: swap sin/cos, sign of cos, sign of sin
 * -2 ln u, sqrt and sin/cos are explicit fp32 sequences (fma polynomials, ~1e-7 relative) so that host and device agree bit for
 * bit; z = (sqrt(w) * noise_scale) * (cos, sin) is added to the observation with one fma. */
static inline uint32_t f2u(float x) {
    uint32_t b;
    memcpy(&b, &x, 4);
    return b;
}
static inline float u2f(uint32_t b) {
    float x;
    memcpy(&x, &b, 4);
    return x;
}

static float neg2ln_spec(float u) { /* -2 ln(u), u in (0, 1) normal: exponent + degree-7 polynomial on [sqrt(1/2), sqrt(2)) */
    const uint32_t ix = f2u(u) + 0x004AFB0Du; /* 0x3F800000 - 0x3F3504F3 */
    const int e = (int)(ix >> 23) - 127;
    const float m = u2f((ix & 0x007FFFFFu) + 0x3F3504F3u);
    const float f = m - 1.0f;
    float p = fmaf(f, 0.2026811391115188f, -0.3246837854385376f); /* -2 x the coefficients of ln(1 + f) / f */
    p = fmaf(f, p, 0.34494027495384216f);
    p = fmaf(f, p, -0.3979713022708893f);
    p = fmaf(f, p, 0.49940142035484314f);
    p = fmaf(f, p, -0.6667022705078125f);
    p = fmaf(f, p, 1.0000072717666626f);
    p = fmaf(f, p, -1.9999998807907104f);
    return fmaf((float)e, -1.3862943649291992f, f * p);
}

static float sqrt_spec(float w) { /* w in [1e-7, 40]: reciprocal square root by two Newton steps from an exponent-halving seed */
    float y = u2f(0x5F1FFFF9u - (f2u(w) >> 1));
    float t = w * y;
    t = fmaf(-t, y, 2.38924456f);
    y = y * (0.703952253f * t);
    const float h = 0.5f * w;
    t = h * y;
    t = fmaf(-t, y, 1.5f);
    y = y * t;
    return w * y;
}

static void unit_vector_spec(uint32_t k2, float *cs, float *sn) { /* k2: 21 bits */
    const float th = fmaf((float)(k2 & 0x3FFFFu), 2.9960562e-06f, 1.4980281e-06f); /* (k + 0.5) * (pi / 4) / 2^18 */
    const float z = th * th;
    float p = fmaf(z, -0.00019587951828725636f, 0.008332748897373676f);
    p = fmaf(z, p, -0.166666641831398f);
    p = z * p;
    const float s = fmaf(th, p, th);
    float q = fmaf(z, 2.4463830413878895e-05f, -0.001388759003020823f);
    q = fmaf(z, q, 0.04166664928197861f);
    q = fmaf(z, q, -0.5f);
    const float c = fmaf(z, q, 1.0f);
    const float a = (k2 & 0x40000u) ? s : c, b = (k2 & 0x40000u) ? c : s;
    *cs = (k2 & 0x80000u) ? -a : a;
    *sn = (k2 & 0x100000u) ? -b : b;
}

/* 22 index pairs (K1, K2) of one row: four Philox blocks -> 11 x (21 + 21) bits */
static void noise_indices(uint64_t seed, uint64_t call_idx, int64_t row, uint32_t k1[11], uint32_t k2[11]) {
    uint32_t w[16];
    for (uint32_t b = 0; b < 4; b++) rng_block(seed, call_idx, row, 2 + b, w + 4 * b);
    for (int i = 0; i < 8; i++) {
        k1[i] = w[2 * i] >> 11;
        k2[i] = w[2 * i + 1] >> 11;
    }
    for (int j = 0; j < 3; j++) { /* low 11 bits of words 4j .. 4j+3: (11 + 10) bits each */
        k1[8 + j] = ((w[4 * j] & 0x7FFu) << 10) | ((w[4 * j + 1] >> 1) & 0x3FFu);
        k2[8 + j] = ((w[4 * j + 2] & 0x7FFu) << 10) | ((w[4 * j + 3] >> 1) & 0x3FFu);
    }
}

/* o[2i], o[2i+1] += (sqrt(-2 ln u_i) * scale) * (cos, sin)(direction_i), one fma each */
static void add_rng_noise(uint64_t seed, uint64_t call_idx, int64_t row, float scale, float o[F16O_NOBS]) {
    uint32_t k1[11], k2[11];
    noise_indices(seed, call_idx, row, k1, k2);
    for (int i = 0; i < 11; i++) {
        const float u = fmaf((float)k1[i], 4.76837158203125e-07f, 2.384185791015625e-07f); /* (K1 + 0.5) * 2^-21, exact */
        const float rs = sqrt_spec(neg2ln_spec(u)) * scale;
        float cs, sn;
        unit_vector_spec(k2[i], &cs, &sn);
        o[2 * i] = fmaf(rs, cs, o[2 * i]);
        o[2 * i + 1] = fmaf(rs, sn, o[2 * i + 1]);
    }
}

void f16o_rng_normals(uint64_t seed, uint64_t call_idx, int64_t row, float z22[F16O_NOBS]) {
    for (int k = 0; k < F16O_NOBS; k++) z22[k] = 0.0f;
    add_rng_noise(seed, call_idx, row, 1.0f, z22);
}

/* ------------------------------------------------------------------------------------------ */
/* hifi_F16: 43 MLP surrogates — envs/models/F16/hifi_F16_AeroData.py                          */
/*   MLP.forward :25-29, normalize/unnormalize :32-37, per-net wrappers :149-746               */
/* ------------------------------------------------------------------------------------------ */

static float net_eval(const net_t *t, const float in3[3]) {
    float x[20], y[20];
    for (int i = 0; i < t->n_in; i++) x[i] = divc_rc(in3[t->sel[i]] - t->in_mean[i], t->in_std[i], t->in_rstd[i]); /* normalize :32-33; sigma is a per-model constant */
    if ((g_mode & F16O_MODE_PWL) && t->pwl) {
        /* numerics spec, "aero_1d_tables": a ReLU MLP of one input is exactly piecewise linear; binary search
         * over the sorted breakpoints, then y = fma(a, x - x0, c) on the segment */
        const float *tb = t->pwl, *ta = t->pwl + 64, *tx0 = t->pwl + 128, *tc = t->pwl + 192;
        int idx = 0;
        for (int h = 32; h >= 1; h >>= 1)
            if (x[0] >= tb[idx + h - 1]) idx += h;
        float yn = fmaf(ta[idx], x[0] - tx0[idx], tc[idx]);
        return yn * t->out_std + t->out_mean;
    }
    if (g_mode & F16O_MODE_MLP_F64) {
        double xd[20], yd[20];
        for (int i = 0; i < t->n_in; i++) xd[i] = (double)x[i];
        for (int l = 0; l < t->n_linear; l++) {
            int in = t->dims[l], out = t->dims[l + 1];
            for (int j = 0; j < out; j++) {
                double acc = (double)t->b[l][j];
                for (int k = 0; k < in; k++) acc += (double)t->w[l][j * in + k] * xd[k];
                if (l + 1 < t->n_linear) acc = acc > 0.0 ? acc : 0.0;
                yd[j] = acc;
            }
            for (int j = 0; j < out; j++) xd[j] = yd[j];
        }
        return (float)xd[0] * t->out_std + t->out_mean; /* unnormalize :36-37 */
    }
    if (g_mode & F16O_MODE_BIAS_LAST) { /* EXPERIMENT, not the spec (tools/parity_report.py --engine oracle-bias-last): every Linear layer as
                                         * acc = 0; acc = fma(W[j][k], x[k], acc), k ascending; acc + bias — the order of a GEMM with a bias
                                         * epilogue; unscaled parameters, plain ReLU */
        /* F16O_BIAS_LAST_LAYERS (experiments): bit 0 = the first hidden layer, bit 1 = the other hidden layers, bit 2 = the output layer take the bias
         * last; the rest keep acc = bias first.  Unset = all three. */
        static int layers = -1;
        if (layers < 0) {
            const char *e = getenv("F16O_BIAS_LAST_LAYERS");
            layers = e ? atoi(e) : 7;
        }
        for (int l = 0; l < t->n_linear; l++) {
            int in = t->dims[l], out = t->dims[l + 1];
            const int cls = (l + 1 == t->n_linear) ? 4 : (l == 0 ? 1 : 2);
            for (int j = 0; j < out; j++) {
                float acc = (layers & cls) ? 0.0f : t->b[l][j];
                for (int k = 0; k < in; k++) acc = fmaf(t->w[l][j * in + k], x[k], acc);
                if (layers & cls) acc = acc + t->b[l][j];
                if (l + 1 < t->n_linear) acc = acc > 0.0f ? acc : 0.0f;
                y[j] = acc;
            }
            for (int j = 0; j < out; j++) x[j] = y[j];
        }
        return x[0] * t->out_std + t->out_mean;
    }
    for (int l = 0; l + 1 < t->n_linear; l++) { /* hidden layers: acc = bias; acc = fma(W[j][k], x[k], acc), k ascending (spec) */
        int in = t->dims[l], out = t->dims[l + 1];
        for (int j = 0; j < out; j++) {
            float acc = t->bs[l][j];
            for (int k = 0; k < in; k++) acc = fmaf(t->ws[l][j * in + k], x[k], acc);
            acc = acc > 0.0f ? acc : 0.0f; /* ReLU :19 on activations carried / 2^ACT_SHIFT; NaN -> 0; saturates at 1 (spec) */
            acc = acc > 1.0f ? 1.0f : acc;
            y[j] = acc;
        }
        for (int j = 0; j < out; j++) x[j] = y[j];
    }
    { /* output layer in -> 1 (spec): two interleaved partial chains — lo starts at the bias and takes the even inputs (and an
       * odd last one), hi starts at 0 and takes the odd inputs; y = lo + hi.  The reference's order is ATen sgemm's,
       * implementation-defined; this one is a single packed FMA per input pair on gfx950 */
        int l = t->n_linear - 1, in = t->dims[l];
        float lo = t->bs[l][0], hi = 0.0f;
        for (int k = 0; k + 1 < in; k += 2) {
            lo = fmaf(t->ws[l][k], x[k], lo);
            hi = fmaf(t->ws[l][k + 1], x[k + 1], hi);
        }
        if (in & 1) lo = fmaf(t->ws[l][in - 1], x[in - 1], lo);
        x[0] = lo + hi;
    }
    return x[0] * t->out_std + t->out_mean; /* unnormalize :36-37 */
}

/* Non-finite inputs: torch propagates NaN through every Linear/ReLU; `acc > 0 ? acc : 0` does
 * not.  The spec restores the reference's outcome for NaN inputs by poisoning all outputs when
 * any of the three inputs is non-finite ("non-finite inputs"). */
void f16o_aero(const f16o_model *m, float alpha_deg, float beta_deg, float el, float out[F16O_NUM_NETS]) {
    float in3[3] = {alpha_deg, beta_deg, el};
    float chk = (alpha_deg - alpha_deg) + (beta_deg - beta_deg) + (el - el); /* 0 or NaN */
    for (int i = 0; i < F16O_NUM_NETS; i++) {
        float v = net_eval(&m->net[i], in3);
        out[i] = (chk == chk) ? v : NAN;
    }
}

/* indices into the 43-vector = blob order = evaluation order in nlplant (F16_dynamics.py:140-195) */
enum {
    N_Cx, N_Cz, N_Cm, N_Cy, N_Cn, N_Cl,
    N_Cxq, N_Cyr, N_Cyp, N_Czq, N_Clr, N_Clp, N_Cmq, N_Cnr, N_Cnp,
    N_dCx_lef, N_dCz_lef, N_dCm_lef, N_dCy_lef, N_dCn_lef, N_dCl_lef,
    N_dCxq_lef, N_dCyr_lef, N_dCyp_lef, N_dCzq_lef, N_dClr_lef, N_dClp_lef, N_dCmq_lef, N_dCnr_lef, N_dCnp_lef,
    N_dCy_r30, N_dCn_r30, N_dCl_r30,
    N_dCy_a20, N_dCy_a20_lef, N_dCn_a20, N_dCn_a20_lef, N_dCl_a20, N_dCl_a20_lef,
    N_dCnbeta, N_dClbeta, N_dCm, N_eta_el
};

/* ------------------------------------------------------------------------------------------ */
/* F16Dynamics.nlplant — envs/models/F16/F16_dynamics.py:37-228 (atmos :22-35)                  */
/* ------------------------------------------------------------------------------------------ */

static void nlplant_row(const f16o_model *m, const float x[17], float xd[12]) {
    /* constants :61-76 (m->af: the airframe as data; Python-double expressions were folded in double and rounded once, exactly where the
     * reference multiplies them into a tensor — f16o_model_set_airframe) */
    const af_t *af = &m->af;
    const float g = af->g, mass = af->mass, B = af->B, S = af->S, cbar = af->cbar, Heng = af->Heng;
    const float Jy = af->Jy, Jxz = af->Jxz, Jz = af->Jz, Jx = af->Jx;
    const float xc = af->xc, cbar_over_B = af->cbar_over_B;
    const float r2d = (float)(180.0 / 3.141592653589793);              /* :76 */
    const float c1 = af->c1, c2 = af->c2, c3 = af->c3, c4 = af->c4, denom = af->denom;

    float alt = x[2], phi = x[3], theta = x[4], psi = x[5];
    float vt = x[6];
    float alpha = x[7] * r2d, beta = x[8] * r2d; /* degrees :85-86 */
    float P = x[9], Q = x[10], R = x[11];
    float sa, ca, sb, cb, st, ct, sphi, cphi, spsi, cpsi;
    f16o_sincos(x[7], &sa, &ca);
    f16o_sincos(x[8], &sb, &cb);
    f16o_sincos(theta, &st, &ct);
    float tt = f16o_tan(theta);
    f16o_sincos(phi, &sphi, &cphi);
    f16o_sincos(psi, &spsi, &cpsi);

    vt = (float)(vt <= 0.01f) * 0.01f + (float)(vt > 0.01f) * vt; /* :104 */

    float T = x[12], el = x[13], ail = x[14], rud = x[15];
    /* lef = x[16] is identically 0 (F16_model.py:57): dlef = 1 - lef/25 = 1 and `* dlef` is exact */
    float dail = divc_rc(ail, af->ail_ref, af->r_ail_ref), drud = divc_rc(rud, af->rud_ref, af->r_rud_ref); /* :114-115 */

    /* atmos :22-35 (mach, ps are dead) */
    float tfac = 1.0f - af->atm_lapse * alt;
    float rho = af->rho0 * f16o_pow(tfac, af->atm_exp);
    float qbar = (0.5f * rho) * (vt * vt);

    float U = (vt * ca) * cb, V = vt * sb, W = (vt * sa) * cb; /* :129-131 */

    xd[0] = (U * (ct * cpsi) + V * ((sphi * cpsi) * st - cphi * spsi)) + W * ((cphi * st) * cpsi + sphi * spsi);
    xd[1] = (U * (ct * spsi) + V * ((sphi * spsi) * st + cphi * cpsi)) + W * ((cphi * st) * spsi - sphi * cpsi);
    xd[2] = (U * st - V * (sphi * ct)) - W * (cphi * ct);
    xd[3] = P + tt * (Q * sphi + R * cphi);
    xd[4] = Q * cphi - R * sphi;
    xd[5] = (Q * sphi + R * cphi) / ct;

    float c[F16O_NUM_NETS];
    f16o_aero(m, alpha, beta, el, c); /* :140-195 */

    float inv2vt = 1.0f / (2.0f * vt);   /* (2*vt).reciprocal() */
    float c2v = inv2vt * cbar;           /* cbar / (2 * vt)  :197 */
    float b2v = inv2vt * B;              /* B / (2 * vt)     :206 */

    float dXdQ = c2v * (c[N_Cxq] + c[N_dCxq_lef]);
    float Cx_tot = (c[N_Cx] + c[N_dCx_lef]) + dXdQ * Q;
    float dZdQ = c2v * (c[N_Czq] + c[N_dCz_lef]); /* reference uses delta_Cz_lef here (:199) */
    float Cz_tot = (c[N_Cz] + c[N_dCz_lef]) + dZdQ * Q;
    float dMdQ = c2v * (c[N_Cmq] + c[N_dCmq_lef]);
    float Cm_tot = ((((c[N_Cm] * c[N_eta_el] + Cz_tot * xc) + c[N_dCm_lef]) + dMdQ * Q) + c[N_dCm]) + 0.0f;
    float dYdail = c[N_dCy_a20] + c[N_dCy_a20_lef];
    float dYdR = b2v * (c[N_Cyr] + c[N_dCyr_lef]);
    float dYdP = b2v * (c[N_Cyp] + c[N_dCyp_lef]);
    float Cy_tot = ((((c[N_Cy] + c[N_dCy_lef]) + dYdail * dail) + c[N_dCy_r30] * drud) + dYdR * R) + dYdP * P;
    float dNdail = c[N_dCn_a20] + c[N_dCn_a20_lef];
    float dNdR = b2v * (c[N_Cnr] + c[N_dCnr_lef]);
    float dNdP = b2v * (c[N_Cnp] + c[N_dCnp_lef]);
    float Cn_tot = ((((((c[N_Cn] + c[N_dCn_lef]) - (Cy_tot * xc) * cbar_over_B) + dNdail * dail) +
                      c[N_dCn_r30] * drud) + dNdR * R) + dNdP * P) + c[N_dCnbeta] * beta;
    float dLdail = c[N_dCl_a20] + c[N_dCl_a20_lef];
    float dLdR = b2v * (c[N_Clr] + c[N_dClr_lef]);
    float dLdP = b2v * (c[N_Clp] + c[N_dClp_lef]);
    float Cl_tot = (((((c[N_Cl] + c[N_dCl_lef]) + dLdail * dail) + c[N_dCl_r30] * drud) + dLdR * R) + dLdP * P) +
                   c[N_dClbeta] * beta;

    float Udot = (((R * V - Q * W) - g * st) + divc_rc((qbar * S) * Cx_tot, mass, af->r_mass)) + divc_rc(T, mass, af->r_mass);
    float Vdot = ((P * W - R * U) + (g * ct) * sphi) + divc_rc((qbar * S) * Cy_tot, mass, af->r_mass);
    float Wdot = ((Q * U - P * V) + (g * ct) * cphi) + divc_rc((qbar * S) * Cz_tot, mass, af->r_mass);
    xd[6] = ((U * Udot + V * Vdot) + W * Wdot) / vt;
    xd[7] = (U * Wdot - W * Udot) / (U * U + W * W);
    xd[8] = (Vdot * vt - V * xd[6]) / ((vt * vt) * cb);
    float L_tot = ((Cl_tot * qbar) * S) * B;
    float M_tot = ((Cm_tot * qbar) * S) * cbar;
    float N_tot = ((Cn_tot * qbar) * S) * B;
    xd[9] = divc_rc((((Jz * L_tot + Jxz * N_tot) - (c1 * Q) * R) + (c2 * P) * Q) + (Jxz * Q) * Heng, denom, af->r_denom);
    xd[10] = divc_rc(((M_tot + (c3 * P) * R) - Jxz * (P * P - R * R)) - R * Heng, Jy, af->r_Jy);
    xd[11] = divc_rc((((Jx * N_tot + Jxz * L_tot) + (c4 * P) * Q) - (c2 * Q) * R) + (Jx * Q) * Heng, denom, af->r_denom);
}

void f16o_nlplant(const f16o_model *m, int64_t n, const float *x17, float *xdot12) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) nlplant_row(m, x17 + 17 * i, xdot12 + 12 * i);
}

static void xdot_su(const f16o_model *m, const float *s, const float *u, float xd[12]) {
    float x[17];
    memcpy(x, s, 12 * sizeof(float));
    memcpy(x + 12, u, 5 * sizeof(float)); /* get_extended_state: hstack((s,u))  F16_model.py:47-49 */
    nlplant_row(m, x, xd);
}

/* F16Model.get_acceleration — envs/models/F16_model.py:132-148 */
static void acceleration_row(const f16o_model *m, const float *s, const float *u, float a[3]) {
    float xd[12];
    xdot_su(m, s, u, xd);
    float sina, cosa, sinb, cosb;
    f16o_sincos(s[7], &sina, &cosa);
    f16o_sincos(s[8], &sinb, &cosb);
    float vt = s[6];
    float vel_u = (vt * cosb) * cosa, vel_v = vt * sinb, vel_w = (vt * cosb) * sina;
    float u_dot = ((cosb * cosa) * xd[6] - ((vt * sinb) * cosa) * xd[8]) - ((vt * cosb) * sina) * xd[7];
    float v_dot = sinb * xd[6] + (vt * cosb) * xd[8];
    float w_dot = ((cosb * sina) * xd[6] - ((vt * sinb) * sina) * xd[8]) + ((vt * cosb) * cosa) * xd[7];
    a[0] = (u_dot + s[10] * vel_w) - s[11] * vel_v;
    a[1] = (v_dot + s[11] * vel_u) - s[9] * vel_w;
    a[2] = (w_dot + s[9] * vel_v) - s[10] * vel_u;
}

void f16o_get_acceleration(const f16o_model *m, int64_t n, const float *s, const float *u, float *a3) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) acceleration_row(m, s + 12 * i, u + 5 * i, a3 + 3 * i);
}

/* F16Model.get_accels — envs/models/F16_model.py:164-181 (grav = 32.174 here, g = 32.17 in nlplant) */
void f16o_get_accels(const f16o_model *m, int64_t n, const float *s, const float *u, float *n3) {
    const float inv_grav = (float)(1.0 / 32.174), minv_grav = (float)(-1.0 / 32.174);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        float a[3];
        acceleration_row(m, s + 12 * i, u + 5 * i, a); /* identical sub-expressions :166-178 */
        float st, ct, sr, cr;
        f16o_sincos(s[12 * i + 4], &st, &ct);
        f16o_sincos(s[12 * i + 3], &sr, &cr);
        n3[3 * i + 0] = inv_grav * a[0] + st;
        n3[3 * i + 1] = inv_grav * a[1] - ct * sr;
        n3[3 * i + 2] = minv_grav * a[2] + ct * cr;
    }
}

/* F16Model.get_EAS2TAS — envs/models/F16_model.py:156-162 */
static float eas2tas_row(const af_t *af, float alt) {
    float tfac = 1.0f - af->atm_lapse * alt;
    float e = (1.0f / f16o_pow(tfac, af->atm_exp)) * 1.0f; /* 1 / t == t.reciprocal() * 1 */
    return sqrtf(e);
}

void f16o_get_eas2tas(const f16o_model *m, int64_t n, const float *s, float *out) {
    for (int64_t i = 0; i < n; i++) out[i] = eas2tas_row(&m->af, s[12 * i + 2]);
}

/* F16Model.get_atmos — envs/models/F16_model.py:183-198 (the same arithmetic as F16Dynamics.atmos, F16_dynamics.py:22-35):
 * (mach, qbar, ps) from altitude and airspeed.  `(alt >= 35000.0) * 390 + (alt < 35000.0) * temp` selects; `1.4 * 1716.3` is a
 * Python double product rounded to fp32 when it meets the tensor; pow(vt, 2) is vt * vt in ATen. */
void f16o_get_atmos(const f16o_model *m, int64_t n, const float *s, float *out3) {
    const float c_gas = (float)(1.4 * 1716.3);
    const af_t *af = &m->af;
    for (int64_t i = 0; i < n; i++) {
        float alt = s[12 * i + 2], vt = s[12 * i + 6];
        float tfac = 1.0f - af->atm_lapse * alt;
        float temp = 519.0f * tfac;
        temp = (float)(alt >= 35000.0f) * 390.0f + (float)(alt < 35000.0f) * temp;
        float rho = af->rho0 * f16o_pow(tfac, af->atm_exp);
        float mach = vt / sqrtf(c_gas * temp);
        float qbar = (0.5f * rho) * (vt * vt);
        float ps = (1715.0f * rho) * temp;
        ps = (float)(ps == 0.0f) * 1715.0f + (float)(ps != 0.0f) * ps;
        out3[3 * i + 0] = mach;
        out3[3 * i + 1] = qbar;
        out3[3 * i + 2] = ps;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* env: reset / update / obs / termination / reward                                            */
/* ------------------------------------------------------------------------------------------ */

/* F16Model.reset :33-45 + {Heading,Control,Tracking}Task.reset + env_base.py:92 for ONE flagged row */
static void reset_row(const f16o_cfg *cfg, float *s, float *u, float *tgt, int64_t *step_count, const float ru[5]) {
    for (int k = 0; k < 12; k++) s[k] = 0.0f;
    for (int k = 0; k < 5; k++) u[k] = 0.0f;
    s[2] = ru[0] * (float)(cfg->max_altitude - cfg->min_altitude) + (float)cfg->min_altitude; /* :41 */
    s[6] = ru[1] * (float)(cfg->max_vt - cfg->min_vt) + (float)cfg->min_vt;                   /* :42 */
    u[0] = (float)cfg->init_T;                                                                /* :43 */
    if (cfg->task == F16O_TASK_HEADING) { /* envs/tasks/heading_task.py:49-69 */
        tgt[0] = s[2] + 1000.0f;
        tgt[1] = f16o_wrap_pi(s[5] + (float)(2.0 * 3.141592653589793 / 3.0));
        tgt[2] = s[6] + 0.0f;
    } else if (cfg->task == F16O_TASK_CONTROL) { /* envs/tasks/control_task.py:49-68 */
        float dp = (2.0f * (ru[2] - 0.5f)) * (float)cfg->max_pitch_increment;
        float dh = (2.0f * (ru[3] - 0.5f)) * (float)cfg->max_heading_increment;
        float dv = (2.0f * (ru[4] - 0.5f)) * (float)cfg->max_velocities_u_increment;
        tgt[0] = f16o_wrap_pi(s[4] + dp);
        tgt[1] = f16o_wrap_pi(s[5] + dh);
        tgt[2] = s[6] + dv;
    } else { /* envs/tasks/tracking_task.py:48-71 */
        float dist = ru[2] * (float)(cfg->max_distance - cfg->min_distance) + (float)cfg->min_distance;
        float th1 = (ru[3] * PI_F) / 3.0f - (float)(3.141592653589793 / 6.0);
        float th2 = (ru[4] * PI_F) / 3.0f - (float)(3.141592653589793 / 6.0);
        float s1, c1, s2, c2;
        f16o_sincos(th1, &s1, &c1);
        f16o_sincos(th2, &s2, &c2);
        tgt[0] = s[0] + (dist * c1) * c2;
        tgt[1] = s[1] + (dist * c1) * s2;
        tgt[2] = s[2] + dist * s1;
    }
    *step_count = 0;
}

/* 22-float observation before noise: heading_task.py:71-152, control_task.py:70-152,
 * tracking_task.py:73-155 (identical except slots 0..2) */
static void obs_row(const af_t *af, const f16o_cfg *cfg, const float *s, const float *u, const float *tgt, float *o) {
    float alt = s[2], roll = s[3], pitch = s[4], heading = s[5], vt = s[6];
    float eas2tas = eas2tas_row(af, alt);
    float TAS = vt + (float)cfg->airspeed * 1.0f; /* get_TAS :96-97 */
    float EAS = TAS / eas2tas;                    /* get_EAS :99-103 */
    if (cfg->task == F16O_TASK_HEADING) {
        o[0] = DIVC((alt - tgt[0]) * 0.3048f, 1000.0f);
        o[1] = f16o_wrap_pi(heading - tgt[1]);
        o[2] = DIVC((vt - tgt[2]) * 0.3048f, 340.0f);
    } else if (cfg->task == F16O_TASK_CONTROL) {
        o[0] = f16o_wrap_pi(pitch - tgt[0]);
        o[1] = f16o_wrap_pi(heading - tgt[1]);
        o[2] = DIVC((vt - tgt[2]) * 0.3048f, 340.0f);
    } else {
        o[0] = DIVC((s[0] - tgt[0]) * 0.3048f, 1000.0f);
        o[1] = DIVC((s[1] - tgt[1]) * 0.3048f, 1000.0f);
        o[2] = DIVC((alt - tgt[2]) * 0.3048f, 1000.0f);
    }
    o[3] = DIVC(alt * 0.3048f, 5000.0f);
    f16o_sincos(roll, &o[4], &o[5]);
    f16o_sincos(pitch, &o[6], &o[7]);
    o[8] = DIVC(EAS * 0.3048f, 340.0f);
    f16o_sincos(s[7], &o[9], &o[10]);
    f16o_sincos(s[8], &o[11], &o[12]);
    o[13] = s[9];
    o[14] = s[10];
    o[15] = s[11];
    o[16] = DIVC(DIVC(u[0], 0.225f), 76300.0f) * 0.3048f;
    o[17] = DIVC(u[1], 45.0f);
    o[18] = DIVC(u[2], 45.0f);
    o[19] = DIVC(u[3], 45.0f);
    o[20] = u[4] / 45.0f;
    o[21] = eas2tas;
}

static void add_noise(const f16o_cfg *cfg, float *o, const float *noise_row, uint64_t seed, uint64_t call_idx,
                      int64_t grow) {
    float scale = (float)cfg->noise_scale;
    if (noise_row) { /* obs + randn_like(obs) * noise_scale */
        for (int k = 0; k < F16O_NOBS; k++) o[k] = o[k] + noise_row[k] * scale;
    } else if (scale != 0.0f) {
        add_rng_noise(seed, call_idx, grow, scale, o);
    }
}

int f16o_reset(const f16o_model *m, const f16o_cfg *cfg, int64_t n, float *s, float *u, float *tgt,
               int64_t *step_count, uint8_t *done, uint8_t *bad, uint8_t *timeout, const float *rand_u,
               const float *noise, uint64_t seed, uint64_t call_idx, int64_t row0, float *obs) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        if (done[i] | bad[i] | timeout[i]) {
            float ru[8];
            if (rand_u) memcpy(ru, rand_u + 5 * i, 5 * sizeof(float));
            else f16o_rng_uniforms(seed, call_idx, row0 + i, ru);
            reset_row(cfg, s + 12 * i, u + 5 * i, tgt + 3 * i, step_count + i, ru);
        }
        done[i] = bad[i] = timeout[i] = 0; /* env_base.py:93-95 */
        if (obs) {
            obs_row(&m->af, cfg, s + 12 * i, u + 5 * i, tgt + 3 * i, obs + F16O_NOBS * i);
            add_noise(cfg, obs + F16O_NOBS * i, noise ? noise + F16O_NOBS * i : NULL, seed, call_idx, row0 + i);
        }
    }
    return 0;
}

/* one torchdiffeq fixed-grid step of the 17-vector x = [s, u] (F16_model.py:64-67); writes s' */
static void integrate_x(const f16o_model *m, int solver, double dt_cfg, const float x[17], float *s) {
    const float dt = (float)dt_cfg - 0.0f; /* t = tensor([0., dt]); dt = t1 - t0 */
    float k1[12];
    nlplant_row(m, x, k1);
    if (solver == F16O_SOLVER_EULER) {
        for (int k = 0; k < 12; k++) s[k] = x[k] + dt * k1[k];
    } else { /* torchdiffeq 0.2.3 rk4_alt_step_func (3/8 rule); parity UNPINNED (see header) */
        const float third = (float)(1.0 / 3.0);
        float y[17], k2[12], k3[12], k4[12];
        memcpy(y, x, sizeof(y));
        for (int k = 0; k < 12; k++) y[k] = x[k] + (dt * k1[k]) * third;
        nlplant_row(m, y, k2);
        for (int k = 0; k < 12; k++) y[k] = x[k] + dt * (k2[k] - k1[k] * third);
        nlplant_row(m, y, k3);
        for (int k = 0; k < 12; k++) y[k] = x[k] + dt * ((k1[k] - k2[k]) + k3[k]);
        nlplant_row(m, y, k4);
        for (int k = 0; k < 12; k++) s[k] = x[k] + (((k1[k] + 3.0f * (k2[k] + k3[k])) + k4[k]) * dt) * 0.125f;
    }
}

/* F16Model.update — envs/models/F16_model.py:51-67 (+ integrator, Appendix A.4 of SURVEY.md) */
static void update_row(const f16o_model *m, const f16o_cfg *cfg, float *s, float *u, const float *a_in) {
    float a[4];
    for (int k = 0; k < 4; k++) { /* torch.clamp(action, -1, 1): NaN stays NaN */
        float v = a_in[k];
        v = v < -1.0f ? -1.0f : v;
        v = v > 1.0f ? 1.0f : v;
        a[k] = v;
    }
    float x[17];
    memcpy(x, s, 12 * sizeof(float));
    const af_t *af = &m->af; /* F16_model.py:52-56: T' = 0.9 T + 0.1 a0 * 0.225 * 76300 / 0.3048; el' = 0.9 el + 0.1 a1 * 45 ... */
    x[12] = af->lag_keep * u[0] + divc_rc(((af->lag_new * a[0]) * af->thrust_frac) * af->thrust_max, af->thrust_unit, af->r_thrust_unit);
    x[13] = af->lag_keep * u[1] + (af->lag_new * a[1]) * af->surf_max[0];
    x[14] = af->lag_keep * u[2] + (af->lag_new * a[2]) * af->surf_max[1];
    x[15] = af->lag_keep * u[3] + (af->lag_new * a[3]) * af->surf_max[2];
    x[16] = 0.0f;
    integrate_x(m, cfg->solver, cfg->dt, x, s);
    for (int k = 0; k < 5; k++) u[k] = x[12 + k];
}

/* terminations (envs/termination_conditions/)) + rewards (envs/reward_functions/)) for one row */
static void done_reward_row(const f16o_model *m, const f16o_cfg *cfg, const float *s, const float *u,
                            const float *tgt, int64_t step_count, int done_prev, int bad_prev, int timeout_prev,
                            uint8_t *done_o, uint8_t *bad_o, uint8_t *timeout_o, float *reward_o, uint8_t *reasons_o) {
    /* Overload — overload.py:37-42 */
    float a[3];
    acceleration_row(m, s, u, a);
    float acc = sqrtf((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]);
    int r_over = (acc - (float)cfg->acceleration_limit) > 0.0f;
    /* LowAltitude — low_altitude.py:29-30 */
    int r_low = (s[2] - (float)cfg->altitude_limit) < 0.0f;
    /* HighSpeed / LowSpeed — high_speed.py:29-30, low_speed.py:29-30 */
    float TAS = s[6] + (float)cfg->airspeed * 1.0f;
    float vel = DIVC(TAS * 0.3048f, 340.0f);
    int r_fast = (vel - (float)cfg->max_velocity) >= 0.0f;
    int r_slow = (vel - (float)cfg->min_velocity) <= 0.0f;
    /* ExtremeState — extreme_state.py:32-36 */
    float alpha = DIVC(s[7] * 180.0f, PI_F), beta = DIVC(s[8] * 180.0f, PI_F);
    int r_ext = ((alpha < (float)cfg->min_alpha) | (alpha > (float)cfg->max_alpha)) |
                ((beta < (float)cfg->min_beta) | (beta > (float)cfg->max_beta));
    int bad = (((r_over | r_low) | r_fast) | r_slow) | r_ext;
    /* Unreach{Heading,Posture,Target} — unreach_heading.py:38-53, unreach_posture.py:40-55, unreach_target.py:38-47 */
    const float pi36 = (float)(3.141592653589793 / 36.0);
    int m1 = step_count >= cfg->max_check_interval, m2 = 1, m3, m4, m5;
    float rew;
    if (cfg->task == F16O_TASK_HEADING) {
        m2 = step_count >= cfg->min_check_interval;
        m3 = fabsf(f16o_wrap_pi(s[5] - tgt[1])) >= pi36;
        m4 = fabsf(s[2] - tgt[0]) >= 100.0f;
        m5 = fabsf(s[6] - tgt[2]) >= 20.0f;
        /* HeadingReward — heading_reward.py:26-36 */
        float da = DIVC((s[2] - tgt[0]) * 0.3048f, 1000.0f);
        float dh = DIVC(f16o_wrap_pi(s[5] - tgt[1]), PI_F);
        float dv = DIVC((s[6] - tgt[2]) * 0.3048f, 340.0f);
        rew = (-(da * da) + -(dh * dh)) + -(dv * dv);
    } else if (cfg->task == F16O_TASK_CONTROL) {
        m3 = fabsf(f16o_wrap_pi(s[5] - tgt[1])) >= pi36;
        m4 = fabsf(s[4] - tgt[0]) >= pi36;
        m5 = fabsf(s[6] - tgt[2]) >= 20.0f;
        /* PostureReward — posture_reward.py:26-35 */
        float dp = DIVC(f16o_wrap_pi(s[4] - tgt[0]), PI_F);
        float dh = DIVC(f16o_wrap_pi(s[5] - tgt[1]), PI_F);
        float dv = DIVC((s[6] - tgt[2]) * 0.3048f, 340.0f);
        rew = (-(dp * dp) + -(dh * dh)) + -(dv * dv);
    } else {
        m3 = fabsf(s[0] - tgt[0]) >= 100.0f;
        m4 = fabsf(s[1] - tgt[1]) >= 100.0f;
        m5 = fabsf(s[2] - tgt[2]) >= 100.0f;
        /* PositionReward — position_reward.py:26-34 */
        float dn = DIVC((s[0] - tgt[0]) * 0.3048f, 1000.0f);
        float de = DIVC((s[1] - tgt[1]) * 0.3048f, 1000.0f);
        float da = DIVC((s[2] - tgt[2]) * 0.3048f, 1000.0f);
        rew = 0.1f * ((-(dn * dn) + -(de * de)) + -(da * da));
    }
    int off = (m3 | m4) | m5;
    int r_unreach = m1 & off, r_reach = ((!off) & (!m1)) & m2;
    if (reasons_o) /* which condition fired at this state: what each condition class prints (torch.sum(mask)) */
        *reasons_o = (uint8_t)(r_over | (r_low << 1) | (r_fast << 2) | (r_slow << 3) | (r_ext << 4) | (r_unreach << 5) | (r_reach << 6));
    bad |= r_unreach;
    int done = r_reach;
    /* BaseEnv.done (env_base.py:70-75): self.is_done = self.is_done + done, ... — the env flags accumulate
     * until the next reset(); inside BaseEnv.step they were just cleared, inside PlanningEnv.step's 50
     * iterations (planning_env.py:153-176) they are not */
    done |= done_prev;
    bad |= bad_prev;
    *done_o = (uint8_t)done;
    *bad_o = (uint8_t)bad;
    *timeout_o = (uint8_t)(timeout_prev != 0);
    /* BaseTask.get_reward task_base.py:70-73: zeros += target reward; += EventDriven (int64 -> float)
     * event_driven_reward.py:28 with the env's accumulated flags */
    rew = 0.0f + rew;
    rew = rew + (float)(-200 * bad + 200 * done);
    *reward_o = rew;
}

static int step_impl(const f16o_model *m, const f16o_cfg *cfg, int64_t n, float *s, float *u, float *tgt,
                     int64_t *step_count, uint8_t *done, uint8_t *bad, uint8_t *timeout, const float *action,
                     int64_t act_stride, const float *rand_u, const float *noise, uint64_t seed, uint64_t call_idx,
                     int64_t row0, float *obs, float *reward, int inner) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        float *si = s + 12 * i, *ui = u + 5 * i, *ti = tgt + 3 * i;
        const int flagged = done[i] | bad[i] | timeout[i];
        const int dp = inner ? done[i] : 0, bp = inner ? bad[i] : 0, tp = inner ? timeout[i] : 0;
        if (flagged && !inner) { /* self.reset()  env_base.py:100 */
            float ru[8];
            if (rand_u) memcpy(ru, rand_u + 5 * i, 5 * sizeof(float));
            else f16o_rng_uniforms(seed, call_idx, row0 + i, ru);
            reset_row(cfg, si, ui, ti, step_count + i, ru);
        }
        float keep[12];
        memcpy(keep, si, sizeof(keep));
        update_row(m, cfg, si, ui, action + act_stride * i); /* :101 */
        if (inner && flagged) memcpy(si, keep, sizeof(keep)); /* planning_env.py:162-166: s[reset] = recent_s[reset] */
        step_count[i] += 1;                                  /* :102 */
        obs_row(&m->af, cfg, si, ui, ti, obs + F16O_NOBS * i); /* :103 */
        add_noise(cfg, obs + F16O_NOBS * i, noise ? noise + F16O_NOBS * i : NULL, seed, call_idx, row0 + i);
        done_reward_row(m, cfg, si, ui, ti, step_count[i], dp, bp, tp, done + i, bad + i, timeout + i, reward + i, NULL); /* :105-106 */
    }
    return 0;
}

int f16o_step(const f16o_model *m, const f16o_cfg *cfg, int64_t n, float *s, float *u, float *tgt,
              int64_t *step_count, uint8_t *done, uint8_t *bad, uint8_t *timeout, const float *action,
              int64_t act_stride, const float *rand_u, const float *noise, uint64_t seed, uint64_t call_idx,
              int64_t row0, float *obs, float *reward) {
    return step_impl(m, cfg, n, s, u, tgt, step_count, done, bad, timeout, action, act_stride, rand_u, noise, seed, call_idx,
                     row0, obs, reward, 0);
}

/* one of the 50 low-level iterations of PlanningEnv.step — envs/planning_env.py:153-176 */
int f16o_step_inner(const f16o_model *m, const f16o_cfg *cfg, int64_t n, float *s, float *u, float *tgt,
                    int64_t *step_count, uint8_t *done, uint8_t *bad, uint8_t *timeout, const float *action,
                    int64_t act_stride, const float *noise, uint64_t seed, uint64_t call_idx, int64_t row0,
                    float *obs, float *reward) {
    return step_impl(m, cfg, n, s, u, tgt, step_count, done, bad, timeout, action, act_stride, NULL, noise, seed, call_idx,
                     row0, obs, reward, 1);
}

/* F16Model.update(action) on its own — envs/models/F16_model.py:51-67 (called directly by envs/planning_env.py:160 and the reference's
 * example/quick_start.ipynb): clamp, control lag, one integrator step for EVERY row; flags, counters and targets are not its business */
int f16o_update(const f16o_model *m, const f16o_cfg *cfg, int64_t n, float *s, float *u, const float *action, int64_t act_stride) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) update_row(m, cfg, s + 12 * i, u + 5 * i, action + act_stride * i);
    return 0;
}

/* F16Model.reset(env) on its own — envs/models/F16_model.py:33-45: state and controls of the rows the env has flagged are re-initialised
 * (two uniform draws: altitude, vt); targets, step counters and the flags themselves stay (they are task.reset's / BaseEnv.reset's) */
int f16o_model_reset(const f16o_cfg *cfg, int64_t n, float *s, float *u, const uint8_t *done, const uint8_t *bad, const uint8_t *timeout,
                     const float *rand_u, uint64_t seed, uint64_t call_idx, int64_t row0) {
    for (int64_t i = 0; i < n; i++) {
        if (!(done[i] | bad[i] | timeout[i])) continue;
        float ru[8], tgt[3];
        int64_t sc;
        if (rand_u) memcpy(ru, rand_u + 5 * i, 5 * sizeof(float));
        else f16o_rng_uniforms(seed, call_idx, row0 + i, ru);
        reset_row(cfg, s + 12 * i, u + 5 * i, tgt, &sc, ru); /* the model's part of it: s, u */
    }
    return 0;
}

/* per-condition termination bits at the given (post-step) state: bit 0 overload, 1 low_altitude, 2 high_speed, 3 low_speed,
 * 4 extreme_state, 5 unreach_* (bad), 6 target reached (done) — what the condition classes count and print */
void f16o_termination_reasons(const f16o_model *m, const f16o_cfg *cfg, int64_t n, const float *s, const float *u,
                              const float *tgt, const int64_t *step_count, uint8_t *reasons) {
    for (int64_t i = 0; i < n; i++) {
        uint8_t d, b, t;
        float r;
        done_reward_row(m, cfg, s + 12 * i, u + 5 * i, tgt + 3 * i, step_count[i], 0, 0, 0, &d, &b, &t, &r, reasons + i);
    }
}

/* PlanningEnv.low_level_obs — envs/planning_env.py:60-142: ControlTask-style observation, caller's targets, no noise */
void f16o_lowlevel_obs(const f16o_model *m, const f16o_cfg *cfg, int64_t n, const float *s, const float *u, const float *tgt3, float *obs) {
    f16o_cfg c = *cfg;
    c.task = F16O_TASK_CONTROL;
    for (int64_t i = 0; i < n; i++) obs_row(&m->af, &c, s + 12 * i, u + 5 * i, tgt3 + 3 * i, obs + F16O_NOBS * i);
}

void f16o_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int f16o_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

#include "f16_combat.inc"
#include "f16_actor.inc"
#include "f16_actor_i8.inc"
#include "f16_rollout.inc"
