// np_nets.h — static description of the F-16 aero-coefficient MLPs and of the kernel-order weight
// blob ("KBLOB") that lives in __constant__ memory.
//
// Reference: envs/models/F16/hifi_F16_AeroData.py:40-129 (shapes), :149-746 (which inputs and which
// normalisation row each net uses), model/mean_std.csv (the constants).  NetId order = evaluation
// order in F16Dynamics.nlplant (F16_dynamics.py:140-195) = order of the NPF16MLP v1 asset blob.
//
// The 43 nets have only 7 distinct shapes.  The kernel evaluates them class by class: a class is a
// run of nets with the SAME shape and the SAME (normalised) inputs, stored back to back in the
// KBLOB with a constant stride, so one compact loop body per class serves all its nets (weights
// come in through scalar loads at `class_base + i*stride`).  Within a class the nets needed for the
// body forces (xdot[6..8], i.e. the Overload re-evaluation) come first, so the force-only
// evaluation is the same loops with smaller trip counts.  `delta_Czq_lef` is evaluated by the
// reference but its value is never used (F16_dynamics.py:167-175 skips temp[3]); it is not stored.
//
// KBLOB layout (floats):
//   [0 .. KBLOB_HEADER)        (mean, sigma, RN(1/sigma)) of the 9 distinct input normalisations (+1 pad)
//   per class, per net:        per hidden Linear layer: bias[out] then W^T[in][out] (k-major: the order
//                              the FMA chains consume them; every row padded to an even length);
//                              output layer: (bias, 0), W[0][0..in) padded to even — its two interleaved
//                              partial chains start from that pair (numerics spec);
//                              then out_std, out_mean, zero padding to whole weight-stream groups
// np_pack_kblob (np_f16_kernels.hip) also verifies that every net of a class has the fp32
// normalisation constants of the class, so this static grouping cannot silently disagree with the
// data.
#pragma once
#include <cstdint>

namespace npf16 {

constexpr int NUM_NETS = 43;

// Hidden activations are carried as relu(h) / 2^ACT_SHIFT.  gfx950 has no packed fp32 max, so a ReLU costs one VALU
// instruction per neuron (14 % of the kernel's instructions) — but the `clamp` output modifier of v_pk_fma_f32 is free and is
// exactly min(max(x, 0), 1), NaN -> 0, -0 -> +0, denormals kept (tools/microbench/clamp_probe.hip).  With the activations
// pre-divided by 2^40 the upper bound sits at 1.1e12 — physical inputs produce activations below 1e4 — and every
// intermediate stays an exact power-of-two multiple of the unscaled one down to 1.3e-26.  The numerics spec
// therefore reads: ReLU saturates at 2^40.
constexpr int ACT_SHIFT = 40;

enum NetId : int {
    N_Cx, N_Cz, N_Cm, N_Cy, N_Cn, N_Cl,
    N_Cxq, N_Cyr, N_Cyp, N_Czq, N_Clr, N_Clp, N_Cmq, N_Cnr, N_Cnp,
    N_dCx_lef, N_dCz_lef, N_dCm_lef, N_dCy_lef, N_dCn_lef, N_dCl_lef,
    N_dCxq_lef, N_dCyr_lef, N_dCyp_lef, N_dCzq_lef, N_dClr_lef, N_dClp_lef, N_dCmq_lef, N_dCnr_lef, N_dCnp_lef,
    N_dCy_r30, N_dCn_r30, N_dCl_r30,
    N_dCy_a20, N_dCy_a20_lef, N_dCn_a20, N_dCn_a20_lef, N_dCl_a20, N_dCl_a20_lef,
    N_dCnbeta, N_dClbeta, N_dCm, N_eta_el
};

// distinct (input, mean, std) normalisations in mean_std.csv
enum NormGroup : int {
    G_A_C = 0,     // alpha: mean 35,   std 32.83  (Cx Cz Cm Cn Cl)
    G_A_DAMP = 1,  // alpha: mean 35,   std 31.77  (damping, delta_Cnbeta, delta_Clbeta, delta_Cm)
    G_A_LEF = 2,   // alpha: mean 12.5, std 18.89  (delta_C*_lef, delta_C*_a20_lef)
    G_A_DLEF = 3,  // alpha: mean 12.5, std 18.77  (damping _lef)
    G_A_RUD = 4,   // alpha: mean 35,   std 31.97  (rudder, Cy, a20)
    G_B_C = 5,     // beta : std 17.91             (Cx Cz Cm Cn Cl)
    G_B_O = 6,     // beta : std 17.44             (all other beta nets)
    G_E_C = 7,     // el   : std 14.92             (Cx Cz Cm Cn Cl)
    G_E_ETA = 8,   // el   : std 14.44             (eta_el)
    NUM_NORM_GROUPS = 9,
    G_NONE = -1
};

constexpr int MAX_CLASS_NETS = 12;

struct NetClass {
    int n_in;                  // 1..3 inputs
    int h1, h2, h3;            // hidden widths (h3 == 0: two hidden layers)
    int grp[3];                // normalisation group of each input slot
    int count;                 // nets in the class
    int n_force;               // the first n_force nets feed xdot[6..8]
    int nets[MAX_CLASS_NETS];  // NetId of each member, force-side first
};

// Classes are ordered so that the 36 nets whose inputs are (alpha, beta) only come first (output
// slots 0..35).  The Overload check at the END of step t evaluates the 14 force-side ones among them
// at the new state; those values are exactly what the integrator needs at the START of step t+1
// (same alpha, beta; only `el` changes with the new action), so they are carried across steps in a
// per-aircraft cache instead of being re-evaluated.  The two el-dependent classes
// (Cx Cz Cm Cn Cl, eta_el) follow (slots 36..41).
enum ClassId : int { CL_DAMP, CL_DLEF, CL_D_RUD, CL_D_LEF, CL_E_LEF, CL_E_RUD, CL_F, CL_YPLEF, CL_YA20, CL_C, CL_ETA, NUM_CLASSES };
constexpr int NUM_AB_CLASSES = 9;  // CL_DAMP .. CL_YA20
constexpr int NUM_AB_NETS = 36;    // output slots 0..35
constexpr int NUM_CACHED = 14;     // force-side alpha/beta-only nets: the first n_force of every AB class
// Experiment switch, OFF in every shipped build (round 3, profiles/r03d_trig_cache_ab.log): with NPF16_TRIG_CACHE=1 the cross-step
// cache also carries the trigonometry of the state a step reaches — sin / cos of alpha, beta, theta, phi, tan(theta) and
// (1 - 0.703e-5 alt)^4.14 — which the NEXT step's integrator evaluation otherwise recomputes from the unchanged state (four fp64
// sine / cosine sequences and one pow).  It removes 269 of 8 747 VALU instructions per wave (-3.1 %) for 80 B more traffic per
// aircraft-step — and is SLOWER: +0.6 % at N = 1e6, +1.2 % at 1e7, +9..+47 % on one-generation grids (98 304 - 262 144), where the ten
// extra loads in front of the first instruction and the longer store burst are not hidden.  Bit-identical either way (351 GPU tests).
#ifndef NPF16_TRIG_CACHE
#define NPF16_TRIG_CACHE 0
#endif
constexpr int NUM_CACHED_TRIG = NPF16_TRIG_CACHE ? 10 : 0;   // sa, ca, sb, cb, st, ct, sphi, cphi, tan(theta), pow
// Rows CACHE_KEY0, CACHE_KEY0 + 1 (round 4): the (alpha, beta) the row's 14 coefficients were evaluated at — s[7], s[8] of the state the
// writing step reached, bit for bit.  A step that takes the coefficients from the cache compares the two with the state it was handed: any
// difference (the caller edited the state between steps through a path no version counter sees: `tensor.data[...] = `, a raw-pointer kernel,
// DLPack) makes the wave re-evaluate the coefficients at the state at hand before it goes on — results never depend on who wrote the state.
constexpr int NUM_CACHE_KEYS = 2;
constexpr int CACHE_KEY0 = NUM_CACHED + NUM_CACHED_TRIG;
constexpr int NUM_CACHE_ROWS = CACHE_KEY0 + NUM_CACHE_KEYS;

constexpr NetClass CLASSES[NUM_CLASSES] = {
    /* CL_DAMP  */ {1, 20, 10, 0, {G_A_DAMP, G_NONE, G_NONE}, 12, 4,
                    {N_Cxq, N_Cyr, N_Cyp, N_Czq, N_Clr, N_Clp, N_Cmq, N_Cnr, N_Cnp, N_dCnbeta, N_dClbeta, N_dCm}},
    /* CL_DLEF  */ {1, 20, 10, 0, {G_A_DLEF, G_NONE, G_NONE}, 7, 2,
                    {N_dCxq_lef, N_dCyr_lef, N_dClr_lef, N_dClp_lef, N_dCmq_lef, N_dCnr_lef, N_dCnp_lef}},
    /* CL_D_RUD */ {2, 20, 10, 0, {G_A_RUD, G_B_O, G_NONE}, 2, 1, {N_Cy, N_dCl_a20}},
    /* CL_D_LEF */ {2, 20, 10, 0, {G_A_LEF, G_B_O, G_NONE}, 2, 1, {N_dCx_lef, N_dCl_lef}},
    /* CL_E_LEF */ {2, 20, 10, 5, {G_A_LEF, G_B_O, G_NONE}, 4, 2, {N_dCz_lef, N_dCy_lef, N_dCm_lef, N_dCn_lef}},
    /* CL_E_RUD */ {2, 20, 10, 5, {G_A_RUD, G_B_O, G_NONE}, 4, 1, {N_dCy_r30, N_dCn_r30, N_dCl_r30, N_dCn_a20}},
    /* CL_F     */ {2, 20, 20, 10, {G_A_LEF, G_B_O, G_NONE}, 3, 1, {N_dCy_a20_lef, N_dCn_a20_lef, N_dCl_a20_lef}},
    /* CL_YPLEF */ {1, 20, 10, 5, {G_A_DLEF, G_NONE, G_NONE}, 1, 1, {N_dCyp_lef}},
    /* CL_YA20  */ {2, 20, 10, 10, {G_A_RUD, G_B_O, G_NONE}, 1, 1, {N_dCy_a20}},
    /* CL_C     */ {3, 20, 10, 0, {G_A_C, G_B_C, G_E_C}, 5, 2, {N_Cx, N_Cz, N_Cm, N_Cn, N_Cl}},
    /* CL_ETA   */ {1, 20, 10, 0, {G_E_ETA, G_NONE, G_NONE}, 1, 0, {N_eta_el}},
};

constexpr int NUM_LIVE_NETS = 42;  // 43 minus the dead delta_Czq_lef
// per-lane LDS column of a kernel that evaluates the nets: the 42 coefficient slots.  (Nine more slots for the normalised
// inputs — tools/gen_mlp_asm.py X_IN_LDS — push a 128-lane workgroup to 26 KB: only 5 instead of 6 workgroups per CU, which
// costs the aero_1d_tables mode 10 %; the inputs therefore stay in VGPRs.)
constexpr int NUM_LDS_SLOTS = NUM_LIVE_NETS;

// Exact piecewise-linear tables of the single-input nets (blob PWL section, tools/export_weights.py):
// per net t[64] breakpoints (sorted, +inf padded), a[64], x0[64], c[64]:  y_norm = fma(a[i], x - x0[i], c[i])
// on segment i = #{breakpoints <= x}.  Table order = blob order of the 1-input nets.
constexpr int PWL_SEG = 64;
constexpr int PWL_TABLE_FLOATS = 4 * PWL_SEG;
constexpr int NUM_PWL_TABLES = 22;
constexpr bool net_is_1d(int net) {
    return (net >= N_Cxq && net <= N_Cnp) || (net >= N_dCxq_lef && net <= N_dCnp_lef) || (net >= N_dCnbeta && net <= N_eta_el);
}
constexpr int pwl_index(int net) {  // position among the 1-input nets in NetId order; -1 for the others
    if (!net_is_1d(net)) return -1;
    int k = 0;
    for (int i = 0; i < net; i++) k += net_is_1d(i) ? 1 : 0;
    return k;
}
static_assert(pwl_index(N_eta_el) == NUM_PWL_TABLES - 1 && pwl_index(N_Cxq) == 0 && pwl_index(N_Cx) == -1, "PWL table order");

// parameters (weights + biases) of one net of a class, as in the asset blob
constexpr int class_params(const NetClass &c) {
    int tot = c.n_in * c.h1 + c.h1 + c.h1 * c.h2 + c.h2;
    if (c.h3 > 0) return tot + c.h2 * c.h3 + c.h3 + c.h3 + 1;
    return tot + c.h2 + 1;
}
// KBLOB record of one net (floats), laid out for the asm bodies (tools/gen_mlp_asm.py): rows padded
// to an even length so that every (neuron j, j+1) weight pair is an even-aligned SGPR pair, the
// record padded to whole weight-stream groups (ASM_GROUP floats).
constexpr int ASM_GROUP = 48;  // floats per weight-stream group (3 x s_load_dwordx16), see tools/gen_mlp_asm.py
constexpr int pad2(int n) { return n + (n & 1); }
constexpr int asm_record_len(int n_in, int h1, int h2, int h3) {
    int n = pad2(h1) + n_in * pad2(h1) + pad2(h2) + h1 * pad2(h2);
    if (h3 > 0) n += pad2(h3) + h2 * pad2(h3) + 2 + pad2(h3);
    else n += 2 + pad2(h2);  // output layer: (bias, 0) pair, then W[0][0..in) padded to even
    n += 2;  // out_std, out_mean
    return (n + ASM_GROUP - 1) / ASM_GROUP * ASM_GROUP;
}
// KBLOB stride of one net of a class
constexpr int class_stride(int cl) { return asm_record_len(CLASSES[cl].n_in, CLASSES[cl].h1, CLASSES[cl].h2, CLASSES[cl].h3); }

constexpr int KBLOB_NORM_STRIDE = 3;  // (mean, sigma, RN(1 / sigma)) per normalisation group
constexpr int KBLOB_HEADER = KBLOB_NORM_STRIDE * NUM_NORM_GROUPS + 1;  // padded to an even number of floats

constexpr int class_base(int cl) {  // KBLOB offset of the first net of class cl
    int off = KBLOB_HEADER;
    for (int k = 0; k < cl; k++) off += CLASSES[k].count * class_stride(k);
    return off;
}
constexpr int class_slot(int cl) {  // output slot of the first net of class cl
    int s = 0;
    for (int k = 0; k < cl; k++) s += CLASSES[k].count;
    return s;
}
// + padding: the weight stream prefetches one group (32 floats) past the last record it evaluates
constexpr int KBLOB_FLOATS = class_base(NUM_CLASSES) + 2 * ASM_GROUP;

// ---- second record layout, for the bodies with TWO accumulator sets (pair variant, np_mlp_asm_dual.inc) -------------------------
// There a packed register holds ONE neuron of BOTH sets (lo = set A, hi = set B), every v_pk_fma_f32 broadcasts ONE weight
// (either half of an SGPR pair, op_sel) to both halves, and the first FMA of a chain takes its bias from the OTHER half of the
// same SGPR pair — `v_pk_fma acc, s[w:b], x, s[w:b] op_sel:[0,0,1] op_sel_hi:[0,1,1]` is legal where two different SGPR pairs in
// one VOP3P are not — so no accumulator is ever initialised by a move.  Record of one net (floats, "neuron-major"):
//   per hidden layer (in -> out):  (W[j][0], bias[j]) for j < out, then for k = 1 .. in-1 the row W[.][k] of `out` weights; padded
//                                  to an even length (the next layer's pairs stay even-aligned)
//   output layer (in -> 1):        (W[0][0], bias), W[0][1 .. in), padded to even   [two interleaved partial chains as before:
//                                  lo = bias + even inputs, hi = 0 + odd inputs, y = lo + hi]
//   out_std, out_mean, zero padding to whole weight-stream groups.
// Same values (same 2^-ACT_SHIFT scaling) as the first layout, no header; np_pack_kblob derives it from the first.
constexpr int dual_record_len(int n_in, int h1, int h2, int h3) {
    int n = pad2(h1 * (n_in + 1)) + pad2(h2 * (h1 + 1));
    if (h3 > 0) n += pad2(h3 * (h2 + 1)) + pad2(h3 + 1);
    else n += pad2(h2 + 1);
    n += 2;  // out_std, out_mean
    return (n + ASM_GROUP - 1) / ASM_GROUP * ASM_GROUP;
}
constexpr int dual_class_stride(int cl) { return dual_record_len(CLASSES[cl].n_in, CLASSES[cl].h1, CLASSES[cl].h2, CLASSES[cl].h3); }
constexpr int dual_class_base(int cl) {
    int off = 0;
    for (int k = 0; k < cl; k++) off += CLASSES[k].count * dual_class_stride(k);
    return off;
}
constexpr int KBLOB_DUAL_FLOATS = dual_class_base(NUM_CLASSES) + 2 * ASM_GROUP;
static_assert(class_slot(NUM_CLASSES) == NUM_LIVE_NETS, "every live net belongs to exactly one class");
static_assert(class_slot(NUM_AB_CLASSES) == NUM_AB_NETS, "alpha/beta-only nets occupy slots 0..35");
constexpr int num_force_ab() {
    int n = 0;
    for (int cl = 0; cl < NUM_AB_CLASSES; cl++) n += CLASSES[cl].n_force;
    return n;
}
static_assert(num_force_ab() == NUM_CACHED, "cached nets = force-side alpha/beta-only nets");
// cache row r (0..13) <-> output slot: the r-th force-side AB net in class order
constexpr int cached_slot(int r) {
    for (int cl = 0; cl < NUM_AB_CLASSES; cl++) {
        if (r < CLASSES[cl].n_force) return class_slot(cl) + r;
        r -= CLASSES[cl].n_force;
    }
    return -1;
}

// output slot (position in class order) of a net; -1 for the dead net
constexpr int slot_of(int net) {
    int s = 0;
    for (int cl = 0; cl < NUM_CLASSES; cl++)
        for (int i = 0; i < CLASSES[cl].count; i++, s++)
            if (CLASSES[cl].nets[i] == net) return s;
    return -1;
}
static_assert(slot_of(N_dCzq_lef) == -1 && slot_of(N_Cx) >= 0 && slot_of(N_eta_el) >= 0, "slot table");

}  // namespace npf16
