"""numpy prototype of the controller's block-fixed-point numerics, version 2 (round 5; built as csrc/np_actor_i8.h, restated in oracle/f16_actor_i8.inc):
every Linear with K <= 128 except the 4-output head on the i8 matrix pipe, LayerNorm / quantiser / epilogue in the ACCUMULATOR layout
of v_mfma_i32_32x32x32_i8 (feature f = 32 w + 8 g + 4 h + t: wave w, lane half h, register 4 g + t), the row exponent from a BOUND that
needs no reduction of its own (max |x - mean| rides in the variance exchange), activations in three limbs, weights in four, the nine limb
products of weight >= 2^16 in four class sums.
Judged against what the REFERENCE recorded: tests/golden/actor_kat.npz (4 open-loop calls, bound 2e-5) and the closed loop of
tests/golden/planning_closed_kat.npz driven through the oracle's FDM (bounds of tests/planning_closed.py).  CPU only.
    python tools/microbench/i8v2_numerics.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
f32 = np.float32
MAGIC = f32(12582912.0)          # 1.5 * 2^23


def feat_order():
    """register order of a lane block (w, h): features 32 w + 8 g + 4 h + t, (g, t) ascending; blocks in (w, h) order"""
    return np.array([[32 * w + 8 * g + 4 * h + t for g in range(4) for t in range(4)] for w in range(4) for h in range(2)])   # [8, 16]


ORD = feat_order()


def exponent_of(b):
    """e with b < 2^e, from the float's exponent field; clamped below"""
    bits = np.asarray(b, f32).view(np.uint32)
    return np.maximum(((bits >> 23) & 255).astype(np.int64) - 126, -100)


def pow2(e):
    return np.ldexp(f32(1.0), np.asarray(e, np.int64)).astype(f32)


def quant_rows(x, ex):
    """q = bits(fmaf(x, 2^(22 - ex), MAGIC)) - bits(MAGIC): round-half-even of x * 2^(22 - ex) for |.| < 2^22"""
    t = (x.astype(np.float64) * np.ldexp(1.0, (22 - ex))[:, None] + float(MAGIC)).astype(f32)   # the fma: exact product + one rounding
    return t.view(np.int32).astype(np.int64) - int(MAGIC.view(np.int32))


def limbs(q):
    p = ((q + 0x808080) ^ 0x808080)
    b = lambda k: (((p >> (8 * k)) & 255) ^ 128) - 128     # noqa: E731  signed byte k
    return b(0), b(1), b(2)


WBITS = 29   # weights: sign + 29 bits = FOUR balanced limbs (22 bits relative to the row maximum left the weights 20 x coarser than fp32
             # for the typical entry — the dominant error of the first prototype: closed-loop states 1.0e-4 against 3.0e-5 with exact weights)


def quant_weights(W):
    """per output feature: ew with max|w| < 2^ew, wq = rint(w * 2^(WBITS - ew)) (load time: plain round-half-even)"""
    m = np.max(np.abs(W), axis=1)
    ew = exponent_of(m.astype(f32))
    wq = np.rint(W.astype(np.float64) * np.ldexp(1.0, WBITS - ew)[:, None]).astype(np.int64)
    return wq, ew


def limbs4(q):
    p = ((q + 0x80808080) ^ 0x80808080)
    b = lambda k: (((p >> (8 * k)) & 255) ^ 128) - 128     # noqa: E731
    top = (q + 0x808080) >> 24
    return b(0), b(1), b(2), top


def dense_i8(x, ex, Wq, ew, bias, drop00=True):
    """x: 3 limbs, w: 4 limbs; the nine limb products of weight >= 2^16 in four i32 class sums (x1 w0 + x0 w1 and x0 w0 dropped: < 2^-24 of
    the sum), combined by three fp32 fused multiply-adds on the exactly converted sums, scaled by a power of two, bias added by the last fma"""
    xq = quant_rows(x, ex)
    x0, x1, x2 = limbs(xq)
    w0, w1, w2, w3 = limbs4(Wq)
    assert np.all(x2 * 65536 + x1 * 256 + x0 == xq) and np.all(np.abs(x2) <= 65)
    assert np.all(((w3 * 256 + w2) * 256 + w1) * 256 + w0 == Wq) and np.all(np.abs(w3) <= 33)
    c0 = x2 @ w3.T
    c1 = x2 @ w2.T + x1 @ w3.T
    c2 = x2 @ w1.T + x1 @ w2.T + x0 @ w3.T
    c3 = x2 @ w0.T + x1 @ w1.T + x0 @ w2.T
    assert max(np.abs(c).max() for c in (c0, c1, c2, c3)) < 2 ** 24
    fma = lambda a, b, c: (a.astype(np.float64) * b + c.astype(np.float64)).astype(f32)   # noqa: E731  (exact in double: |.| < 2^53)
    u = fma(c0.astype(f32), 256.0, c1.astype(f32))
    u = fma(u, 256.0, c2.astype(f32))
    u = fma(u, 256.0, c3.astype(f32))
    if not drop00:
        u = (u.astype(np.float64) + (x1 @ w0.T + x0 @ w1.T) / 256.0 + (x0 @ w0.T) / 65536.0).astype(f32)
    t = u * pow2(ex - 17)[:, None]                    # exact: a power of two
    return fma(t, pow2(ew - 18)[None, :].astype(np.float64), np.broadcast_to(bias.astype(f32), t.shape))


def layernorm_stats(x):
    """sums in lane blocks of 16 features (register order), blocks added in (w, h) order; N = 128"""
    n = x.shape[1]
    if n == 128:
        blk = x[:, ORD]                                # [rows, 8, 16]
        s = np.zeros((x.shape[0], 8), f32)
        for j in range(16):
            s = s + blk[:, :, j]
        tot = np.zeros(x.shape[0], f32)
        for b in range(8):
            tot = tot + s[:, b]
        mean = tot * f32(1.0 / 128)
        d = (x - mean[:, None]).astype(f32)
        db = d[:, ORD]
        q = np.zeros((x.shape[0], 8), f32)
        for j in range(16):
            q = (db[:, :, j].astype(np.float64) ** 2 + q).astype(f32)
        qt = np.zeros(x.shape[0], f32)
        for b in range(8):
            qt = qt + q[:, b]
        var = qt * f32(1.0 / 128)
    else:                                              # LN0: 22 features, one sequential chain
        tot = np.zeros(x.shape[0], f32)
        for j in range(n):
            tot = tot + x[:, j]
        mean = tot * f32(1.0 / n)
        d = (x - mean[:, None]).astype(f32)
        qt = np.zeros(x.shape[0], f32)
        for j in range(n):
            qt = (d[:, j].astype(np.float64) ** 2 + qt).astype(f32)
        var = qt * f32(1.0 / n)
    rstd = (f32(1.0) / np.sqrt(var + f32(1e-5))).astype(f32)
    return d, rstd, np.max(np.abs(d), axis=1)


def layernorm(x, g, b):
    d, rstd, m = layernorm_stats(x.astype(f32))
    y = ((d * rstd[:, None]).astype(f32).astype(np.float64) * g.astype(np.float64) + b.astype(np.float64)).astype(f32)
    bound = ((m * rstd).astype(f32).astype(np.float64) * float(np.max(np.abs(g))) + float(np.max(np.abs(b)))).astype(f32) * f32(1.000001)
    ex = exponent_of(bound)
    assert np.all(np.abs(y) < np.ldexp(1.0, ex)[:, None])
    return y, ex


def act_exp(x):
    x = np.clip(x, f32(-87), f32(88)).astype(f32)
    k = np.rint(x * f32(1.44269504)).astype(f32)
    fma = lambda a, b, c: (np.float64(a) * np.float64(b) + np.float64(c)).astype(f32)   # noqa: E731
    r = fma(k, f32(-0.693145752), x)
    r = fma(k, f32(-1.42860677e-6), r)
    p = fma(r, f32(1.38888889e-3), f32(8.33333333e-3))
    for c in (4.16666667e-2, 1.66666667e-1, 0.5, 1.0, 1.0):
        p = fma(r, p, f32(c))
    return (p.view(np.uint32) + (k.astype(np.int32) << 23).astype(np.uint32)).view(f32)


sig = lambda x: (f32(1) / (f32(1) + act_exp(-x))).astype(f32)           # noqa: E731
tanh = lambda x: (f32(1) - f32(2) / (act_exp(f32(2) * x) + f32(1))).astype(f32)   # noqa: E731


class ActorI8:
    def __init__(self, sd, drop00=True):
        self.sd, self.drop00 = sd, drop00
        self.q = {}
        for name, key in (('l1', 'base.mlp.fc.0'), ('l2', 'base.mlp.fc.3'), ('a1', 'act.mlp.fc.0'), ('a2', 'act.mlp.fc.3')):
            self.q[name] = (*quant_weights(sd[key + '.weight']), sd[key + '.bias'])
        self.q['gi'] = (*quant_weights(sd['rnn.gru.weight_ih_l0']), sd['rnn.gru.bias_ih_l0'])
        self.q['gh'] = (*quant_weights(sd['rnn.gru.weight_hh_l0']), sd['rnn.gru.bias_hh_l0'])

    def dense(self, name, x, ex):
        wq, ew, b = self.q[name]
        if wq.shape[1] < x.shape[1]:
            raise ValueError
        return dense_i8(x, ex, wq, ew, b, self.drop00)

    def forward(self, obs, h, mask):
        sd = self.sd
        relu = lambda v: np.maximum(v, f32(0))   # noqa: E731
        x, ex = layernorm(obs.astype(f32), sd['base.feature_norm.weight'], sd['base.feature_norm.bias'])
        x, ex = layernorm(relu(self.dense('l1', x, ex)), sd['base.mlp.fc.2.weight'], sd['base.mlp.fc.2.bias'])
        x, ex = layernorm(relu(self.dense('l2', x, ex)), sd['base.mlp.fc.5.weight'], sd['base.mlp.fc.5.bias'])
        hm = (h.astype(f32) * mask.reshape(-1, 1).astype(f32)).astype(f32)
        eh = exponent_of(np.max(np.abs(hm), axis=1))
        gi, gh = self.dense('gi', x, ex), self.dense('gh', hm, eh)
        r, z = sig(gi[:, :128] + gh[:, :128]), sig(gi[:, 128:256] + gh[:, 128:256])
        n = tanh(gi[:, 256:] + r * gh[:, 256:])
        hn = ((hm - n) * z + n).astype(f32)
        x, ex = layernorm(hn, sd['rnn.norm.weight'], sd['rnn.norm.bias'])
        x, ex = layernorm(relu(self.dense('a1', x, ex)), sd['act.mlp.fc.2.weight'], sd['act.mlp.fc.2.bias'])
        x, ex = layernorm(relu(self.dense('a2', x, ex)), sd['act.mlp.fc.5.weight'], sd['act.mlp.fc.5.bias'])
        W, b = sd['act.action_out.mu_net.fc.0.weight'].astype(f32), sd['act.action_out.mu_net.fc.0.bias'].astype(f32)
        out = np.zeros((x.shape[0], 4), f32)
        for o in range(4):                             # head: per lane block a sequential fmaf chain, blocks added to the bias in order
            tot = np.full(x.shape[0], b[o], f32)
            for blk in range(8):
                p = np.zeros(x.shape[0], f32)
                for j in ORD[blk]:
                    p = (np.float64(W[o, j]) * x[:, j].astype(np.float64) + p).astype(f32)
                tot = tot + p
            out[:, o] = tot
        return tanh(out), hn


def main():
    d = np.load(os.path.join(ROOT, 'tests', 'golden', 'actor_kat.npz'))
    sd = {k[4:]: d[k] for k in d.files if k.startswith('sd::')}
    for drop in (False, True):
        a8 = ActorI8(sd, drop)
        h = np.zeros((96, 128), f32)
        ea = eh = 0.0
        for t in range(d['obs'].shape[0]):
            a, h = a8.forward(d['obs'][t], h, d['masks'][t])
            ea = max(ea, float(np.max(np.abs(a - d['actions'][t]))))
            eh = max(eh, float(np.max(np.abs(h - d['rnn'][t][:, 0]))))
        print(f'actor_kat, low classes {"dropped" if drop else "kept"}: max |action - reference| {ea:.3e}, max |recurrent state - reference| {eh:.3e}   (bound 2e-5 / 5e-5)')
    # closed loop: the reference's PlanningEnv.step x 3 (150 inner steps), the FDM half through the oracle
    from oracle.f16_oracle import Oracle
    from tests.planning_closed import STATE_FLOORS, actor_state_dict, planning_targets, relerr
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'planning_closed_kat.npz'))
    a8 = ActorI8(actor_state_dict(g), True)
    n = g['hi_actions'].shape[1]
    o, st, h, ones = Oracle('tracking'), Oracle.new_state(n), np.zeros((n, 128), f32), np.ones(n, f32)
    for k in range(3):
        o.reset(st, rand_u=g[f'rand_u_{k}'], want_obs=False)
        tgt3 = planning_targets(st['s'], g['hi_actions'][k])
        ea = 0.0
        for i in range(50):
            a, h = a8.forward(o.lowlevel_obs(st, tgt3), h, ones)
            ea = max(ea, float(np.max(np.abs(a - g[f'll_act_{k}'][i]))))
            obs, rew, dn, bd, tm = o.step_inner(st, a)
        fl = np.stack([dn, bd, tm]).astype(np.uint8)
        print(f'closed loop, macro-step {k + 1}: masks equal {np.array_equal(fl, g[f"flags_{k}"])}, states {relerr(st["s"], g[f"s_{k}"], STATE_FLOORS):.2e} (1e-4), '
              f'recurrent state {np.max(np.abs(h - g[f"rnn_{k}"])):.2e} (5e-5), low-level actions {ea:.2e} (2e-5)')


if __name__ == '__main__':
    main()
