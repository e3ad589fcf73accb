"""Would a block-fixed-point controller (DESIGN.md section 14 b) stay inside the bound the fp32 controller is held to against the reference
recording (tests/golden/actor_kat.npz: the reference PPOActor's weights, four consecutive calls on 96 rows, its actions and recurrent
states; tests: <= 2e-5)?  numpy prototype of the forward pass with the nine Linear(128, .) layers in (a) float64, (b) float32,
(c) block fixed point (sign + 22 bits: the top balanced limb of a 24-bit value would not fit an int8, tools/microbench/i8_mlp_stack.hip):
row-wise exponent for the input, per-output-feature exponent for the weights, three balanced signed 8-bit limbs each, the six limb products of
weight >= 2^16, combined as two fp32 fused multiply-adds (fx6) or exactly with one rounding (fx9: all nine products).  CPU only; no GPU, no reference code.
    python tools/microbench/i8_actor_numerics.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = np.load(os.path.join(ROOT, 'tests', 'golden', 'actor_kat.npz'))
sd = {k[4:]: d[k] for k in d.files if k.startswith('sd::')}

def limbs(v):   # balanced signed digits: v = l2 * 65536 + l1 * 256 + l0
    v = v.astype(np.int64)
    l0 = ((v + 128) & 255) - 128; v = (v - l0) >> 8
    l1 = ((v + 128) & 255) - 128; v = (v - l1) >> 8
    return l0, l1, v

def quant(x, axis):
    m = np.max(np.abs(x), axis=axis, keepdims=True).astype(np.float64)
    e = (np.floor(np.log2(np.maximum(m, 1e-300))) + 1).astype(np.int64) * (m > 0)      # max|x| < 2^e
    q = np.rint(x.astype(np.float64) * np.exp2(22 - e)).astype(np.int64)      # |q| <= 2^22
    return q, e

def linear(x, W, b, mode):
    if mode == 'f64':
        return x.astype(np.float64) @ W.astype(np.float64).T + b
    if mode == 'f32':
        return (x.astype(np.float32) @ W.astype(np.float32).T + b.astype(np.float32)).astype(np.float32)
    xq, ex = quant(np.asarray(x, np.float32), 1)              # [rows,128], [rows,1]
    wq, ew = quant(W.astype(np.float32), 1)                   # [out,128], [out,1]
    if mode == 'fx9':
        S = xq @ wq.T
        y = (S.astype(np.float64) * np.exp2((ex + ew.T - 44).astype(np.float64))).astype(np.float32)   # one rounding (the int64 -> double product is exact: |S| < 2^53)
    else:                                                      # six of nine limb products, combined as the device would: two fp32 fused multiply-adds
        x0, x1, x2 = limbs(xq); w0, w1, w2 = limbs(wq)
        c0 = (x2 @ w2.T).astype(np.float64); c1 = (x2 @ w1.T + x1 @ w2.T).astype(np.float64); c2 = (x2 @ w0.T + x1 @ w1.T + x0 @ w2.T).astype(np.float64)
        u = (c0 * 256.0 + c1).astype(np.float32).astype(np.float64)          # fmaf: exact product and sum in double (|.| < 2^53), one rounding to fp32
        v = (u * 256.0 + c2).astype(np.float32)
        y = (v.astype(np.float64) * np.exp2((ex + ew.T - 44 + 16).astype(np.float64))).astype(np.float32)   # a power of two: exact
    return (y + b.astype(np.float32)).astype(np.float32)

def ln(x, w, b, f):
    x = x.astype(f)
    mu = x.mean(1, keepdims=True); v = ((x - mu) ** 2).mean(1, keepdims=True)
    return ((x - mu) / np.sqrt(v + f(1e-5)) * w.astype(f) + b.astype(f)).astype(f)

def forward(obs, h, mask, mode):
    f = np.float64 if mode == 'f64' else np.float32
    act = lambda v: np.maximum(v, 0)
    x = ln(obs, sd['base.feature_norm.weight'], sd['base.feature_norm.bias'], f)
    x = (x.astype(f) @ sd['base.mlp.fc.0.weight'].astype(f).T + sd['base.mlp.fc.0.bias'].astype(f)).astype(f)   # K = 22: stays floating point
    x = ln(act(x), sd['base.mlp.fc.2.weight'], sd['base.mlp.fc.2.bias'], f)
    x = ln(act(linear(x, sd['base.mlp.fc.3.weight'], sd['base.mlp.fc.3.bias'], mode)), sd['base.mlp.fc.5.weight'], sd['base.mlp.fc.5.bias'], f)
    h = (h * mask).astype(f)
    gi = linear(x, sd['rnn.gru.weight_ih_l0'], sd['rnn.gru.bias_ih_l0'], mode).astype(f)
    gh = linear(h, sd['rnn.gru.weight_hh_l0'], sd['rnn.gru.bias_hh_l0'], mode).astype(f)
    sig = lambda v: 1 / (1 + np.exp(-v))
    r = sig(gi[:, :128] + gh[:, :128]); z = sig(gi[:, 128:256] + gh[:, 128:256])
    n = np.tanh(gi[:, 256:] + r * gh[:, 256:])
    h = ((1 - z) * n + z * h).astype(f)
    x = ln(h, sd['rnn.norm.weight'], sd['rnn.norm.bias'], f)
    x = ln(act(linear(x, sd['act.mlp.fc.0.weight'], sd['act.mlp.fc.0.bias'], mode)), sd['act.mlp.fc.2.weight'], sd['act.mlp.fc.2.bias'], f)
    x = ln(act(linear(x, sd['act.mlp.fc.3.weight'], sd['act.mlp.fc.3.bias'], mode)), sd['act.mlp.fc.5.weight'], sd['act.mlp.fc.5.bias'], f)
    a = (x.astype(f) @ sd['act.action_out.mu_net.fc.0.weight'].astype(f).T + sd['act.action_out.mu_net.fc.0.bias'].astype(f)).astype(f)   # N = 4: floating point
    return np.tanh(a).astype(f), h   # MuNet: Linear + Tanh (distributions.py:79-88); deterministic = the mean

for mode in ('f64', 'f32', 'fx9', 'fx6'):
    h = np.zeros((96, 128))
    ea = eh = 0.0
    for t in range(d['obs'].shape[0]):
        a, h = forward(d['obs'][t], h, d['masks'][t], mode)
        ea = max(ea, float(np.max(np.abs(a - d['actions'][t]))))
        eh = max(eh, float(np.max(np.abs(h - d['rnn'][t][:, 0]))))
    print(f'{mode}: max |action - reference| {ea:.3e}, max |recurrent state - reference| {eh:.3e}   (the shipped fp32 controller is held to 2e-5)')
