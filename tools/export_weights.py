#!/usr/bin/env python3
"""Export the F-16 aero-coefficient surrogate (43 ReLU MLPs + normalisation constants) to a data blob.

BUILD-CONTAINER ONLY.  Imports the reference (`/root/reference`, via tools/oracle_shims) and reads
the *data* it ships — `envs/models/F16/model/*.pth` state_dicts and `model/mean_std.csv`
(reference: envs/models/F16/hifi_F16_AeroData.py:40-129 loads them, :149-746 uses them) — and
re-packs the numbers into `neuralplane_amd/assets/f16_aero_mlp.bin` (+ `.json` manifest).
No reference source text is copied; the blob holds weights only (GPL-3.0 data of
xuecy22/NeuralPlane, attribution kept in the manifest).

Blob layout (little endian), version 1:
  0   char[8]  magic "NPF16MLP"
  8   u32      version (=1)
  12  u32      n_nets (=43)
  16  n_nets x 128-byte records:
        char[24] name
        u32      input_mask   bit0=alpha[deg] bit1=beta[deg] bit2=el[deg]
        u32      n_linear     number of Linear layers (3 or 4)
        u32[6]   dims         in, hidden..., out(=1), zero padded
        f64[3]   in_mean      alpha, beta, el   (exactly the CSV doubles)
        f64[3]   in_std
        f64      out_mean
        f64      out_std
        u32      param_offset (in floats, relative to the parameter section)
        u32      n_params
  then f32[] parameter section: per net, per Linear layer: W[out][in] row-major, then b[out]
  then (version 2 of the blob) the PWL section — exact piecewise-linear form of the 22 SINGLE-INPUT nets:
        char[4] "PWL1", u32 n_tables, u32 seg_cap (=64), then per table:
        u32 net_index, u32 n_segments, f32 t[64], f32 a[64], f32 x0[64], f32 c[64]
        A ReLU MLP of ONE scalar input is exactly piecewise linear.  Segment i covers t[i-1] <= x < t[i]
        (t sorted, padded with +inf); on it  y_norm = a[i] * (x - x0[i]) + c[i]  with x0[i] an anchor on the
        segment (its left breakpoint; the right one for the first segment) and c[i] the net's value there.
        Breakpoints and lines are derived in fp64 from the fp32 weights and rounded once to fp32.
Net order = order of evaluation in F16Dynamics.nlplant (F16_dynamics.py:140-195).
"""
import contextlib
import hashlib
import io
import json
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, 'oracle_shims'))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, '/root/reference/envs')

import numpy as np  # noqa: E402
import torch  # noqa: E402

# (blob name, csv row / model attr stem, inputs)
NETS = [
    ('Cx', 'abe'), ('Cz', 'abe'), ('Cm', 'abe'), ('Cy', 'ab'), ('Cn', 'abe'), ('Cl', 'abe'),
    ('Cxq', 'a'), ('Cyr', 'a'), ('Cyp', 'a'), ('Czq', 'a'), ('Clr', 'a'), ('Clp', 'a'),
    ('Cmq', 'a'), ('Cnr', 'a'), ('Cnp', 'a'),
    ('delta_Cx_lef', 'ab'), ('delta_Cz_lef', 'ab'), ('delta_Cm_lef', 'ab'),
    ('delta_Cy_lef', 'ab'), ('delta_Cn_lef', 'ab'), ('delta_Cl_lef', 'ab'),
    ('delta_Cxq_lef', 'a'), ('delta_Cyr_lef', 'a'), ('delta_Cyp_lef', 'a'), ('delta_Czq_lef', 'a'),
    ('delta_Clr_lef', 'a'), ('delta_Clp_lef', 'a'), ('delta_Cmq_lef', 'a'), ('delta_Cnr_lef', 'a'),
    ('delta_Cnp_lef', 'a'),
    ('delta_Cy_r30', 'ab'), ('delta_Cn_r30', 'ab'), ('delta_Cl_r30', 'ab'),
    ('delta_Cy_a20', 'ab'), ('delta_Cy_a20_lef', 'ab'), ('delta_Cn_a20', 'ab'),
    ('delta_Cn_a20_lef', 'ab'), ('delta_Cl_a20', 'ab'), ('delta_Cl_a20_lef', 'ab'),
    ('delta_Cnbeta', 'a'), ('delta_Clbeta', 'a'), ('delta_Cm', 'a'), ('eta_el', 'e'),
]


SEG_CAP = 64


def pwl_of_net(linears):
    """Exact piecewise-linear form of a scalar-input ReLU MLP, in fp64.  Returns (breaks, alpha, beta):
    sorted breakpoints and, for each of the len(breaks)+1 segments, y = alpha*x + beta."""
    Ws = [l.weight.detach().numpy().astype(np.float64) for l in linears]
    bs = [l.bias.detach().numpy().astype(np.float64) for l in linears]
    # every interval carries the affine map x -> activations of the current layer: h = A*x + B
    intervals = [(-np.inf, np.inf, np.array([1.0]), np.array([0.0]))]
    for li, (W, b) in enumerate(zip(Ws, bs)):
        last = li == len(Ws) - 1
        nxt = []
        for lo, hi, A, B in intervals:
            za, zb = W @ A, W @ B + b  # pre-activation z = za*x + zb on (lo, hi)
            if last:
                nxt.append((lo, hi, za, zb))
                continue
            cuts = sorted({float(-zb[j] / za[j]) for j in range(len(za)) if za[j] != 0.0 and lo < -zb[j] / za[j] < hi})
            edges = [lo] + cuts + [hi]
            for a_, b_ in zip(edges[:-1], edges[1:]):
                if np.isinf(a_) and np.isinf(b_):
                    mid = 0.0
                elif np.isinf(a_):
                    mid = b_ - 1.0 - abs(b_)
                elif np.isinf(b_):
                    mid = a_ + 1.0 + abs(a_)
                else:
                    mid = 0.5 * (a_ + b_)
                act = (za * mid + zb) > 0.0
                nxt.append((a_, b_, np.where(act, za, 0.0), np.where(act, zb, 0.0)))
        intervals = nxt
    # merge neighbours with identical lines (cuts of dead units)
    merged = []
    for lo, hi, A, B in intervals:
        if merged and merged[-1][2] == float(A[0]) and merged[-1][3] == float(B[0]):
            merged[-1] = (merged[-1][0], hi, merged[-1][2], merged[-1][3])
        else:
            merged.append((lo, hi, float(A[0]), float(B[0])))
    breaks = [m[1] for m in merged[:-1]]
    return np.array(breaks), np.array([m[2] for m in merged]), np.array([m[3] for m in merged])


def pwl_table(linears):
    breaks, alpha, beta = pwl_of_net(linears)
    nseg = len(alpha)
    assert 1 <= nseg <= SEG_CAP, nseg
    t = np.full(SEG_CAP, np.inf, np.float32)
    t[:nseg - 1] = breaks.astype(np.float32)
    assert np.all(np.diff(t[:nseg - 1].astype(np.float64)) > 0), 'breakpoints collapse in fp32'
    a = np.zeros(SEG_CAP, np.float32)
    x0 = np.zeros(SEG_CAP, np.float32)
    c = np.zeros(SEG_CAP, np.float32)
    for i in range(nseg):
        anchor = t[i - 1] if i > 0 else (t[0] if nseg > 1 else np.float32(0))
        a[i] = np.float32(alpha[i])
        x0[i] = anchor
        c[i] = np.float32(alpha[i] * float(anchor) + beta[i])
    # pad the unused tail with the last segment (the search can never land there: t is +inf)
    a[nseg:], x0[nseg:], c[nseg:] = a[nseg - 1], x0[nseg - 1], c[nseg - 1]
    return nseg, t, a, x0, c


def main():
    from envs.models.F16.F16_dynamics import F16Dynamics
    with contextlib.redirect_stdout(io.StringIO()):
        dyn = F16Dynamics('cpu')
    hifi = dyn.hifi_F16
    csv = hifi.data
    names = list(csv['name'])

    records = []
    params = []
    manifest_nets = []
    off = 0
    for name, inputs in NETS:
        row = names.index(name)
        model = getattr(hifi, name + '_model')
        linears = [m for m in model.layers if isinstance(m, torch.nn.Linear)]
        dims = [linears[0].in_features] + [l.out_features for l in linears]
        mask = (1 if 'a' in inputs else 0) | (2 if 'b' in inputs else 0) | (4 if 'e' in inputs else 0)
        assert dims[0] == len(inputs) and dims[-1] == 1, (name, dims)
        in_mean = [float(csv[c][row]) for c in ('alpha_mean', 'beta_mean', 'el_mean')]
        in_std = [float(csv[c][row]) for c in ('alpha_std', 'beta_std', 'el_std')]
        out_mean, out_std = float(csv['mean'][row]), float(csv['std'][row])
        flat = []
        for l in linears:
            w = l.weight.detach().numpy().astype('<f4')
            b = l.bias.detach().numpy().astype('<f4')
            assert w.dtype == np.float32 and l.weight.dtype == torch.float32
            flat.append(w.reshape(-1))
            flat.append(b.reshape(-1))
        flat = np.concatenate(flat)
        rec = struct.pack('<24sII6I3d3dddII', name.encode(), mask, len(linears),
                          *(dims + [0] * (6 - len(dims))), *in_mean, *in_std, out_mean, out_std,
                          off, flat.size)
        assert len(rec) == 128, len(rec)
        records.append(rec)
        params.append(flat)
        manifest_nets.append(dict(name=name, inputs=inputs, dims=dims, n_params=int(flat.size),
                                  in_mean=in_mean, in_std=in_std, out_mean=out_mean, out_std=out_std))
        off += flat.size

    # PWL section: exact piecewise-linear tables of the single-input nets
    pwl = []
    n_tab = 0
    worst = 0.0
    for idx, (name, inputs) in enumerate(NETS):
        if len(inputs) != 1:
            continue
        model = getattr(hifi, name + '_model')
        linears = [m for m in model.layers if isinstance(m, torch.nn.Linear)]
        nseg, t, a, x0, c = pwl_table(linears)
        # self-check against the fp64 evaluation of the net on a dense grid of normalised inputs
        xs = np.concatenate([np.linspace(-40, 40, 40001), np.random.RandomState(idx).uniform(-6, 6, 20000)]).astype(np.float32)
        with torch.no_grad():
            h = torch.from_numpy(xs.astype(np.float64)).reshape(-1, 1)
            for m in model.layers:
                h = torch.nn.functional.linear(h, m.weight.double(), m.bias.double()) if isinstance(m, torch.nn.Linear) else torch.relu(h)
            ref = h.reshape(-1).numpy()
        seg = np.searchsorted(t[:nseg - 1], xs, side='right')
        got = a[seg].astype(np.float64) * (xs.astype(np.float64) - x0[seg]) + c[seg]
        err = np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref)))
        worst = max(worst, err)
        assert err < 2e-6, (name, err)
        pwl.append(struct.pack('<II', idx, nseg) + t.tobytes() + a.tobytes() + x0.tobytes() + c.tobytes())
        manifest_nets[idx]['pwl_segments'] = int(nseg)
        n_tab += 1
    print('PWL tables:', n_tab, 'worst |table - fp64 net| / max(1,|net|) =', worst)
    pwl_section = b'PWL1' + struct.pack('<II', n_tab, SEG_CAP) + b''.join(pwl)
    blob = b'NPF16MLP' + struct.pack('<II', 2, len(NETS)) + b''.join(records) + np.concatenate(params).tobytes() + pwl_section
    out_dir = os.path.join(REPO, 'neuralplane_amd', 'assets')
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, 'f16_aero_mlp.bin'), 'wb') as f:
        f.write(blob)
    manifest = dict(
        format='NPF16MLP v2 (v1 + PWL section)', n_nets=len(NETS), n_params=int(off), nbytes=len(blob),
        sha256=hashlib.sha256(blob).hexdigest(),
        source='xuecy22/NeuralPlane @ 2024-12-18: envs/models/F16/model/*.pth + model/mean_std.csv (GPL-3.0 data)',
        nets=manifest_nets)
    with open(os.path.join(out_dir, 'f16_aero_mlp.json'), 'w') as f:
        json.dump(manifest, f, indent=1)
    print('wrote', len(blob), 'bytes;', off, 'params; sha256', manifest['sha256'])


if __name__ == '__main__':
    main()
