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

    blob = b'NPF16MLP' + struct.pack('<II', 1, len(NETS)) + b''.join(records) + np.concatenate(params).tobytes()
    out_dir = os.path.join(REPO, 'neuralplane_amd', 'assets')
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, 'f16_aero_mlp.bin'), 'wb') as f:
        f.write(blob)
    manifest = dict(
        format='NPF16MLP v1', n_nets=len(NETS), n_params=int(off), nbytes=len(blob),
        sha256=hashlib.sha256(blob).hexdigest(),
        source='xuecy22/NeuralPlane @ 2024-12-18: envs/models/F16/model/*.pth + model/mean_std.csv (GPL-3.0 data)',
        nets=manifest_nets)
    with open(os.path.join(out_dir, 'f16_aero_mlp.json'), 'w') as f:
        json.dump(manifest, f, indent=1)
    print('wrote', len(blob), 'bytes;', off, 'params; sha256', manifest['sha256'])


if __name__ == '__main__':
    main()
