"""A from-scratch eager-PyTorch formulation of the F-16 Heading env.step — TEST / BENCH
INFRASTRUCTURE ONLY, like everything under oracle/: it is the "same-box eager PyTorch on the host cores" baseline of bench.py
(`cpu_baseline.torch_eager`) and is checked against the C oracle on the reference-generated fixtures; nothing under
neuralplane_amd/ imports it, and there is no code path from the product to it.

It is NOT the reference's files (those cannot travel to the GPU box): it is the arithmetic of Appendix A written the way a
tensor-op implementation would naturally be written — the 43 aero MLPs evaluated class by class as batched matmuls over [n]
aircraft (the reference calls 43 separate nn.Module instances), everything else as elementwise torch ops in the reference's
operator order (envs/models/F16/F16_dynamics.py:37-228, F16_model.py:51-67, tasks/heading_task.py, termination_conditions/*.py,
reward_functions/*.py, env_base.py:83-109).  Library sin / cos / pow / sqrt and ATen GEMM accumulation order, as the reference:
agreement with the oracle is therefore to ~1e-5 relative, masks equal on the fixtures (tests/test_oracle_golden.py).
"""
import json
import math
import os
import struct

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
BLOB = os.path.join(os.path.dirname(HERE), 'neuralplane_amd', 'assets', 'f16_aero_mlp.bin')


def _load_nets(path=BLOB):
    """NPF16MLP blob (tools/export_weights.py) -> list of dicts with torch weights, torch layout W[out][in]."""
    raw = open(path, 'rb').read()
    assert raw[:8] == b'NPF16MLP'
    ver, nn = struct.unpack_from('<II', raw, 8)
    recs, off = [], 16
    for _ in range(nn):
        name = raw[off:off + 24].split(b'\0')[0].decode()
        mask, nlin = struct.unpack_from('<II', raw, off + 24)
        dims = struct.unpack_from('<6I', raw, off + 32)
        in_mean = struct.unpack_from('<3d', raw, off + 56)
        in_std = struct.unpack_from('<3d', raw, off + 80)
        out_mean, out_std = struct.unpack_from('<2d', raw, off + 104)
        poff, npar = struct.unpack_from('<II', raw, off + 120)
        recs.append(dict(name=name, mask=mask, dims=dims[:nlin + 1], in_mean=in_mean, in_std=in_std, out_mean=out_mean, out_std=out_std,
                         poff=poff, npar=npar))
        off += 128
    par = np.frombuffer(raw, dtype='<f4', count=max(r['poff'] + r['npar'] for r in recs), offset=off)
    for r in recs:
        p, layers = r['poff'], []
        for a, b in zip(r['dims'][:-1], r['dims'][1:]):
            W = torch.from_numpy(par[p:p + a * b].reshape(b, a).copy())
            p += a * b
            bias = torch.from_numpy(par[p:p + b].copy())
            p += b
            layers.append((W, bias))
        r['layers'] = layers
    return recs


class _NetGroup:
    """Nets of identical shape and identical inputs, evaluated together: x[n, in] -> y[n, count] (one bmm per layer)."""

    def __init__(self, recs, idx):
        r0 = recs[idx[0]]
        self.idx = idx
        self.cols = [k for k in range(3) if r0['mask'] & (1 << k)]
        f32 = torch.float32
        self.mean = torch.tensor([[recs[i]['in_mean'][k] for k in self.cols] for i in idx], dtype=f32)     # [c, in]
        self.std = torch.tensor([[recs[i]['in_std'][k] for k in self.cols] for i in idx], dtype=f32)
        self.layers = []
        for li in range(len(r0['layers'])):
            W = torch.stack([recs[i]['layers'][li][0].t() for i in idx])        # [c, in, out]
            b = torch.stack([recs[i]['layers'][li][1] for i in idx])[:, None]   # [c, 1, out]
            self.layers.append((W.contiguous(), b.contiguous()))
        self.out_std = torch.tensor([recs[i]['out_std'] for i in idx], dtype=f32)
        self.out_mean = torch.tensor([recs[i]['out_mean'] for i in idx], dtype=f32)

    def __call__(self, abe):
        x = abe[:, self.cols]                                       # [n, in]
        h = (x[None] - self.mean[:, None]) / self.std[:, None]      # [c, n, in]   normalize: (X - mean) / std
        for k, (W, b) in enumerate(self.layers):
            h = torch.baddbmm(b, h, W)
            if k + 1 < len(self.layers):
                h = torch.relu(h)
        return h[:, :, 0].t() * self.out_std + self.out_mean       # [n, c]   unnormalize


class TorchEagerHeading:
    def __init__(self, n, cfg=None, seed=0, noise=True, threads=None):
        if threads:
            torch.set_num_threads(threads)
        self.n = n
        c = dict(dt=0.02, airspeed=0.0, noise_scale=0.01, altitude_limit=2500.0, acceleration_limit=300.0, max_velocity=3.0, min_velocity=0.01,
                 min_alpha=-20.0, max_alpha=45.0, min_beta=-30.0, max_beta=30.0, max_check_interval=2500, min_check_interval=300,
                 init_T=2000.0, max_altitude=20000.0, min_altitude=19000.0, max_vt=1200.0, min_vt=1000.0)
        c.update(cfg or {})
        self.c = c
        if not noise:
            self.c['noise_scale'] = 0.0
        recs = _load_nets()
        groups = {}
        for i, r in enumerate(recs):
            key = (r['mask'], tuple(r['dims']), tuple(r['in_mean']), tuple(r['in_std']))
            groups.setdefault(key, []).append(i)
        self.groups = [_NetGroup(recs, idx) for idx in groups.values()]
        self.names = [r['name'] for r in recs]
        self.gen = torch.Generator().manual_seed(seed)
        f32 = torch.float32
        self.s = torch.zeros((n, 12), dtype=f32)
        self.u = torch.zeros((n, 5), dtype=f32)
        self.tgt = torch.zeros((n, 3), dtype=f32)          # altitude, heading, vt
        self.step_count = torch.zeros(n, dtype=torch.int64)
        self.is_done = torch.ones(n, dtype=torch.bool)
        self.bad_done = torch.ones(n, dtype=torch.bool)
        self.exceed = torch.ones(n, dtype=torch.bool)

    # -- the 43 coefficients at (alpha_deg, beta_deg, el): [n, 43] in blob order ---------------------------------------------------
    def aero(self, alpha_deg, beta_deg, el):
        abe = torch.stack((alpha_deg, beta_deg, el), 1)
        out = torch.empty((abe.shape[0], 43), dtype=torch.float32)
        for g in self.groups:
            out[:, g.idx] = g(abe)
        return out

    def nlplant(self, s, u):
        """xdot[n, 12] — F16_dynamics.py:37-228 (Appendix A.3)."""
        g, m, B, S, cbar, xcgr, xcg = 32.17, 636.94, 30.0, 300.0, 11.32, 0.35, 0.30
        Jy, Jxz, Jz, Jx = 55814.0, 982.0, 63100.0, 9496.0
        r2d = 180.0 / math.pi
        alt, phi, theta, psi, vt0, alpha, beta, P, Q, R = (s[:, k] for k in (2, 3, 4, 5, 6, 7, 8, 9, 10, 11))
        T, el, ail, rud = u[:, 0], u[:, 1], u[:, 2], u[:, 3]
        a_deg, b_deg = alpha * r2d, beta * r2d
        sa, ca, sb, cb = torch.sin(alpha), torch.cos(alpha), torch.sin(beta), torch.cos(beta)
        st, ct, tt = torch.sin(theta), torch.cos(theta), torch.tan(theta)
        sphi, cphi, spsi, cpsi = torch.sin(phi), torch.cos(phi), torch.sin(psi), torch.cos(psi)
        vt = (vt0 <= 0.01) * 0.01 + (vt0 > 0.01) * vt0
        dail, drud = ail / 21.5, rud / 30.0
        tfac = 1 - 0.703e-5 * alt
        rho = 2.377e-3 * torch.pow(tfac, 4.14)
        qbar = 0.5 * rho * vt * vt
        U, V, W = vt * ca * cb, vt * sb, vt * sa * cb
        xd = torch.empty((s.shape[0], 12), dtype=torch.float32)
        xd[:, 0] = U * (ct * cpsi) + V * (sphi * cpsi * st - cphi * spsi) + W * (cphi * st * cpsi + sphi * spsi)
        xd[:, 1] = U * (ct * spsi) + V * (sphi * spsi * st + cphi * cpsi) + W * (cphi * st * spsi - sphi * cpsi)
        xd[:, 2] = U * st - V * (sphi * ct) - W * (cphi * ct)
        xd[:, 3] = P + tt * (Q * sphi + R * cphi)
        xd[:, 4] = Q * cphi - R * sphi
        xd[:, 5] = (Q * sphi + R * cphi) / ct
        C = self.aero(a_deg, b_deg, el)
        k = {nm: C[:, i] for i, nm in enumerate(self.names)}
        c2v, b2v = cbar / (2 * vt), B / (2 * vt)
        dXdQ = c2v * (k['Cxq'] + k['delta_Cxq_lef'])
        Cx_tot = k['Cx'] + k['delta_Cx_lef'] + dXdQ * Q
        dZdQ = c2v * (k['Czq'] + k['delta_Cz_lef'])                 # the reference's quirk (F16_dynamics.py:199)
        Cz_tot = k['Cz'] + k['delta_Cz_lef'] + dZdQ * Q
        dMdQ = c2v * (k['Cmq'] + k['delta_Cmq_lef'])
        Cm_tot = k['Cm'] * k['eta_el'] + Cz_tot * (xcgr - xcg) + k['delta_Cm_lef'] + dMdQ * Q + k['delta_Cm']
        dYdail = k['delta_Cy_a20'] + k['delta_Cy_a20_lef']
        dYdR, dYdP = b2v * (k['Cyr'] + k['delta_Cyr_lef']), b2v * (k['Cyp'] + k['delta_Cyp_lef'])
        Cy_tot = k['Cy'] + k['delta_Cy_lef'] + dYdail * dail + k['delta_Cy_r30'] * drud + dYdR * R + dYdP * P
        dNdail = k['delta_Cn_a20'] + k['delta_Cn_a20_lef']
        dNdR, dNdP = b2v * (k['Cnr'] + k['delta_Cnr_lef']), b2v * (k['Cnp'] + k['delta_Cnp_lef'])
        Cn_tot = (k['Cn'] + k['delta_Cn_lef'] - Cy_tot * (xcgr - xcg) * (cbar / B) + dNdail * dail + k['delta_Cn_r30'] * drud + dNdR * R + dNdP * P
                  + k['delta_Cnbeta'] * b_deg)
        dLdail = k['delta_Cl_a20'] + k['delta_Cl_a20_lef']
        dLdR, dLdP = b2v * (k['Clr'] + k['delta_Clr_lef']), b2v * (k['Clp'] + k['delta_Clp_lef'])
        Cl_tot = k['Cl'] + k['delta_Cl_lef'] + dLdail * dail + k['delta_Cl_r30'] * drud + dLdR * R + dLdP * P + k['delta_Clbeta'] * b_deg
        Udot = R * V - Q * W - g * st + qbar * S * Cx_tot / m + T / m
        Vdot = P * W - R * U + g * ct * sphi + qbar * S * Cy_tot / m
        Wdot = Q * U - P * V + g * ct * cphi + qbar * S * Cz_tot / m
        xd[:, 6] = (U * Udot + V * Vdot + W * Wdot) / vt
        xd[:, 7] = (U * Wdot - W * Udot) / (U * U + W * W)
        xd[:, 8] = (Vdot * vt - V * xd[:, 6]) / (vt * vt * cb)
        L, M, N = Cl_tot * qbar * S * B, Cm_tot * qbar * S * cbar, Cn_tot * qbar * S * B
        den = Jx * Jz - Jxz * Jxz
        xd[:, 9] = (Jz * L + Jxz * N - (Jz * (Jz - Jy) + Jxz * Jxz) * Q * R + Jxz * (Jx - Jy + Jz) * P * Q) / den
        xd[:, 10] = (M + (Jz - Jx) * P * R - Jxz * (P * P - R * R)) / Jy
        xd[:, 11] = (Jx * N + Jxz * L + (Jx * (Jx - Jy) + Jxz * Jxz) * P * Q - Jxz * (Jx - Jy + Jz) * Q * R) / den
        return xd

    @staticmethod
    def wrap_pi(x):
        r = torch.remainder(x, 2 * math.pi)
        r = r + 2 * math.pi * (r < 0)
        return r - 2 * math.pi * (r > math.pi)

    def _reset_rows(self, mask, rand_u=None):
        n = int(mask.sum())
        if n == 0:
            return
        c = self.c
        ru = rand_u[mask] if rand_u is not None else torch.rand((n, 2), generator=self.gen)
        s = torch.zeros((n, 12), dtype=torch.float32)
        s[:, 2] = ru[:, 0] * (c['max_altitude'] - c['min_altitude']) + c['min_altitude']
        s[:, 6] = ru[:, 1] * (c['max_vt'] - c['min_vt']) + c['min_vt']
        self.s[mask] = s
        u = torch.zeros((n, 5), dtype=torch.float32)
        u[:, 0] = c['init_T']
        self.u[mask] = u
        self.tgt[mask] = torch.stack((s[:, 2] + 1000.0, self.wrap_pi(s[:, 5] + 2 * math.pi / 3), s[:, 6] + 0.0), 1)
        self.step_count[mask] = 0

    def obs(self, noise=None):
        s, u, c = self.s, self.u, self.c
        alt, vt = s[:, 2], s[:, 6]
        e2t = torch.sqrt(1.0 / torch.pow(1 - 0.703e-5 * alt, 4.14))
        eas = (vt + c['airspeed']) / e2t
        o = torch.stack(((alt - self.tgt[:, 0]) * 0.3048 / 1000, self.wrap_pi(s[:, 5] - self.tgt[:, 1]), (vt - self.tgt[:, 2]) * 0.3048 / 340,
                         alt * 0.3048 / 5000, torch.sin(s[:, 3]), torch.cos(s[:, 3]), torch.sin(s[:, 4]), torch.cos(s[:, 4]), eas * 0.3048 / 340,
                         torch.sin(s[:, 7]), torch.cos(s[:, 7]), torch.sin(s[:, 8]), torch.cos(s[:, 8]), s[:, 9], s[:, 10], s[:, 11],
                         u[:, 0] / 0.225 / 76300 * 0.3048, u[:, 1] / 45, u[:, 2] / 45, u[:, 3] / 45, u[:, 4] / 45, e2t), 1)
        if noise is not None:       # parity hook: the reference's own draws
            o = o + noise * c['noise_scale']
        elif c['noise_scale']:
            o = o + torch.randn(o.shape, generator=self.gen) * c['noise_scale']
        return o

    def step(self, action, rand_u=None, noise=None):
        """BaseEnv.step (env_base.py:99-109): auto-reset of flagged rows, control lag, one Euler step, obs, terminations, reward."""
        c = self.c
        self._reset_rows(self.is_done | self.bad_done | self.exceed, rand_u)
        self.is_done = torch.zeros(self.n, dtype=torch.bool)
        self.bad_done = torch.zeros(self.n, dtype=torch.bool)
        self.exceed = torch.zeros(self.n, dtype=torch.bool)
        a = torch.clamp(action, -1, 1)
        u = self.u
        T = 0.9 * u[:, 0] + 0.1 * a[:, 0] * 0.225 * 76300 / 0.3048
        self.u = torch.stack((T, 0.9 * u[:, 1] + 0.1 * a[:, 1] * 45, 0.9 * u[:, 2] + 0.1 * a[:, 2] * 45, 0.9 * u[:, 3] + 0.1 * a[:, 3] * 45,
                              torch.zeros_like(T)), 1)
        self.s = self.s + np.float32(c['dt']) * self.nlplant(self.s, self.u)
        self.step_count += 1
        obs = self.obs(noise)
        s = self.s
        # terminations at the new state (Appendix A.6)
        xd = self.nlplant(s, self.u)
        vt, al, be = s[:, 6], s[:, 7], s[:, 8]
        sa, ca, sb, cb = torch.sin(al), torch.cos(al), torch.sin(be), torch.cos(be)
        ud = cb * ca * xd[:, 6] - vt * sb * ca * xd[:, 8] - vt * cb * sa * xd[:, 7]
        vd = sb * xd[:, 6] + vt * cb * xd[:, 8]
        wd = cb * sa * xd[:, 6] - vt * sb * sa * xd[:, 8] + vt * cb * ca * xd[:, 7]
        uu, vv, ww = vt * cb * ca, vt * sb, vt * cb * sa
        ax, ay, az = ud + s[:, 10] * ww - s[:, 11] * vv, vd + s[:, 11] * uu - s[:, 9] * ww, wd + s[:, 9] * vv - s[:, 10] * uu
        bad = (torch.sqrt(ax * ax + ay * ay + az * az) - c['acceleration_limit']) > 0
        bad |= (s[:, 2] - c['altitude_limit']) < 0
        mach = (vt + c['airspeed']) * 0.3048 / 340
        bad |= (mach - c['max_velocity']) >= 0
        bad |= (mach - c['min_velocity']) <= 0
        ad, bd = al * 180 / math.pi, be * 180 / math.pi
        bad |= (ad < c['min_alpha']) | (ad > c['max_alpha']) | (bd < c['min_beta']) | (bd > c['max_beta'])
        dpsi = self.wrap_pi(s[:, 5] - self.tgt[:, 1])
        off = (dpsi.abs() >= math.pi / 36) | ((s[:, 2] - self.tgt[:, 0]).abs() >= 100) | ((vt - self.tgt[:, 2]).abs() >= 20)
        m1, m2 = self.step_count >= c['max_check_interval'], self.step_count >= c['min_check_interval']
        bad |= m1 & off
        done = ~off & ~m1 & m2
        self.is_done |= done
        self.bad_done |= bad
        rew = -((s[:, 2] - self.tgt[:, 0]) * 0.3048 / 1000) ** 2 - (dpsi / math.pi) ** 2 - ((vt - self.tgt[:, 2]) * 0.3048 / 340) ** 2
        rew = rew + (-200.0 * self.bad_done + 200.0 * self.is_done)
        return obs, rew, self.is_done, self.bad_done, self.exceed


def timed_rate(n=100_000, budget_s=10.0, threads=None):
    """aircraft-steps/s of this formulation on the host cores: random actions, noise on, at least 3 steps or `budget_s` seconds."""
    import time
    if threads:
        torch.set_num_threads(threads)
    env = TorchEagerHeading(n, seed=0)
    g = torch.Generator().manual_seed(1)
    acts = [torch.rand((n, 4), generator=g) * 2 - 1 for _ in range(4)]
    with torch.no_grad():
        env.step(acts[0])
        t0, k = time.perf_counter(), 0
        while k < 3 or time.perf_counter() - t0 < budget_s:
            env.step(acts[k % 4])
            k += 1
        el = time.perf_counter() - t0
    return {'value': n * k / el, 'unit': 'aircraft-steps/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'F-16 heading, N={n} aircraft x {k} steps, oracle/torch_eager.py (eager PyTorch on the host cores, fp32, the 43 nets as batched '
                      f'matmuls per net class), {el:.1f} s'}


if __name__ == '__main__':
    print(json.dumps(timed_rate()))
