"""OracleBatch — a CPU stand-in for neuralplane_amd.core.F16Batch backed by the parity oracle (oracle/f16_oracle.c).

TEST INFRASTRUCTURE ONLY.  The product has no CPU path (core.F16Batch raises without a gfx950 device); this class exists so that,
in the build container (no GPU), the reference's REAL consumers — runner/F16sim_runner.py, algorithms/pid/controller.py,
renders/render_control.py — can be run against the Python mirror `neuralplane_amd.envs` and every attribute they touch is
exercised (tests/test_reference_consumers_cpu.py patches `neuralplane_amd.envs.env_base.F16Batch` with it).  Same public
members as F16Batch, torch CPU tensors that alias the oracle's numpy state (SoA views [12, n] of the AoS arrays, so
`model.s` = `.t()` is the reference's [n, 12]); the arithmetic is the oracle's, which the GPU tests hold the HIP kernels
bit-exact to.
"""
import numpy as np
import torch

from oracle.f16_oracle import Oracle

TERM_NAMES = ('overload', 'low_altitude', 'high_speed', 'low_speed', 'extreme_state', 'unreach', 'reached')


class OracleBatch:
    TERM_NAMES = TERM_NAMES

    def __init__(self, n, config, task, device, seed=0, solver=None, row0=0, blob_path=None, aero_1d_tables=None):
        self.device = torch.device('cpu')       # whatever the caller asked for ('cuda:0' in the reference's scripts)
        self.n = int(n)
        self.task = task
        kw = {} if blob_path is None else {'blob_path': blob_path}
        self.o = Oracle(task, solver=solver or getattr(config, 'solver', None), **kw)
        self.noise_scale = float(self.o.cfg.noise_scale)
        self.aero_1d_tables = False
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.row0 = int(row0)
        self.call_idx = 0
        self.st = Oracle.new_state(self.n)
        self.s = torch.from_numpy(self.st['s']).t()          # [12, n] view of the oracle's [n, 12]
        self.u = torch.from_numpy(self.st['u']).t()          # [5, n]
        self.tgt = torch.from_numpy(self.st['tgt']).t()      # [3, n]
        self.step_count = torch.from_numpy(self.st['step_count'])
        self.flags = torch.ones((3, self.n), dtype=torch.uint8)
        self.term_counters = np.zeros(7, np.int64)
        self.term_reasons = None
        self.reward_task = None
        self._version = 0
        self._io_epoch = 0

    # -- flags: the oracle keeps three arrays, F16Batch one [3, n] tensor that is replaced every step ---------------------------
    def _push_flags(self):
        f = self.flags.numpy()
        self.st['done'][:], self.st['bad'][:], self.st['timeout'][:] = f[0], f[1], f[2]

    def _pull_flags(self):
        self.flags = torch.from_numpy(np.stack((self.st['done'], self.st['bad'], self.st['timeout'])).copy())

    def reset(self, rand_u=None, noise=None, want_obs=True):
        self._push_flags()
        obs = self.o.reset(self.st, rand_u=None if rand_u is None else np.asarray(rand_u), noise=None if noise is None else np.asarray(noise),
                           seed=self.seed, call_idx=self.call_idx, row0=self.row0, want_obs=want_obs)
        self._pull_flags()
        if self.term_reasons is not None:
            self.term_reasons.zero_()
        self.call_idx += 1
        self._version += 1
        return None if obs is None else torch.from_numpy(obs)

    def observe(self, noise=None):
        keep = {k: self.st[k].copy() for k in ('done', 'bad', 'timeout')}
        for k in keep:
            self.st[k][:] = 0
        obs = self.o.reset(self.st, noise=None if noise is None else np.asarray(noise), seed=self.seed, call_idx=self.call_idx, row0=self.row0)
        for k, v in keep.items():
            self.st[k][:] = v
        self.call_idx += 1
        return torch.from_numpy(obs)

    def step(self, action, rand_u=None, noise=None, inner=False, ll_tgt=None, ll_obs=None, want_obs=True):
        a = torch.as_tensor(action, dtype=torch.float32).detach().cpu().numpy()
        if a.ndim != 2 or a.shape[0] != self.n or a.shape[1] < 4:
            raise ValueError(f'action must be [n={self.n}, >=4], got {tuple(a.shape)}')
        self._push_flags()
        fn = self.o.step_inner if inner else self.o.step
        kw = {} if inner else {'rand_u': None if rand_u is None else np.asarray(rand_u)}
        obs, rew, dn, bd, tm = fn(self.st, a, noise=None if noise is None else np.asarray(noise), seed=self.seed, call_idx=self.call_idx,
                                  row0=self.row0, **kw)
        self._pull_flags()
        reasons = self.o.termination_reasons(self.st)
        for k in range(7):
            self.term_counters[k] += int(((reasons >> k) & 1).sum())
        if self.term_reasons is not None:
            self.term_reasons.copy_(torch.from_numpy(reasons))
        self.call_idx += 1
        self._version += 1
        if ll_obs is not None:
            ll_obs.copy_(self.lowlevel_obs(ll_tgt))
        return (torch.from_numpy(obs) if want_obs else None), torch.from_numpy(rew), self.flags

    def lowlevel_obs(self, tgt3):
        t = torch.as_tensor(tgt3, dtype=torch.float32).numpy()
        return torch.from_numpy(self.o.lowlevel_obs(self.st, np.ascontiguousarray(t.T)))

    def derived(self):
        """[23, n] as np_f16_derived: xdot[0..11], body accelerations, nx ny nz, EAS2TAS, EAS, mach, qbar, ps."""
        s, u = self.st['s'], self.st['u']
        x17 = np.hstack((s, u)).astype(np.float32)
        out = np.empty((23, self.n), np.float32)
        out[0:12] = self.o.nlplant(x17).T
        out[12:15] = self.o.get_acceleration(s, u).T
        out[15:18] = self.o.get_accels(s, u).T
        e2t = self.o.get_eas2tas(s)
        out[18] = e2t
        out[19] = (s[:, 6] + np.float32(self.o.cfg.airspeed) * np.float32(1.0)) / e2t
        out[20:23] = self.o.get_atmos(s).T
        return torch.from_numpy(out)

    def aero_coefficients(self, alpha_deg, beta_deg, el):
        a, b, e = (torch.as_tensor(v, dtype=torch.float32).reshape(-1).numpy() for v in (alpha_deg, beta_deg, el))
        return torch.from_numpy(self.o.aero(a, b, e).T.copy())

    # -- optional per-aircraft outputs / statistics ---------------------------------------------------------------------------
    def track_termination_reasons(self, enable=True):
        self.term_reasons = torch.zeros(self.n, dtype=torch.uint8) if enable else None
        return self.term_reasons

    def track_reward_terms(self, enable=True):
        self.reward_task = torch.zeros(self.n, dtype=torch.float32) if enable else None
        return self.reward_task

    def termination_counts(self, reset=False):
        out = {k: int(v) for k, v in zip(TERM_NAMES, self.term_counters)}
        if reset:
            self.term_counters[:] = 0
        return out

    def state_dict(self):
        return {'s': self.s.clone(), 'u': self.u.clone(), 'tgt': self.tgt.clone(), 'step_count': self.step_count.clone(),
                'flags': self.flags.clone(), 'call_idx': int(self.call_idx), 'seed': int(self.seed), 'row0': int(self.row0),
                'task': self.task, 'n': self.n}

    def load_state_dict(self, sd):
        for k in ('s', 'u', 'tgt', 'step_count'):
            getattr(self, k).copy_(sd[k])
        self.flags = sd['flags'].clone()
        self.call_idx, self.seed, self.row0 = int(sd['call_idx']), int(sd['seed']), int(sd['row0'])
        self._version += 1

    def set_timing(self, enable):
        pass

    def get_timing_samples(self):
        return []
