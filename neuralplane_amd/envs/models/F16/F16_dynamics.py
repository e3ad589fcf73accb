"""F16Dynamics — the reference's dynamics object (envs/models/F16/F16_dynamics.py:10-228) for callers that use it on its own
(trim, linearisation, plots): `nlplant(x[m,17]) -> xdot[m,17]` and `atmos(alt, vt) -> (mach, qbar, ps)` for ARBITRARY states,
evaluated by the device code the step kernels run (one np_f16_derived launch per call).  Inside env.step the same functions
are part of the fused kernel; nothing here is on that path.
"""
import torch

from ....core import NUM_DERIVED
from .... import _lib
from .hifi_F16_AeroData import hifi_F16


class F16Dynamics:
    def __init__(self, batch):
        self._b = batch
        self.hifi_F16 = hifi_F16(batch)

    def _derived(self, s, u):
        """s[12,m], u[5,m] contiguous on the batch's device -> [NUM_DERIVED, m]."""
        b = self._b
        m = s.shape[1]
        out = torch.empty((NUM_DERIVED, m), dtype=torch.float32, device=b.device)
        _lib.check(b.lib.np_f16_derived(b._ctx, m, s.data_ptr(), u.data_ptr(), m, out.data_ptr(), m, b._stream()))
        return out

    def nlplant(self, x):
        """x[m,17] = (12 states, T, el, ail, rud, lef) -> xdot[m,17]; the five control derivatives are 0 (F16_dynamics.py:228)."""
        x = torch.as_tensor(x, dtype=torch.float32, device=self._b.device)
        if x.dim() != 2 or x.shape[1] != 17:
            raise ValueError(f'x must be [m, 17], got {tuple(x.shape)}')
        xt = x.t().contiguous()
        d = self._derived(xt[:12], xt[12:17])
        out = torch.zeros_like(x)
        out[:, :12] = d[:12].t()
        return out

    def compute_extended_state(self, x):
        return self.nlplant(x)

    def forward(self, t, x):
        return self.compute_extended_state(x)

    __call__ = forward

    def atmos(self, alt, vt):
        """(mach, qbar, ps) for altitude [ft] and airspeed [ft/s] tensors of any shape (F16_dynamics.py:22-35)."""
        alt = torch.as_tensor(alt, dtype=torch.float32, device=self._b.device)
        vt = torch.as_tensor(vt, dtype=torch.float32, device=self._b.device).expand_as(alt)
        m = alt.numel()
        s = torch.zeros((12, m), dtype=torch.float32, device=self._b.device)
        s[2], s[6] = alt.reshape(-1), vt.reshape(-1)
        d = self._derived(s, torch.zeros((5, m), dtype=torch.float32, device=self._b.device))
        return tuple(d[k].reshape(alt.shape) for k in (20, 21, 22))
