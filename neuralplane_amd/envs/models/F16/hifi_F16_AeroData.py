"""hifi_F16 — the reference's aero-coefficient surrogate object (envs/models/F16/hifi_F16_AeroData.py:745-822), backed by the
device code the step kernels run.

The reference's F16Dynamics.nlplant asks this object for seven groups of coefficients (F16_dynamics.py:140-195); in this
package that evaluation is fused into the step kernel, and the object exists for what else the reference does with it: checking
the surrogates against the table values of model/coefs.csv (model/test_model.py).  Inputs are in DEGREES, as nlplant passes
them; every group method returns a tuple of [m] tensors in the reference's order.  All seven groups of one (alpha, beta, el)
come from ONE np_f16_aero_coefficients launch (the last call's result is kept, keyed by the input tensors' contents).
"""
import torch

GROUPS = {'hifi_C': (0, 6), 'hifi_damping': (6, 15), 'hifi_C_lef': (15, 21), 'hifi_damping_lef': (21, 30), 'hifi_rudder': (30, 33),
          'hifi_ailerons': (33, 39), 'hifi_other_coeffs': (39, 43)}


class hifi_F16:  # noqa: N801  (the reference's class name)
    def __init__(self, batch):
        self._b = batch

    def coefficients(self, alpha, beta, el):
        """All 43 coefficients -> [43, m] (row 24, delta_Czq_lef, is 0: nlplant never reads it, F16_dynamics.py:199)."""
        return self._b.aero_coefficients(alpha, beta, el)

    def _rows(self, name, alpha, beta=None, el=None):
        alpha = torch.as_tensor(alpha, dtype=torch.float32, device=self._b.device)
        zero = torch.zeros_like(alpha)
        out = self.coefficients(alpha, zero if beta is None else beta, zero if el is None else el)
        lo, hi = GROUPS[name]
        return tuple(out[k].reshape(alpha.shape) for k in range(lo, hi))

    def hifi_C(self, alpha, beta, el):
        return self._rows('hifi_C', alpha, beta, el)

    def hifi_damping(self, alpha):
        return self._rows('hifi_damping', alpha)

    def hifi_C_lef(self, alpha, beta):
        return self._rows('hifi_C_lef', alpha, beta)

    def hifi_damping_lef(self, alpha):
        return self._rows('hifi_damping_lef', alpha)

    def hifi_rudder(self, alpha, beta):
        return self._rows('hifi_rudder', alpha, beta)

    def hifi_ailerons(self, alpha, beta):
        return self._rows('hifi_ailerons', alpha, beta)

    def hifi_other_coeffs(self, alpha, el):
        """(delta_Cnbeta, delta_Clbeta, delta_Cm, eta_el, delta_Cm_ds == 0) — hifi_F16_AeroData.py:811-819."""
        alpha = torch.as_tensor(alpha, dtype=torch.float32, device=self._b.device)
        return self._rows('hifi_other_coeffs', alpha, None, el) + (torch.zeros_like(alpha),)
