"""F16Model — the reference's model object (envs/models/F16_model.py) backed by SoA device buffers.

The numbers live in `F16Batch` (neuralplane_amd/core.py); `s`/`u` are `[n,12]`/`[n,5]` transposed
VIEWS of the SoA buffers, so `model.s[:, 2]` is a contiguous row and in-place writes by callers
(`model.s[mask] = ...`, planning_env.py:166) land in the buffers the kernels read.  Getters that
need the dynamics (acceleration, load factors, EAS, ...) come from ONE np_f16_derived launch per
state version.  The FDM step itself is not here: it is fused into BaseEnv.step's single kernel.
"""
import torch

from .F16.F16_dynamics import F16Dynamics
from .model_base import BaseModel


class F16Model(BaseModel):
    def __init__(self, config, n, device, random_seed, batch):
        super().__init__(config, n, device, random_seed)
        self._b = batch
        self.num_states = getattr(config, 'num_states', 12)
        self.num_controls = getattr(config, 'num_controls', 5)
        self.dt = getattr(config, 'dt', 0.02)
        self.solver = getattr(config, 'solver', 'euler')
        self.airspeed = getattr(config, 'airspeed', 0)
        self.max_altitude = getattr(config, 'max_altitude', 20000)
        self.min_altitude = getattr(config, 'min_altitude', 19000)
        self.max_vt = getattr(config, 'max_vt', 1200)
        self.min_vt = getattr(config, 'min_vt', 1000)
        self.init_state = config.init_state
        self.recent_s = None
        self.recent_u = None
        self.dynamics = F16Dynamics(batch)          # F16_model.py:20; `dynamics.hifi_F16` is the surrogate object (F16_dynamics.py:13)
        self.hifi_F16 = self.dynamics.hifi_F16

    # state / control as the reference lays them out ([n,k]); views, not copies
    @property
    def s(self):
        return self._b.s.t()

    @s.setter
    def s(self, value):
        self._b.s.copy_(torch.as_tensor(value, device=self._b.device).t())

    @property
    def u(self):
        return self._b.u.t()

    @u.setter
    def u(self, value):
        self._b.u.copy_(torch.as_tensor(value, device=self._b.device).t())

    def reset(self, env):
        """F16Model.reset(env) (reference F16_model.py:33-45): state and controls of the rows `env` has flagged (is_done | bad_done |
        exceed_time_limit) are re-initialised — one launch of the reset kernel on the env's flags; targets, step counters and the flags
        themselves stay (BaseEnv.reset() is the call that also runs task.reset and clears them, in the same single launch)."""
        if env is not None and getattr(env, '_batch', None) is not self._b:
            raise ValueError('F16Model.reset(env): env must be the env this model belongs to')
        self._b.model_reset()
        self.recent_s, self.recent_u = None, None      # (the reference copies s / u of the reset rows into recent_*: see update)

    def update(self, action):
        """F16Model.update(action) (reference F16_model.py:51-67): clamp, first-order control lag, one integrator step — ONE kernel launch
        (np_f16_io.inner_step = NP_INNER_UPDATE_ONLY), for callers that drive the model directly as envs/planning_env.py:160 and
        example/quick_start.ipynb do.  `recent_s` / `recent_u` hold the state / controls from before the call, as the reference's do
        ([n, 12] / [n, 5] copies: the one thing here that is not free — BaseEnv.step does not keep them)."""
        self.recent_s, self.recent_u = self.s.clone(), self.u.clone()
        self._b.update(torch.as_tensor(action, device=self._b.device))

    def get_extended_state(self):
        """nlplant(hstack(s,u)) — reference returns [n,17] with zero control derivatives (F16_model.py:47-49)."""
        d = self._b.derived()
        out = torch.zeros((self.n, 17), dtype=torch.float32, device=self._b.device)
        out[:, :12] = d[0:12].t()
        return out

    def get_state(self):
        return self.s

    def get_control(self):
        return self.u

    def get_position(self):
        b = self._b.s
        return b[0], b[1], b[2]

    def get_ground_speed(self):
        d = self._b.derived()
        return d[0], d[1]

    def get_climb_rate(self):
        return self._b.derived()[2]

    def get_posture(self):
        b = self._b.s
        return b[3], b[4], b[5]

    def get_euler_angular_velocity(self):
        d = self._b.derived()
        return d[3], d[4], d[5]

    def get_vt(self):
        return self._b.s[6]

    def get_TAS(self):
        return self._b.s[6] + self.airspeed * torch.ones_like(self._b.s[6])

    def get_EAS(self):
        return self._b.derived()[19]

    def get_AOA(self):
        return self._b.s[7]

    def get_AOS(self):
        return self._b.s[8]

    def get_angular_velocity(self):
        b = self._b.s
        return b[9], b[10], b[11]

    def get_thrust(self):
        return self._b.u[0]

    def get_control_surface(self):
        u = self._b.u
        return u[1], u[2], u[3], u[4]

    def get_velocity(self):
        s = self._b.s
        sina, cosa, sinb, cosb = torch.sin(s[7]), torch.cos(s[7]), torch.sin(s[8]), torch.cos(s[8])
        return s[6] * cosb * cosa, s[6] * sinb, s[6] * cosb * sina

    def get_acceleration(self):
        d = self._b.derived()
        return d[12], d[13], d[14]

    def get_accels(self):
        d = self._b.derived()
        return d[15], d[16], d[17]

    def get_G(self):
        nx, ny, nz = self.get_accels()
        return torch.sqrt(nx ** 2 + ny ** 2 + nz ** 2)

    def get_EAS2TAS(self):
        return self._b.derived()[18]

    def get_atmos(self):
        """(mach, qbar, ps) — F16_model.py:183-198."""
        d = self._b.derived()
        return d[20], d[21], d[22]
