"""Getter surface of an aircraft model — the API tasks, PID controllers and renders call.

Mirrors the reference's abstract BaseModel (envs/models/model_base.py:7-250): same method names,
same units, `f32[n]` tensors (or tuples of them) on the env device.
"""
import random
from abc import ABC, abstractmethod

import numpy as np
import torch


def _getter(doc):
    """An abstract no-argument getter with this docstring."""
    @abstractmethod
    def getter(self):
        raise NotImplementedError
    getter.__doc__ = doc
    return getter


class BaseModel(ABC):
    def __init__(self, config, n, device, random_seed):
        self.config = config
        self.n = n
        self.device = device
        self.random_seed = random_seed

    def seed(self, random_seed):
        """model_base.py:19-23: seeds the process-wide generators (the env's own draws are counter-based and keyed by
        BaseEnv.seed; this is the side effect a caller of the reference relies on for its policy's sampling)."""
        torch.manual_seed(random_seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(random_seed)
        np.random.seed(random_seed)
        random.seed(random_seed)

    @abstractmethod
    def reset(self, env):
        raise NotImplementedError

    @abstractmethod
    def update(self, action):
        raise NotImplementedError

    @abstractmethod
    def get_extended_state(self):
        raise NotImplementedError

    @abstractmethod
    def get_state(self):
        raise NotImplementedError

    @abstractmethod
    def get_control(self):
        raise NotImplementedError

    # the remaining getters of model_base.py:62-250 — every aircraft model answers all of them (declared abstract: what / unit)
    get_position               = _getter('(npos, epos, altitude) [ft]')
    get_ground_speed           = _getter('(npos_dot, epos_dot) [ft/s]')
    get_climb_rate             = _getter('altitude rate [ft/s]')
    get_posture                = _getter('(roll, pitch, yaw) [rad]')
    get_euler_angular_velocity = _getter('(roll_dot, pitch_dot, yaw_dot) [rad/s]')
    get_vt                     = _getter('airspeed vt [ft/s]')
    get_TAS                    = _getter('true airspeed [ft/s]')
    get_EAS                    = _getter('equivalent airspeed [ft/s]')
    get_AOA                    = _getter('angle of attack [rad]')
    get_AOS                    = _getter('sideslip angle [rad]')
    get_angular_velocity       = _getter('(P, Q, R) [rad/s]')
    get_thrust                 = _getter('thrust [lbf]')
    get_control_surface        = _getter('(el, ail, rud, lef) [deg]')
    get_velocity               = _getter('body-axis (U, V, W) [ft/s]')
    get_acceleration           = _getter('body-axis (ax, ay, az) [ft/s^2]')
    get_accels                 = _getter('load factors (nx, ny, nz) [g]')
    get_G                      = _getter('total load factor [g]')
    get_EAS2TAS                = _getter('EAS -> TAS ratio')
    get_atmos                  = _getter('(mach, qbar, ps)')
