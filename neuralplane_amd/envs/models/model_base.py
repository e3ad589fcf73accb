"""Getter surface of an aircraft model — the API tasks, PID controllers and renders call.

Mirrors the reference's abstract BaseModel (envs/models/model_base.py:7-250): same method names,
same units, `f32[n]` tensors (or tuples of them) on the env device.
"""
import random
from abc import ABC, abstractmethod

import numpy as np
import torch


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

    # the remaining getters of model_base.py:62-250 — every aircraft model answers all of them
    @abstractmethod
    def get_position(self):
        """(npos, epos, altitude) [ft]"""
        raise NotImplementedError

    @abstractmethod
    def get_ground_speed(self):
        """(npos_dot, epos_dot) [ft/s]"""
        raise NotImplementedError

    @abstractmethod
    def get_climb_rate(self):
        """altitude rate [ft/s]"""
        raise NotImplementedError

    @abstractmethod
    def get_posture(self):
        """(roll, pitch, yaw) [rad]"""
        raise NotImplementedError

    @abstractmethod
    def get_euler_angular_velocity(self):
        """(roll_dot, pitch_dot, yaw_dot) [rad/s]"""
        raise NotImplementedError

    @abstractmethod
    def get_vt(self):
        """airspeed vt [ft/s]"""
        raise NotImplementedError

    @abstractmethod
    def get_TAS(self):
        """true airspeed [ft/s]"""
        raise NotImplementedError

    @abstractmethod
    def get_EAS(self):
        """equivalent airspeed [ft/s]"""
        raise NotImplementedError

    @abstractmethod
    def get_AOA(self):
        """angle of attack [rad]"""
        raise NotImplementedError

    @abstractmethod
    def get_AOS(self):
        """sideslip angle [rad]"""
        raise NotImplementedError

    @abstractmethod
    def get_angular_velocity(self):
        """(P, Q, R) [rad/s]"""
        raise NotImplementedError

    @abstractmethod
    def get_thrust(self):
        """thrust [lbf]"""
        raise NotImplementedError

    @abstractmethod
    def get_control_surface(self):
        """(el, ail, rud, lef) [deg]"""
        raise NotImplementedError

    @abstractmethod
    def get_velocity(self):
        """body-axis (U, V, W) [ft/s]"""
        raise NotImplementedError

    @abstractmethod
    def get_acceleration(self):
        """body-axis (ax, ay, az) [ft/s^2]"""
        raise NotImplementedError

    @abstractmethod
    def get_accels(self):
        """load factors (nx, ny, nz) [g]"""
        raise NotImplementedError

    @abstractmethod
    def get_G(self):
        """total load factor [g]"""
        raise NotImplementedError

    @abstractmethod
    def get_EAS2TAS(self):
        """EAS -> TAS ratio"""
        raise NotImplementedError

    @abstractmethod
    def get_atmos(self):
        """(mach, qbar, ps)"""
        raise NotImplementedError
