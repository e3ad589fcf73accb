"""Getter surface of an aircraft model — the API tasks, PID controllers and renders call.

Mirrors the reference's abstract BaseModel (envs/models/model_base.py:7-250): same method names,
same units, `f32[n]` tensors (or tuples of them) on the env device.
"""
from abc import ABC, abstractmethod


class BaseModel(ABC):
    def __init__(self, config, n, device, random_seed):
        self.config = config
        self.n = n
        self.device = device
        self.random_seed = random_seed

    @abstractmethod
    def reset(self, env):
        raise NotImplementedError

    @abstractmethod
    def update(self, action):
        raise NotImplementedError

    @abstractmethod
    def get_state(self):
        raise NotImplementedError

    @abstractmethod
    def get_control(self):
        raise NotImplementedError
