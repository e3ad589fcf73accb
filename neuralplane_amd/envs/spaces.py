"""Observation/action spaces.  The RL side only needs `isinstance(space, gym.spaces.Box)` and
`.shape` (algorithms/utils/utils.py:16-21, act.py:24-27): use gym's Box when gym (or gymnasium) is
importable, otherwise a shape-only stand-in so that the env itself has no gym dependency."""
import numpy as np

try:  # pragma: no cover - depends on the host environment
    import gym as _gym
    Box = _gym.spaces.Box
    Env = _gym.Env
except Exception:  # gym absent
    try:  # pragma: no cover
        import gymnasium as _gym
        Box = _gym.spaces.Box
        Env = _gym.Env
    except Exception:
        class Box:  # minimal stand-in
            def __init__(self, low, high, shape, dtype=np.float32):
                self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

            def sample(self):
                return np.random.uniform(-1.0, 1.0, self.shape).astype(self.dtype)

            def __repr__(self):
                return f'Box({self.low}, {self.high}, {self.shape}, {self.dtype})'

        class Env:
            def __init__(self, *a, **k):
                pass
