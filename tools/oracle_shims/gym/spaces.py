"""Build-container-only placeholder for `gym.spaces` (see ../README.md)."""
import numpy as np


class Box:
    def __init__(self, low, high, shape, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

    def sample(self):
        return np.random.uniform(-1, 1, self.shape).astype(self.dtype)


class Discrete:
    pass


class MultiDiscrete:
    pass


class MultiBinary:
    pass


class Tuple:
    pass


class Dict:
    pass
