"""Shape/dtype containers only (the reference reads .shape, .low, .spaces and .sample())."""
import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.full(self.shape, low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype)


class Discrete:
    def __init__(self, n):
        self.n = int(n)
        self._rng = np.random.RandomState()

    def sample(self):
        return int(self._rng.randint(self.n))


class Dict:
    def __init__(self, spaces):
        self.spaces = dict(spaces)
