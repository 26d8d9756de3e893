"""Space descriptors.  gym's own classes are used when gym is importable; otherwise these
minimal stand-ins carry the same fields the reference exposes (ray.py:84-110,133)."""
import numpy as np

try:  # pragma: no cover - gym is not installed in the build image
    from gym.spaces import Box, Dict, Discrete  # noqa: F401
    HAVE_GYM = True
except Exception:  # noqa: BLE001
    HAVE_GYM = False

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.shape = tuple(shape)
            self.dtype = np.dtype(dtype)
            self.low = np.full(self.shape, low, dtype=self.dtype)
            self.high = np.full(self.shape, high, dtype=self.dtype)

        def __repr__(self):
            return 'Box(%s, %s, %s, %s)' % (self.low.min(), self.high.max(), self.shape, self.dtype)

    class Discrete:
        def __init__(self, n):
            self.n = int(n)
            self.shape = ()
            self.dtype = np.dtype(np.int64)
            self._rng = np.random.RandomState()

        def seed(self, seed=None):
            self._rng = np.random.RandomState(seed)

        def sample(self):
            return int(self._rng.randint(self.n))

        def __repr__(self):
            return 'Discrete(%d)' % self.n

    class Dict:
        def __init__(self, spaces):
            self.spaces = dict(spaces)

        def __getitem__(self, k):
            return self.spaces[k]

        def __repr__(self):
            return 'Dict(%s)' % ', '.join('%s:%r' % kv for kv in self.spaces.items())


class MultiDiscrete:
    """action_space of the batch (gym.vector batches Discrete(6) into MultiDiscrete([6]*N))."""

    def __init__(self, nvec):
        self.nvec = np.asarray(nvec, dtype=np.int64)
        self.shape = self.nvec.shape
        self.dtype = np.dtype(np.int64)
        self._rng = np.random.RandomState()

    def seed(self, seed=None):
        self._rng = np.random.RandomState(seed)

    def sample(self):
        return (self._rng.random_sample(self.nvec.shape) * self.nvec).astype(np.int64)


def batch_space(space, n):
    """gym.vector.utils.batch_space for the spaces used here: a leading axis of n on every Box (low/high tiled), Dict
    batched key by key, Discrete -> MultiDiscrete.  Works on gym's own classes and on the stand-ins above."""
    if hasattr(space, 'spaces'):
        return Dict({k: batch_space(v, n) for k, v in space.spaces.items()})
    if hasattr(space, 'n'):
        return MultiDiscrete([space.n] * n)
    low = np.broadcast_to(np.asarray(space.low), space.shape)
    high = np.broadcast_to(np.asarray(space.high), space.shape)
    b = Box(low=low.min() if low.size else 0, high=high.max() if high.size else 0, shape=(n,) + tuple(space.shape), dtype=space.dtype)
    return b
