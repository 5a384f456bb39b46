"""Stand-in for `gymnasium.spaces` (test infrastructure only, see __init__.py)."""
import numpy as np


class Space:
    pass


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        self.dtype = np.dtype(dtype)
        low = np.asarray(low, dtype=self.dtype)
        high = np.asarray(high, dtype=self.dtype)
        if shape is not None:
            low = np.broadcast_to(low, shape).copy()
            high = np.broadcast_to(high, shape).copy()
        self.low, self.high = low, high
        self.shape = low.shape
        self._rng = np.random.RandomState(seed)

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return f'Box({self.low}, {self.high}, {self.shape}, {self.dtype})'


class Discrete(Space):
    def __init__(self, n, seed=None, start=0):
        self.n, self.start = int(n), int(start)
        self.shape = ()
        self._rng = np.random.RandomState(seed)

    def sample(self):
        return self.start + int(self._rng.randint(self.n))


class MultiDiscrete(Space):
    def __init__(self, nvec, seed=None):
        self.nvec = np.asarray(nvec, dtype=np.int64)
        self.shape = self.nvec.shape
        self._rng = np.random.RandomState(seed)

    def sample(self):
        return (self._rng.random_sample(self.nvec.shape) * self.nvec).astype(np.int64)


class Dict(Space, dict):
    def __init__(self, spaces=None, **kw):
        dict.__init__(self, spaces or {}, **kw)
        self.spaces = self
