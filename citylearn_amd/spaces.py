"""`Box` space: gymnasium's when it is installed, otherwise a minimal local equivalent (the reference imports
`gymnasium.spaces.Box` for `action_space` / `observation_space`, citylearn.py:10; gymnasium is not part of this image)."""
from __future__ import annotations

import numpy as np

try:                                       # pragma: no cover - depends on the environment
    from gymnasium.spaces import Box       # type: ignore
except Exception:                          # noqa: BLE001
    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
            self.dtype = np.dtype(dtype)
            low = np.asarray(low, dtype=self.dtype)
            high = np.asarray(high, dtype=self.dtype)
            if shape is not None:
                low = np.broadcast_to(low, shape).copy()
                high = np.broadcast_to(high, shape).copy()
            self.low, self.high, self.shape = low, high, low.shape
            self._rng = np.random.RandomState(seed)

        def sample(self) -> np.ndarray:
            lo = np.where(np.isfinite(self.low), self.low, -1e6)
            hi = np.where(np.isfinite(self.high), self.high, 1e6)
            return self._rng.uniform(lo, hi).astype(self.dtype)

        def contains(self, x) -> bool:
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self) -> str:
            return f'Box({self.low}, {self.high}, {self.shape}, {self.dtype})'
