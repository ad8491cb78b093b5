"""Observation / action spaces.

The reference builds ``gymnasium.spaces.Box`` / ``Dict`` from dm_env specs
(mujoco_sim/environments/dmc2gym.py:18-63). gymnasium is used when importable; otherwise the
minimal stand-ins below provide the attributes the reference's callers touch
(shape, dtype, low, high, sample, seed, contains, Dict mapping access).
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

try:  # pragma: no cover - depends on the environment
    from gymnasium.spaces import Box, Dict  # type: ignore

    HAVE_GYMNASIUM = True
except Exception:  # noqa: BLE001
    HAVE_GYMNASIUM = False

    class Box:  # type: ignore[no-redef]
        def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
            self.dtype = np.dtype(dtype)
            if shape is None:
                shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
            self.shape = tuple(shape)
            self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
            self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()
            self._rng = np.random.default_rng(seed)

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)
            return [seed]

        def sample(self):
            lo = np.where(np.isfinite(self.low), self.low, -1.0)
            hi = np.where(np.isfinite(self.high), self.high, 1.0)
            return self._rng.uniform(lo, hi).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    class Dict(OrderedDict):  # type: ignore[no-redef]
        def __init__(self, spaces=None, seed=None):
            super().__init__(spaces or {})

        @property
        def spaces(self):
            return self

        def seed(self, seed=None):
            for i, s in enumerate(self.values()):
                s.seed(None if seed is None else seed + i)
            return [seed]

        def sample(self):
            return OrderedDict((k, s.sample()) for k, s in self.items())

        def contains(self, x):
            return all(k in x and s.contains(x[k]) for k, s in self.items())


def batch_box(space: "Box", n: int) -> "Box":
    return Box(np.broadcast_to(space.low, (n,) + space.shape).copy(), np.broadcast_to(space.high, (n,) + space.shape).copy(), dtype=space.dtype)
