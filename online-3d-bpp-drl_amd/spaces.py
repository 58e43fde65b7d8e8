"""Minimal stand-ins for the two gym spaces the reference boundary exposes (gym is not a dependency).

The reference's Policy only looks at `action_space.__class__.__name__ == "Discrete"`, `.n`
(acktr/model.py:27-29) and `observation_space.shape` (main.py:68,80,114); the class names and
attributes below are therefore what matters."""
import numpy as np


class Discrete(object):
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype(np.int64)

    def __repr__(self):
        return "Discrete(%d)" % self.n

    def __eq__(self, other):
        return getattr(other, "n", None) == self.n and getattr(other, "shape", None) == ()


class Box(object):
    def __init__(self, low, high, shape, dtype=np.float32):
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.full(self.shape, low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype)

    def __repr__(self):
        return "Box(%s, %s, %s, %s)" % (self.low.min(), self.high.max(), self.shape, self.dtype)
