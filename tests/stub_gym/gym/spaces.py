"""Minimal gym.spaces (TEST STAND-IN, see gym/__init__.py)."""
import numpy as np


class Space(object):
    pass


class Discrete(Space):
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.int64

    def sample(self):
        return int(np.random.randint(self.n))


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype


class Tuple(Space):
    def __init__(self, spaces):
        self.spaces = tuple(spaces)

    def __len__(self):
        return len(self.spaces)

    def __getitem__(self, i):
        return self.spaces[i]

    def sample(self):
        return tuple(s.sample() for s in self.spaces)
