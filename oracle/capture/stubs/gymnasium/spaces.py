"""Space descriptors for the gymnasium stand-in (see package docstring)."""
import numpy as np


class Box(object):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = low, high, shape, dtype


class Discrete(object):
    def __init__(self, n):
        self.n = n
