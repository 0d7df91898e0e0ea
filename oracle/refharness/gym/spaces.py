"""TEST INFRASTRUCTURE ONLY -- see gym/__init__.py."""
import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.full(self.shape, low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype)
