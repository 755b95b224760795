import numpy as np


class Space:
    pass


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.shape(low)
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()


class Discrete(Space):
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype(np.int64)


class Tuple(Space, tuple):
    dtype = np.dtype(np.int64)

    def __new__(cls, spaces):
        return tuple.__new__(cls, spaces)

    def __init__(self, spaces):
        pass


class Dict(Space):
    def __init__(self, spaces):
        self.spaces = dict(spaces)


class MultiDiscrete(Space):
    def __init__(self, nvec):
        self.nvec = np.asarray(nvec)
