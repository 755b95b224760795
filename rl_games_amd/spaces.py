"""Minimal observation/action space descriptors.

rl_games reads spaces only through `.shape`, `.dtype`, `.low`, `.high`, `.n`, `.spaces` and the
class *name* ('Box', 'Discrete', 'Tuple', 'Dict') - see rl_games/common/experience.py:349-366,
:406-431 - so gymnasium/gym space objects work unchanged with this package.  These classes
exist because gymnasium is not a dependency here (synthetic envs, tests, bench)."""
import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.shape(low)
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()

    def __repr__(self):
        return f'Box({self.shape}, {self.dtype})'


class Discrete:
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype(np.int64)


class Tuple(tuple):
    """Tuple of Discrete spaces (multi-discrete actions)."""
    dtype = np.dtype(np.int64)

    def __new__(cls, spaces):
        return super().__new__(cls, spaces)


class Dict:
    def __init__(self, spaces):
        self.spaces = dict(spaces)


def space_kind(space):
    return type(space).__name__
