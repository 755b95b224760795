"""Test-only stub of the gymnasium names rl_games imports (see ../README.md)."""
from . import spaces, vector, wrappers  # noqa: F401


class _Anything:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        return _Anything()


class Env:
    metadata = {}

    def __init__(self, *a, **k):
        pass


class Wrapper(Env):
    def __init__(self, env=None, *a, **k):
        self.env = env


class ObservationWrapper(Wrapper):
    pass


class RewardWrapper(Wrapper):
    pass


class ActionWrapper(Wrapper):
    pass


def register(*a, **k):
    pass


def make(*a, **k):
    raise RuntimeError('gymnasium stub: no environments available')


def __getattr__(name):
    return _Anything
