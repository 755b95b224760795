class SyncVectorEnv:
    def __init__(self, *a, **k):
        raise RuntimeError('gymnasium stub')


class AsyncVectorEnv(SyncVectorEnv):
    pass


def __getattr__(name):
    return SyncVectorEnv
