class _W:
    def __init__(self, *a, **k):
        raise RuntimeError('gymnasium stub')


AtariPreprocessing = FrameStackObservation = FlattenObservation = TimeLimit = _W


def __getattr__(name):
    return _W
