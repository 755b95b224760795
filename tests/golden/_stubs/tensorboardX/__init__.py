"""Test-only stub (see ../README.md)."""


class SummaryWriter:
    def __init__(self, *a, **k):
        self.scalars = {}

    def add_scalar(self, tag, value, step=None, *a, **k):
        self.scalars.setdefault(tag, []).append((step, value))

    def flush(self):
        pass

    def close(self):
        pass
