"""TEST INFRASTRUCTURE: makes the real reference importable in the build container.

Adds /root/reference and the test-only stubs (gymnasium, tensorboardX) to sys.path, and plants a
stub `torch.utils.tensorboard` (the reference's SAC agent imports it; tensorboard is not
installed).  Raises ReferenceUnavailable when /root/reference is absent (GPU box)."""
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get('RLG_REFERENCE', '/root/reference')


class ReferenceUnavailable(RuntimeError):
    pass


def enable():
    if not os.path.isdir(os.path.join(REFERENCE, 'rl_games')):
        raise ReferenceUnavailable(f'{REFERENCE} not present')
    os.environ['RLG_NO_TRITON'] = '1'
    stubs = os.path.join(HERE, '_stubs')
    for p in (stubs, REFERENCE):
        if p not in sys.path:
            sys.path.append(p)
    if 'torch.utils.tensorboard' not in sys.modules:
        import tensorboardX
        mod = types.ModuleType('torch.utils.tensorboard')
        mod.SummaryWriter = tensorboardX.SummaryWriter
        sys.modules['torch.utils.tensorboard'] = mod
    return REFERENCE
