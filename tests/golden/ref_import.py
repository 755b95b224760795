"""TEST / BASELINE INFRASTRUCTURE: makes the real reference importable.

Adds /root/reference (the build container's read-only checkout; the reference is Python and does not travel to the GPU
box in any form) and the test-only stubs (gymnasium, tensorboardX) to sys.path, and plants a stub
`torch.utils.tensorboard` (the reference's SAC agent imports it; tensorboard is not installed).  Raises
ReferenceUnavailable where the checkout is absent.  `source()` tells what enable() used."""
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get('RLG_REFERENCE', '/root/reference')
_source = None


class ReferenceUnavailable(RuntimeError):
    pass


def source():
    """'checkout' | None (enable() not called yet)."""
    return _source


def enable():
    global _source
    if os.path.isdir(os.path.join(REFERENCE, 'rl_games')):
        root, _source = REFERENCE, 'checkout'
    else:
        raise ReferenceUnavailable(f'{REFERENCE} not present (the reference is only available in the build container)')
    os.environ['RLG_NO_TRITON'] = '1'
    stubs = os.path.join(HERE, '_stubs')
    for p in (stubs, root):
        if p not in sys.path:
            sys.path.append(p)
    if 'torch.utils.tensorboard' not in sys.modules:
        import tensorboardX
        mod = types.ModuleType('torch.utils.tensorboard')
        mod.SummaryWriter = tensorboardX.SummaryWriter
        sys.modules['torch.utils.tensorboard'] = mod
    return root
