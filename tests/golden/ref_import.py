"""TEST / BASELINE INFRASTRUCTURE: makes the real reference importable.

Adds /root/reference - or, where that does not exist (the GPU box), the archive oracle/stage_reference.py staged
from it (oracle/_ref/rl_games_ref.zip: the same files byte for byte, imported through zipimport) - and the
test-only stubs (gymnasium, tensorboardX) to sys.path, and plants a stub `torch.utils.tensorboard` (the reference's
SAC agent imports it; tensorboard is not installed).  Raises ReferenceUnavailable when neither is present.
`source()` tells which one enable() used."""
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get('RLG_REFERENCE', '/root/reference')
STAGED = os.path.join(os.path.dirname(os.path.dirname(HERE)), 'oracle', '_ref', 'rl_games_ref.zip')
_source = None


class ReferenceUnavailable(RuntimeError):
    pass


def source():
    """'checkout' | 'staged archive' | None (enable() not called yet)."""
    return _source


def enable():
    global _source
    if os.path.isdir(os.path.join(REFERENCE, 'rl_games')):
        root, _source = REFERENCE, 'checkout'
    elif os.path.isfile(STAGED):
        root, _source = STAGED, 'staged archive'
    else:
        raise ReferenceUnavailable(f'neither {REFERENCE} nor {STAGED} present')
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
