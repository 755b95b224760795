"""CPU: host-side logic around the kernels that needs no device.

  * FlatArena.weights_token - the stamp ops.MlpChain's derived copies of the weights (bf16 planes, fp32 fragments)
    carry: it must change for every way the parameter VALUES can change (this package's own writers, and torch
    in-place writes from outside the package), and must not change for anything else the training loop does.
(The torch forms of the value_size > 1 path are held to the reference's own functions in tests/test_vs_reference_cpu.py.)
"""
import copy

import pytest
import torch
from torch import nn


def _net():
    torch.manual_seed(0)
    return nn.Sequential(nn.Linear(12, 16), nn.ELU(), nn.Linear(16, 8), nn.ELU(), nn.Linear(8, 4))


def test_weights_token_changes_with_every_write_to_the_parameters():
    from rl_games_amd.flat_optim import FlatArena
    net = _net()
    other = copy.deepcopy(net.state_dict())
    for v in other.values():
        v.mul_(1.5)
    arena = FlatArena(net.parameters())
    seen = {arena.weights_token()}

    def changed():
        t = arena.weights_token()
        new = t not in seen
        seen.add(t)
        return new

    # this package's writers (kernels behind raw pointers) announce themselves
    arena.weights_changed()
    assert changed()
    # torch writes from outside the package: load_state_dict, in-place ops on a parameter under no_grad, writes to the arena
    net.load_state_dict(other)
    assert changed()
    with torch.no_grad():
        net[0].weight.mul_(0.5)
    assert changed()
    with torch.no_grad():
        net[2].bias.copy_(torch.ones(8))
    assert changed()
    arena.flat_params.add_(1.0)
    assert changed()
    # parameters are still views of the arena behind all of that
    assert net[0].weight.data_ptr() == arena.flat_params.data_ptr()
    # nothing else the loop does moves it: forward / backward / gradient writes / zero_grad
    x = torch.randn(5, 12)
    arena.zero_grad()
    net(x).sum().backward()
    arena.flat_grads.mul_(0.5)
    arena.zero_grad()
    assert not changed()
    with torch.no_grad():
        net(x)
    assert not changed()
