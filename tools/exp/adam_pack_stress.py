"""rlg_adam_step_pack run again and again on identical inputs (optionally in several processes on the same GPU at once):
are parameters, moments and plane bytes the same every time?    python tools/exp/adam_pack_stress.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rl_games_amd import ops  # noqa: E402

DEV = torch.device('cuda:0')
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
g = torch.Generator().manual_seed(7)
shapes, last = [], 108
for u in [400, 200, 100, 22]:
    shapes.append((u, last))
    last = u
n = sum(u * i + u for u, i in shapes)
flat = torch.empty(n, device=DEV)
layers, off = [], 0
for u, i in shapes:
    wv, bv = flat[off:off + u * i].view(u, i), flat[off + u * i:off + u * i + u]
    wv.copy_(torch.randn(u, i, generator=g) / i ** 0.5)
    bv.copy_(0.1 * torch.randn(u, generator=g))
    off += u * i + u
    layers.append((wv, bv, 'elu'))
layers[-1] = (layers[-1][0], layers[-1][1], 'None')
init = flat.clone()
grads = (0.1 * torch.randn(n, generator=g)).to(DEV)
m0 = (0.01 * torch.randn(n, generator=g)).to(DEV)
v0 = (0.001 * torch.rand(n, generator=g)).to(DEV)
version = [0]
chain = ops.MlpChain(layers, DEV, weights_version=lambda: version[0])
target = chain.adam_pack_target()
ref = None
bad = 0
for k in range(reps):
    flat.copy_(init)
    chain.pack_planes(2, flat)
    g_, m_, v_ = grads.clone(), m0.clone(), v0.clone()
    lr_slots = torch.tensor([3e-4, 3e-4], dtype=torch.float64, device=DEV)
    counter = torch.tensor([3], dtype=torch.int64, device=DEV)
    norm = torch.zeros(ops.grad_norm_blocks(n), dtype=torch.float64, device=DEV)
    ops.grad_sumsq(g_, 1.0, norm, None)
    kl = torch.tensor([0.001], device=DEV)
    stats = torch.zeros(4, device=DEV)
    ops.adam_step(flat, g_, m_, v_, norm, 1.0, 0.5, lr_slots, counter, schedule_kind=1, kl=kl, stats_out=stats, pack=target)
    torch.cuda.synchronize()
    out = (flat.clone(), m_, v_, chain._plane_buffer().clone())
    if ref is None:
        ref = out
    else:
        d = [not torch.equal(a, b) for a, b in zip(out, ref)]
        if any(d):
            bad += 1
            if bad <= 3:
                pb = (out[3] != ref[3]).nonzero().flatten()
                print(f'rep {k}: differs in (params, m, v, planes) = {d}; plane bytes differing: {pb.numel()} first {pb[:6].tolist()}', flush=True)
print(f'pid {os.getpid()}: {bad} of {reps - 1} repetitions differ from the first', flush=True)
