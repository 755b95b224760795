"""One process repeats ONE optimiser launch on identical inputs while other processes use the same GPU (round 5): does the
launch's result depend on what else runs on the device?
    python tools/exp/adam_stress_shared.py MODE SECONDS      MODE = plain | pack
Every repetition restores parameters / moments from device copies, launches the step and compares exp_avg_sq, exp_avg and the
parameters with the first repetition's ON THE DEVICE (no host sync per repetition); the mismatch mask is read once at the end.
Run it next to e.g. `python bench.py --steps 40 --no-cpu-baseline` (another process, same GPU) and alone."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rl_games_amd import ops  # noqa: E402

DEV = torch.device('cuda:0')
mode = sys.argv[1] if len(sys.argv) > 1 else 'pack'
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
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
grads0 = (1e-3 * torch.randn(n, generator=g)).to(DEV)
m0 = (1e-4 * torch.randn(n, generator=g)).to(DEV)
v0 = (1e-6 * torch.rand(n, generator=g) + 1e-8).to(DEV)
version = [0]
chain = ops.MlpChain(layers, DEV, weights_version=lambda: version[0])
pack = chain.adam_pack_target() if mode == 'pack' else None
if mode == 'pack':
    chain.pack_planes(2, flat)
assert mode == 'plain' or pack is not None, 'target unavailable'
grads, m_, v_ = grads0.clone(), m0.clone(), v0.clone()
lr_slots = torch.tensor([3e-4, 3e-4], dtype=torch.float64, device=DEV)
counter = torch.tensor([3], dtype=torch.int64, device=DEV)
norm = torch.zeros(ops.grad_norm_blocks(n), dtype=torch.float64, device=DEV)
kl = torch.tensor([0.001], device=DEV)
stats = torch.zeros(4, device=DEV)
ops.grad_sumsq(grads0, 0.5, norm, None)
torch.cuda.synchronize()


def one():
    flat.copy_(init)
    grads.copy_(grads0)
    m_.copy_(m0)
    v_.copy_(v0)
    lr_slots.fill_(3e-4)
    ops.adam_step(flat, grads, m_, v_, norm, 0.5, 1.0, lr_slots, counter, schedule_kind=1, kl=kl, stats_out=stats,
                  pack=pack)


one()
torch.cuda.synchronize()
ref = (flat.clone(), m_.clone(), v_.clone())
bad = [torch.zeros(n, dtype=torch.int32, device=DEV) for _ in range(3)]
reps = 0
t0 = time.time()
while time.time() - t0 < seconds:
    for _ in range(200):
        one()
        for k, (a, r) in enumerate(zip((flat, m_, v_), ref)):
            bad[k] += (a != r).to(torch.int32)
        reps += 1
    torch.cuda.synchronize()
names = ('params', 'exp_avg', 'exp_avg_sq')
total = [int(b.sum().item()) for b in bad]
print(f'pid {os.getpid()} mode {mode}: {reps} repetitions, mismatching '
      f'(element, repetition) pairs {dict(zip(names, total))}', flush=True)
for k, b in enumerate(bad):
    idx = b.nonzero().flatten()
    if idx.numel():
        print(f'   {names[k]}: {idx.numel()} distinct elements, first {idx[:12].tolist()}, counts {b[idx[:12]].tolist()}', flush=True)
