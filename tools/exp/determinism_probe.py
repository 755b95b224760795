"""Which output of the full-size chain / weight-gradient launches differs between two identical runs?
(tests/test_mlp_chain_gpu.py::test_full_size_rows_are_computed_independently_of_their_position failed on its
determinism line after the packed-residual split went in.)  python tools/exp/determinism_probe.py [runs]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rl_games_amd import ops  # noqa: E402

DEV = torch.device('cuda:0')


def net(in_dim, units, out_dim, seed):
    g = torch.Generator().manual_seed(seed)
    shapes, last = [], in_dim
    for u in list(units) + [out_dim]:
        shapes.append((u, last))
        last = u
    flat = torch.empty(sum(u * i + u for u, i in shapes), device=DEV)
    layers, off = [], 0
    for u, i in shapes:
        w = torch.randn(u, i, generator=g) / i ** 0.5
        b = 0.1 * torch.randn(u, generator=g)
        wv, bv = flat[off:off + u * i].view(u, i), flat[off + u * i:off + u * i + u]
        wv.copy_(w)
        bv.copy_(b)
        off += u * i + u
        layers.append([wv, bv, 'elu'])
    layers[-1][2] = 'None'
    return [tuple(l) for l in layers], g


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    rows = 65536
    layers, g = net(108, [400, 200, 100], 22, 65536)
    chain = ops.MlpChain(layers, DEV)
    x = torch.randn(rows, 108, generator=g).to(DEV)
    d_heads = torch.randn(rows, 22, generator=g).to(DEV)
    names = ['heads', 'a0', 'a1', 'a2', 'dz0', 'dz1', 'dz2', 'gW3', 'gW2', 'gW1', 'gW0']

    def run(fill):
        heads = torch.full((rows, 22), fill, device=DEV)
        acts = [torch.full((rows, u), fill, device=DEV) for u in (400, 200, 100)]
        dzs = [torch.full((rows, u), fill, device=DEV) for u in (400, 200, 100)]
        nb = chain.num_blocks(rows, 1)
        parts = [torch.empty(nb * u, dtype=torch.float64, device=DEV) for u in (400, 200, 100)]
        chain.forward(x, heads, act_out=acts)
        chain.backward(d_heads, acts, dzs, parts)
        jobs = [(d_heads, acts[2], torch.full((22, 100), fill, device=DEV)), (dzs[2], acts[1], torch.full((100, 200), fill, device=DEV)),
                (dzs[1], acts[0], torch.full((200, 400), fill, device=DEV)), (dzs[0], x, torch.full((400, 108), fill, device=DEV))]
        plan = ops.MlpDwPlan([tuple(j[2].shape) for j in jobs], rows, DEV)
        plan.launch(jobs)
        torch.cuda.synchronize()
        return [heads] + acts + dzs + [j[2] for j in jobs]
    ref = run(float('nan'))
    print('lib', os.environ.get('RLG_HIP_LIB', 'default'), 'finite:', [bool(torch.isfinite(t).all()) for t in ref])
    for k in range(runs):
        out = run(float(k))
        bad = []
        for n, a, b in zip(names, out, ref):
            if not torch.equal(a, b):
                d = (a - b).abs()
                w = (d > 0).nonzero()
                bad.append(f'{n}: {int((d > 0).sum())} elements differ, max {float(d.max()):.3e} (scale {float(b.abs().max()):.2e}), first at {w[0].tolist()} last {w[-1].tolist()}')
        print('run', k, 'identical' if not bad else bad)


if __name__ == '__main__':
    main()
