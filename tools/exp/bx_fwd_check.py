"""Forward chain: split-bf16 kernel (csrc/mlp_chain_bx_fwd.hip, RLG_CHAIN_BX_FWD=1) against fp64 torch and the
exact-product kernels, then timing.    RLG_CHAIN_BX_FWD=1 python tools/exp/bx_fwd_check.py [rows]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rl_games_amd import ops

dev = 'cuda:0'
ACT = {'elu': torch.nn.functional.elu, 'relu': torch.relu, 'tanh': torch.tanh, 'None': lambda t: t}
SHAPES = [(108, [400, 200, 100], 22, 'elu'), (60, [256, 128, 64], 9, 'elu'), (13, [20, 36], 5, 'tanh'), (3, [64, 64], 2, 'relu'),
          (48, [32], 7, 'None'), (12, [100, 52], 22, 'elu'), (130, [256, 256, 128], 34, 'elu'), (33, [512, 64], 8, 'relu')]


def net(in_dim, units, out_dim, act, seed):
    g = torch.Generator().manual_seed(seed)
    shapes, last = [], in_dim
    for u in list(units) + [out_dim]:
        shapes.append((u, last))
        last = u
    flat = torch.empty(sum(u * i + u for u, i in shapes), device=dev)
    layers, off = [], 0
    for u, i in shapes:
        wv, bv = flat[off:off + u * i].view(u, i), flat[off + u * i:off + u * i + u]
        wv.copy_(torch.randn(u, i, generator=g) / i ** 0.5)
        bv.copy_(0.1 * torch.randn(u, generator=g))
        off += u * i + u
        layers.append([wv, bv, act])
    layers[-1][2] = 'None'
    return [tuple(l) for l in layers], g


bad = 0
for in_dim, units, out_dim, act in SHAPES:
    for rows in (16384, 20000 + 17):
        layers, g = net(in_dim, units, out_dim, act, rows + in_dim)
        chain = ops.MlpChain(layers, dev)
        used = chain.split_products(rows, 0)
        x = (3 * torch.randn(rows, in_dim, generator=g) + 1).to(dev)
        mean = torch.randn(in_dim, generator=g, dtype=torch.float64).to(dev)
        var = (torch.rand(in_dim, generator=g, dtype=torch.float64) * 4 + 0.1).to(dev)
        for rms in (None, (mean, var)):
            heads = torch.full((rows, out_dim), float('nan'), device=dev)
            acts = [torch.full((rows, u), float('nan'), device=dev) for u in units]
            xn = torch.full((rows, in_dim), float('nan'), device=dev) if rms is not None else None
            chain.forward(x, heads, act_out=acts, rms=rms, eps=1e-5, xn_out=xn)
            xin = x
            if rms is not None:
                xin = ops.rms_apply(x, mean, var, 1e-5, 0)
                if not torch.equal(xn, xin):
                    print('xn differs'); bad += 1
            a64, a32 = xin.double(), xin
            worst = 0.0
            for (w, b, an), got in zip(layers, acts + [heads]):
                a64 = ACT[an](torch.addmm(b.double(), a64, w.double().t()))
                a32 = ACT[an](torch.addmm(b, a32, w.t()))
                err = (got.double() - a64).abs().max().item()
                lib = (a32.double() - a64).abs().max().item()
                scale = a64.abs().max().item()
                worst = max(worst, err / max(scale, 1e-30))
                if not (err <= max(1e-6 * scale, 4 * lib)) or not torch.isfinite(got).all():
                    print(f'  BAD layer out {w.shape[0]}: err {err:.3e} lib {lib:.3e} scale {scale:.3e}')
                    bad += 1
            heads2 = torch.full((rows, out_dim), float('nan'), device=dev)
            chain.forward(x, heads2, rms=rms, eps=1e-5)
            if not torch.equal(heads2, heads):
                print('  inference form differs'); bad += 1
        print(f'in {in_dim} units {units} out {out_dim} {act} rows {rows}: split kernel {used}, worst err/scale {worst:.2e}')
print('bad', bad)


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


layers, g = net(108, [400, 200, 100], 22, 'elu', 0)
chain = ops.MlpChain(layers, dev)
for rows in (32768, 65536):
    x = (3 * torch.randn(rows, 108, generator=g) + 1).to(dev)
    mean = torch.zeros(108, dtype=torch.float64, device=dev)
    var = torch.ones(108, dtype=torch.float64, device=dev)
    heads = torch.empty(rows, 22, device=dev)
    acts = [torch.empty(rows, u, device=dev) for u in [400, 200, 100]]
    xn = torch.empty(rows, 108, device=dev)
    t_train = timeit(lambda: chain.forward(x, heads, act_out=acts, rms=(mean, var), xn_out=xn))
    t_infer = timeit(lambda: chain.forward(x, heads, rms=(mean, var)))
    flops = 2 * rows * sum(w.shape[0] * w.shape[1] for w, _, _ in layers)
    print(f'rows {rows}: split kernel {chain.split_products(rows, 0)}: training forward {t_train:.1f} us ({flops / t_train * 1e-6:.1f} TF), '
          f'inference {t_infer:.1f} us ({flops / t_infer * 1e-6:.1f} TF)   [incl. the plane pack launch when split]')
