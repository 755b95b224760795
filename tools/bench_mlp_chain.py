"""Microbenchmark of the fused MLP kernels (csrc/mlp_chain.hip, csrc/mlp_dw.hip) on one MI355X.

    python tools/bench_mlp_chain.py [--rows 32768] [--net humanoid|ant] [--reps 50]

Prints us per launch and useful TFLOP/s (2*rows*sum(in*out), tile padding not counted) for
the training forward (activations written), the inference forward, the dX chain and the weight
gradients, for every row-group setting, next to the per-layer library path (addmm + elu, mm).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

NETS = {'humanoid': (108, [400, 200, 100], 22), 'ant': (60, [256, 128, 64], 9)}


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, nargs='+', default=[32768])
    ap.add_argument('--net', default='humanoid')
    ap.add_argument('--reps', type=int, default=50)
    ap.add_argument('--no-lib', action='store_true', help='skip the library comparison rows')
    ap.add_argument('--phases', action='store_true', help='per-phase shader-clock breakdown of the forward')
    ap.add_argument('--groups', type=int, nargs='+', default=[4, 2, 1])
    ap.add_argument('--act', default='elu', help='hidden activation: elu / relu / tanh / None')
    ap.add_argument('--cold-x', type=int, default=0, help='rotate the forward input over this many buffers (cold observations)')
    ap.add_argument('--dw-blocks', type=int, nargs='+', default=[256, 512, 1024])
    args = ap.parse_args()
    from rl_games_amd import ops
    dev = 'cuda:0'
    in_dim, units, out_dim = NETS[args.net]
    g = torch.Generator().manual_seed(0)
    layers, last = [], in_dim
    for u in units + [out_dim]:
        layers.append(((torch.randn(u, last, generator=g) / last ** 0.5).to(dev), (0.1 * torch.randn(u, generator=g)).to(dev), args.act))
        last = u
    layers[-1] = (layers[-1][0], layers[-1][1], 'None')
    chain = ops.MlpChain(layers, dev)
    macs_f = sum(w.shape[0] * w.shape[1] for w, _, _ in layers)
    macs_b = sum(w.shape[0] * w.shape[1] for w, _, _ in layers[1:])
    for rows in args.rows:
        x = (3 * torch.randn(rows, in_dim, generator=g) + 1).to(dev)
        mean = torch.zeros(in_dim, dtype=torch.float64, device=dev) + 1.0
        var = torch.ones(in_dim, dtype=torch.float64, device=dev) * 9.0
        heads = torch.empty(rows, out_dim, device=dev)
        acts = [torch.empty(rows, u, device=dev) for u in units]
        xn = torch.empty(rows, in_dim, device=dev)
        dzs = [torch.empty(rows, u, device=dev) for u in units]
        d_heads = torch.randn(rows, out_dim, generator=g).to(dev)
        print(f'== {args.net} rows {rows}: forward {2e-9 * rows * macs_f:.2f} GFLOP, dX {2e-9 * rows * macs_b:.2f}, dW {2e-9 * rows * macs_f:.2f}')
        for G in args.groups:
            if G > chain.max_groups[0]:
                continue
            nblk = chain.num_blocks(rows, 1, G)
            parts = [torch.empty(nblk * u, dtype=torch.float64, device=dev) for u in units]
            t = timeit(lambda: chain.forward(x, heads, act_out=acts, rms=(mean, var), xn_out=xn, groups=G), args.reps)
            print(f'  G={G} forward train   {t:8.1f} us  {2e-6 * rows * macs_f / t:6.1f} TFLOP/s')
            if args.cold_x > 1:
                xs = [x.clone() for _ in range(args.cold_x)]
                state = {'k': 0}

                def cold():
                    state['k'] = (state['k'] + 1) % len(xs)
                    chain.forward(xs[state['k']], heads, act_out=acts, rms=(mean, var), xn_out=xn, groups=G)
                t = timeit(cold, args.reps)
                print(f'  G={G} forward train, observations rotating over {len(xs)} buffers  {t:8.1f} us')
                del xs
            t = timeit(lambda: chain.forward(x, heads, rms=(mean, var), groups=G), args.reps)
            print(f'  G={G} forward infer   {t:8.1f} us  {2e-6 * rows * macs_f / t:6.1f} TFLOP/s')
            if args.phases:
                from rl_games_amd import _lib
                nb = chain.num_blocks(rows, 0, G)
                for train in (False, True):
                    dbg = torch.zeros(nb * 4 * 32, dtype=torch.int64, device=dev)
                    _lib.load().rlg_mlp_chain_debug_stamps(dbg.data_ptr())
                    if train:
                        chain.forward(x, heads, act_out=acts, rms=(mean, var), xn_out=xn, groups=G)
                    else:
                        chain.forward(x, heads, rms=(mean, var), groups=G)
                    torch.cuda.synchronize()
                    _lib.load().rlg_mlp_chain_debug_stamps(None)
                    d = dbg.view(nb, 4, 32).cpu().double()
                    n = int((d[0, 0] != 0).sum())
                    # first-round blocks only (they start together): blocks 0..255
                    sel = d[:min(nb, 256), :, :n]
                    rel = sel - sel[:, :, :1]
                    names = ['start', 'prologue done', 'barrier']
                    for l in range(len(layers)):
                        if l < 2:
                            names += [f'L{l} u0 batches', f'L{l} u0 tail', f'L{l} u0 epilogue', f'L{l} u1 batches', f'L{l} u1 tail', f'L{l} u1 epilogue']
                        names += [f'L{l} whole blocks', f'L{l} remainder', f'L{l} barrier']
                    print(f'    phases ({"train" if train else "infer"}), mean shader-clock ticks since block start over the first {sel.shape[0]} blocks (per wave min/mean/max of the phase length):')
                    for k in range(1, n):
                        seg = sel[:, :, k] - sel[:, :, k - 1]
                        print(f'      {names[k] if k < len(names) else k:22s} +{seg.mean().item():9.0f}  (min {seg.min().item():8.0f} max {seg.max().item():8.0f})   t={rel[:, :, k].mean().item():9.0f}')
            if G <= chain.max_groups[1]:
                t = timeit(lambda: chain.backward(d_heads, acts, dzs, parts, groups=G), args.reps)
                print(f'  G={G} backward (dX)   {t:8.1f} us  {2e-6 * rows * macs_b / t:6.1f} TFLOP/s')
        # weight gradients
        shapes = [(w.shape[0], w.shape[1]) for w, _, _ in layers]
        grads = [torch.empty(s, device=dev) for s in shapes]
        ins = [xn] + acts
        outs = dzs + [d_heads]
        jobs = sorted([(outs[l], ins[l], grads[l]) for l in range(len(layers))], key=lambda j: -j[2].numel())
        for tb in args.dw_blocks:
            plan = ops.MlpDwPlan([tuple(j[2].shape) for j in jobs], rows, dev, target_blocks=tb)
            t = timeit(lambda: plan.launch(jobs), args.reps)
            print(f'  dW (16x16x4 MFMA, target_blocks {tb}, ksplit {[plan.plan(k)[3] for k in range(plan.n)]})  {t:8.1f} us  '
                  f'{2e-6 * rows * macs_f / t:6.1f} TFLOP/s')
            # as the engine launches it: bias gradients (column sums of the backward launch's
            # per-block fp64 partials) ride along in the finalise kernel
            nblk = chain.num_blocks(rows, 1, 0)
            cparts = [torch.randn(nblk * u, dtype=torch.float64, device=dev) for u in units]
            couts = [torch.empty(u, device=dev) for u in units]
            colsums = [(cparts[k], nblk, units[k], couts[k]) for k in range(len(units))]
            t = timeit(lambda: plan.launch(jobs, colsums), args.reps)
            print(f'     + bias column sums of {nblk} partial rows in the finalise launch   {t:8.1f} us')

        if args.no_lib:
            continue

        # per-layer library path
        def lib_fwd():
            a = torch.empty_like(xn)
            ops.rms_apply(x, mean, var, 1e-5, 0, out=a)
            for (w, b, act), h in zip(layers[:-1], acts):
                torch.addmm(b, a, w.t(), out=h)
                torch.ops.aten.elu.out(h, out=h)
                a = h
            torch.addmm(layers[-1][1], a, layers[-1][0].t(), out=heads)
        t = timeit(lib_fwd, args.reps)
        print(f'  library forward (rms_apply + addmm + elu per layer) {t:8.1f} us  {2e-6 * rows * macs_f / t:6.1f} TFLOP/s')

        def lib_bwd():
            d = d_heads
            for l in range(len(units), 0, -1):
                torch.mm(d, layers[l][0], out=dzs[l - 1])
                h = acts[l - 1]
                dzs[l - 1].mul_(torch.where(h > 0, 1.0, h + 1))
                d = dzs[l - 1]
        t = timeit(lib_bwd, args.reps)
        print(f'  library dX (mm + eager act-backward per layer)      {t:8.1f} us')

        def lib_dw():
            for dz, xx, gr in jobs:
                torch.mm(dz.t(), xx, out=gr)
        t = timeit(lib_dw, args.reps)
        print(f'  library dW (4 x mm)                                 {t:8.1f} us  {2e-6 * rows * macs_f / t:6.1f} TFLOP/s')


if __name__ == '__main__':
    main()
