"""Chain launches at a data-parallel rank's minibatch sizes (4,096 / 8,192 rows) on the network laid out like the
optimiser's arena (the pipelined kernels need it): time per launch of the training / inference forward and of the
backward, and the forward's phase stamps.  Variants are selected by the environment (RLG_CHAIN_PIPE1, RLG_PIPE1_WAVES,
RLG_CHAIN_WAVES), one process each.      python tools/exp/rank_chain_probe.py [rows ...] [--phases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rl_games_amd import ops, _lib
dev = 'cuda:0'
in_dim, units, out_dim = 108, [400, 200, 100], 22
rows_list = [int(a) for a in sys.argv[1:] if a.isdigit()] or [4096, 8192]
phases = '--phases' in sys.argv
g = torch.Generator().manual_seed(0)
shapes = []
last = in_dim
for u in units + [out_dim]:
    shapes.append((u, last)); last = u
flat = torch.empty(sum(u * i + u for u, i in shapes), device=dev)
layers, off = [], 0
for u, i in shapes:
    wv, bv = flat[off:off + u * i].view(u, i), flat[off + u * i:off + u * i + u]
    wv.copy_(torch.randn(u, i, generator=g) / i ** 0.5); bv.copy_(0.1 * torch.randn(u, generator=g))
    off += u * i + u
    layers.append((wv, bv, 'elu'))
layers[-1] = (layers[-1][0], layers[-1][1], 'None')
chain = ops.MlpChain(layers, dev)
macs_f = sum(u * i for u, i in shapes); macs_b = sum(u * i for u, i in shapes[1:])
env = {k: os.environ.get(k) for k in ('RLG_CHAIN_PIPE1', 'RLG_PIPE1_WAVES', 'RLG_CHAIN_WAVES')}

def timeit(fn, reps=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

for rows in rows_list:
    x = (3 * torch.randn(rows, in_dim, generator=g) + 1).to(dev)
    mean = torch.zeros(in_dim, dtype=torch.float64, device=dev) + 1.0
    var = torch.ones(in_dim, dtype=torch.float64, device=dev) * 9.0
    heads = torch.empty(rows, out_dim, device=dev)
    acts = [torch.empty(rows, u, device=dev) for u in units]
    xn = torch.empty(rows, in_dim, device=dev)
    dzs = [torch.empty(rows, u, device=dev) for u in units]
    d_heads = torch.randn(rows, out_dim, generator=g).to(dev)
    nblk = chain.num_blocks(rows, 1)
    parts = [torch.empty(nblk * u, dtype=torch.float64, device=dev) for u in units]
    tf = timeit(lambda: chain.forward(x, heads, act_out=acts, rms=(mean, var), xn_out=xn))
    ti = timeit(lambda: chain.forward(x, heads, rms=(mean, var)))
    tb = timeit(lambda: chain.backward(d_heads, acts, dzs, parts))
    print(f'{env} rows {rows}: forward train {tf:6.1f} us ({2e-6 * rows * macs_f / tf:5.1f} TF)  infer {ti:6.1f} us  backward {tb:6.1f} us ({2e-6 * rows * macs_b / tb:5.1f} TF)', flush=True)
    if phases:
        nb = chain.num_blocks(rows, 0)
        dbg = torch.zeros(nb * 4 * 32, dtype=torch.int64, device=dev)
        _lib.load().rlg_mlp_chain_debug_stamps(dbg.data_ptr())
        chain.forward(x, heads, rms=(mean, var))
        torch.cuda.synchronize()
        _lib.load().rlg_mlp_chain_debug_stamps(None)
        d = dbg.view(nb, 4, 32).cpu().double()
        n = int((d[0, 0] != 0).sum())
        sel = d[:min(nb, 256), :, :n]
        tot = 0.0
        print(f'   forward (infer) phase stamps, first {sel.shape[0]} workgroups, waves 0-3: mean ticks (min .. max)')
        for k in range(1, n):
            seg = sel[:, :, k] - sel[:, :, k - 1]
            tot += seg.mean().item()
            print(f'     stamp {k:2d}  +{seg.mean().item():8.0f}  ({seg.min().item():7.0f} .. {seg.max().item():7.0f})  t = {tot:8.0f}')
