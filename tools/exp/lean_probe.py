"""The lean 16-row forward (csrc/mlp_chain.hip, mlp_chain_fwd_lean_kernel - experimental) against the pipelined forward and
fp64 at a data-parallel rank's minibatch sizes: max error of the heads / activations, time per launch (training and
inference form).      python tools/exp/lean_probe.py [rows ...]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rl_games_amd import _lib, ops  # noqa: E402

dev = torch.device('cuda:0')
in_dim, units, out_dim = 108, [400, 200, 100], 22
rows_list = [int(a) for a in sys.argv[1:] if a.isdigit()] or [4096, 8192, 37]
g = torch.Generator().manual_seed(0)
shapes, last = [], in_dim
for u in units + [out_dim]:
    shapes.append((u, last))
    last = u
flat = torch.empty(sum(u * i + u for u, i in shapes), device=dev)
layers, off = [], 0
for u, i in shapes:
    wv, bv = flat[off:off + u * i].view(u, i), flat[off + u * i:off + u * i + u]
    wv.copy_(torch.randn(u, i, generator=g) / i ** 0.5)
    bv.copy_(0.1 * torch.randn(u, generator=g))
    off += u * i + u
    layers.append((wv, bv, 'elu'))
layers[-1] = (layers[-1][0], layers[-1][1], 'None')
chain = ops.MlpChain(layers, dev)
lib = _lib.load()
n = len(layers)
I = (ctypes.c_int * n)
P = (ctypes.c_void_p * n)
LL = (ctypes.c_longlong * n)
ins, outs = I(*[s[1] for s in shapes]), I(*[s[0] for s in shapes])
acts_code = I(*([1] * (n - 1) + [0]))
nbytes = lib.rlg_mlp_chain_frags_bytes(n, ins, outs, 0)
print('fragment stream:', nbytes, 'bytes for', 4 * flat.numel(), 'bytes of parameters')
frags = torch.empty(nbytes // 4, device=dev)
wp, bp = P(*[l[0].data_ptr() for l in layers]), P(*[l[1].data_ptr() for l in layers])
st = _lib.stream_handle(dev)
_lib.check(lib.rlg_mlp_chain_pack_frags(n, wp, bp, ins, outs, 0, frags.data_ptr(), st), 'pack_frags')
nbytes_b = lib.rlg_mlp_chain_frags_bytes(n, ins, outs, 1)
bfrags = torch.empty(nbytes_b // 4, device=dev)
_lib.check(lib.rlg_mlp_chain_pack_frags(n, wp, None, ins, outs, 1, bfrags.data_ptr(), st), 'pack_frags (backward)')
print('backward fragment stream:', nbytes_b, 'bytes')


def lean_backward(d_heads, acts, dzs, parts, desc=None):
    ai = P(*[t.data_ptr() for t in acts] + [None])
    ald = LL(*[t.stride(0) for t in acts] + [0])
    dz = P(*[t.data_ptr() for t in dzs] + [None])
    dzld = LL(*[t.stride(0) for t in dzs] + [0])
    bpp = P(*[t.data_ptr() for t in parts] + [None])
    _lib.check(lib.rlg_mlp_chain_backward_lean(n, ins, outs, acts_code, ai, ald, d_heads.data_ptr(), d_heads.stride(0), dz, dzld, bpp,
                                               None if desc is None else ctypes.addressof(desc), d_heads.shape[0],
                                               bfrags.data_ptr(), st), 'backward_lean')



def lean_forward(x, heads, acts, rms, xn):
    outs_t = list(acts) + [heads] if acts is not None else [None] * (n - 1) + [heads]
    ao = P(*[None if t is None else t.data_ptr() for t in outs_t])
    ald = LL(*[0 if t is None else t.stride(0) for t in outs_t])
    _lib.check(lib.rlg_mlp_chain_forward_lean(n, bp, ins, outs, acts_code, ao, ald, x.data_ptr(), x.stride(0),
                                              rms[0].data_ptr(), rms[1].data_ptr(), 1e-5,
                                              None if xn is None else xn.data_ptr(), None, None, None, None, None,
                                              x.shape[0], frags.data_ptr(), st), 'forward_lean')


def timeit(fn, reps=100):
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


for rows in rows_list:
    x = (3 * torch.randn(rows, in_dim, generator=g) + 1).to(dev)
    mean = torch.zeros(in_dim, dtype=torch.float64, device=dev) + 1.0
    var = torch.ones(in_dim, dtype=torch.float64, device=dev) * 9.0
    # fp64 reference
    a = ((x.double() - mean) / torch.sqrt(var.float() + 1e-5).double()).clamp(-5, 5)
    ref = []
    for (w, b, act) in layers:
        a = torch.addmm(b.double(), a, w.double().t())
        if act == 'elu':
            a = torch.nn.functional.elu(a)
        ref.append(a)
    res = {}
    for name in ('pipe', 'lean'):
        heads = torch.full((rows, out_dim), float('nan'), device=dev)
        acts = [torch.full((rows, u), float('nan'), device=dev) for u in units]
        xn = torch.full((rows, in_dim), float('nan'), device=dev)
        if name == 'pipe':
            chain.forward(x, heads, act_out=acts, rms=(mean, var), xn_out=xn)
        else:
            lean_forward(x, heads, acts, (mean, var), xn)
        torch.cuda.synchronize()
        res[name] = [xn] + acts + [heads]
        errs = [float((t.double() - r).abs().max() / r.abs().max()) for t, r in zip(acts + [heads], ref)]
        print(f'rows {rows} {name}: finite {all(bool(torch.isfinite(t).all()) for t in res[name])}  max error / scale per layer vs fp64: '
              + ' '.join(f'{e:.2e}' for e in errs))
    print(f'rows {rows}: lean vs pipe max abs diff per tensor:', ' '.join(f'{float((p - q).abs().max()):.2e}' for p, q in zip(res['pipe'], res['lean'])))
    heads = torch.empty(rows, out_dim, device=dev)
    acts = [torch.empty(rows, u, device=dev) for u in units]
    xn = torch.empty(rows, in_dim, device=dev)
    t_pipe = timeit(lambda: chain.forward(x, heads, act_out=acts, rms=(mean, var), xn_out=xn))
    t_pipe_i = timeit(lambda: chain.forward(x, heads, rms=(mean, var)))
    t_lean = timeit(lambda: lean_forward(x, heads, acts, (mean, var), xn))
    t_lean_i = timeit(lambda: lean_forward(x, heads, None, (mean, var), None))
    print(f'rows {rows}: pipelined forward train {t_pipe:6.1f} us  infer {t_pipe_i:6.1f} us   |   lean train {t_lean:6.1f} us  infer {t_lean_i:6.1f} us', flush=True)
    # backward: dZ chain + bias partial sums against the pipelined kernel (same inputs), without and with the loss tile
    d_heads = torch.randn(rows, out_dim, generator=g).to(dev)
    nblk = chain.num_blocks(rows, 1)
    outb = {}
    for name in ('pipe', 'lean'):
        dzs = [torch.full((rows, u), float('nan'), device=dev) for u in units]
        parts = [torch.full((nblk * u,), float('nan'), dtype=torch.float64, device=dev) for u in units]
        if name == 'pipe':
            chain.backward(d_heads, acts, dzs, parts)
        else:
            lean_backward(d_heads, acts, dzs, parts)
        torch.cuda.synchronize()
        outb[name] = dzs + parts
    print(f'rows {rows}: backward lean vs pipe, max abs diff / scale per tensor (dZ0 dZ1 dZ2 | bias partials):',
          ' '.join(f'{float((p.double() - q.double()).abs().max() / q.double().abs().max()):.2e}' for p, q in zip(outb['lean'], outb['pipe'])),
          'finite', all(bool(torch.isfinite(t).all()) for t in outb['lean']))
    A = out_dim - 1
    z = lambda *s_: torch.zeros(*s_, device=dev)
    outl = {}
    for name in ('pipe', 'lean'):
        dh = torch.full((rows, out_dim), float('nan'), device=dev)
        dzs = [torch.empty(rows, u, device=dev) for u in units]
        parts = [torch.empty(nblk * u, dtype=torch.float64, device=dev) for u in units]
        partials = torch.full((nblk, ops.ppo_loss_partials_per_block(A)), float('nan'), dtype=torch.float64, device=dev)
        gg = torch.Generator().manual_seed(5)
        old_mu, old_sigma = torch.randn(rows, A, generator=gg).to(dev), (0.5 + torch.rand(rows, A, generator=gg)).to(dev)
        desc = ops.ppo_loss_desc(heads[:, 1:], z(A), heads[:, 0], torch.randn(rows, A, generator=gg).to(dev), torch.randn(rows, generator=gg).to(dev),
                                 torch.randn(rows, generator=gg).to(dev), torch.randn(rows, generator=gg).to(dev), torch.randn(rows, generator=gg).to(dev),
                                 old_mu, old_sigma, dh[:, 1:], dh[:, 0], partials, 0.2, 2.0, 1e-4, clip_value=True, smooth=False, bound_kind=1)
        if name == 'pipe':
            chain.backward(dh, acts, dzs, parts, ppo_loss=desc)
        else:
            lean_backward(dh, acts, dzs, parts, desc)
        torch.cuda.synchronize()
        outl[name] = [dh, partials] + dzs + parts
    print(f'rows {rows}: backward WITH the loss tile, lean vs pipe max abs diff / scale (d heads, loss partials, dZ..., bias partials...):',
          ' '.join(f'{float((p.double() - q.double()).abs().max() / q.double().abs().max()):.2e}' for p, q in zip(outl['lean'], outl['pipe'])))
    dzs = [torch.empty(rows, u, device=dev) for u in units]
    parts = [torch.empty(nblk * u, dtype=torch.float64, device=dev) for u in units]
    tb_pipe = timeit(lambda: chain.backward(d_heads, acts, dzs, parts))
    tb_lean = timeit(lambda: lean_backward(d_heads, acts, dzs, parts))
    print(f'rows {rows}: pipelined backward {tb_pipe:6.1f} us   |   lean backward {tb_lean:6.1f} us', flush=True)

if '--phases' in sys.argv:
    rows = rows_list[0]
    x = (3 * torch.randn(rows, in_dim, generator=g) + 1).to(dev)
    mean = torch.zeros(in_dim, dtype=torch.float64, device=dev) + 1.0
    var = torch.ones(in_dim, dtype=torch.float64, device=dev) * 9.0
    heads = torch.empty(rows, out_dim, device=dev)
    nb = (rows + 15) // 16
    for _ in range(3):
        lean_forward(x, heads, None, (mean, var), None)
    dbg = torch.zeros(nb * 4 * 32, dtype=torch.int64, device=dev)
    lib.rlg_mlp_chain_debug_stamps(dbg.data_ptr())
    lean_forward(x, heads, None, (mean, var), None)
    torch.cuda.synchronize()
    lib.rlg_mlp_chain_debug_stamps(None)
    d = dbg.view(nb, 4, 32).cpu().double()
    n_st = int((d[0, 0] != 0).sum())
    sel = d[:min(nb, 256), :, :n_st]
    tot = 0.0
    print(f'lean forward (infer) phase stamps, rows {rows}, first {sel.shape[0]} workgroups, waves 0-3: mean ticks (min .. max)')
    print('  (1 behind the prologue + bias-one barrier, then per layer: units done, behind the barrier)')
    for k in range(1, n_st):
        seg = sel[:, :, k] - sel[:, :, k - 1]
        tot += seg.mean().item()
        print(f'     stamp {k:2d}  +{seg.mean().item():8.0f}  ({seg.min().item():7.0f} .. {seg.max().item():7.0f})  t = {tot:8.0f}')
