"""Shader-clock phase stamps of the pipelined forward (csrc/mlp_chain.hip: mlp_chain_fwd_pipe_kernel) next to the MFMA
cycles each phase would take at full issue rate (32 cycles per v_mfma_f32_16x16x4_f32 per SIMD)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rl_games_amd import ops, _lib
dev = 'cuda:0'
in_dim, units, out_dim = 108, [400, 200, 100], 22
rows = 32768
g = torch.Generator().manual_seed(0)
layers, last = [], in_dim
for u in units + [out_dim]:
    layers.append(((torch.randn(u, last, generator=g) / last ** 0.5).to(dev), (0.1 * torch.randn(u, generator=g)).to(dev), 'elu'))
    last = u
layers[-1] = (layers[-1][0], layers[-1][1], 'None')
chain = ops.MlpChain(layers, dev)
x = (3 * torch.randn(rows, in_dim, generator=g) + 1).to(dev)
mean = torch.zeros(in_dim, dtype=torch.float64, device=dev) + 1.0
var = torch.ones(in_dim, dtype=torch.float64, device=dev) * 9.0
heads = torch.empty(rows, out_dim, device=dev)
acts = [torch.empty(rows, u, device=dev) for u in units]
xn = torch.empty(rows, in_dim, device=dev)
dims = [in_dim] + units + [out_dim]
for G in (4, 2):
    nb = chain.num_blocks(rows, 0, G)
    names = ['start', 'prologue', 'prologue barrier']
    ideal = [0, 0, 0]
    for L in range(4):
        KC, NOB = (dims[L] + 15) // 16, (dims[L + 1] + 15) // 16
        full, rem = NOB // 4, (NOB % 4) * G
        names += [f'L{L} whole units', f'L{L} remainder units', f'L{L} flush', f'L{L} barrier']
        ideal += [full * KC * 4 * G * 32, -(-rem // 4) * KC * 4 * 32, 0, 0]
    for train in (False, True):
        for rep in range(3):        # warm
            chain.forward(x, heads, act_out=acts if train else None, rms=(mean, var), xn_out=xn if train else None, groups=G)
        dbg = torch.zeros(nb * 4 * 32, dtype=torch.int64, device=dev)
        _lib.load().rlg_mlp_chain_debug_stamps(dbg.data_ptr())
        chain.forward(x, heads, act_out=acts if train else None, rms=(mean, var), xn_out=xn if train else None, groups=G)
        torch.cuda.synchronize()
        _lib.load().rlg_mlp_chain_debug_stamps(None)
        d = dbg.view(nb, 4, 32).cpu().double()
        n = int((d[0, 0] != 0).sum())
        per_cu = 256 if G == 4 else 512
        for label, sel in (('first round', d[:per_cu, :, :n]), ('last round', d[-per_cu:, :, :n])):
            print(f'G={G} {"train" if train else "infer"} {label}: phase, mean ticks (min..max over waves and workgroups), ideal MFMA cycles')
            tot = 0
            for k in range(1, n):
                seg = sel[:, :, k] - sel[:, :, k - 1]
                tot += seg.mean().item()
                print(f'   {names[k] if k < len(names) else k:22s} {seg.mean().item():9.0f}  ({seg.min().item():8.0f} .. {seg.max().item():8.0f})   ideal {ideal[k] if k < len(ideal) else 0:7d}   t = {tot:9.0f}')
        span = (d[:, :, n - 1].max() - d[:, :, 0].min()).item()
        print(f'   whole launch: {span:.0f} ticks from the first start stamp to the last end stamp')
