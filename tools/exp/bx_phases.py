"""Shader-clock phase stamps (s_memtime, 100 MHz) of the split-bf16 backward (csrc/mlp_chain_bx.hip) next to the MFMA
cycles each phase would take at full issue rate (16 cycles per v_mfma_f32_16x16x32_bf16 per SIMD = 2/3 tick at 2.4 GHz)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rl_games_amd import ops, _lib
dev = 'cuda:0'
in_dim, units, out_dim = 108, [400, 200, 100], 22
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
g = torch.Generator().manual_seed(0)
layers, last = [], in_dim
for u in units + [out_dim]:
    layers.append(((torch.randn(u, last, generator=g) / last ** 0.5).to(dev), (0.1 * torch.randn(u, generator=g)).to(dev), 'elu'))
    last = u
layers[-1] = (layers[-1][0], layers[-1][1], 'None')
chain = ops.MlpChain(layers, dev)
x = torch.randn(rows, in_dim, generator=g).to(dev)
heads = torch.empty(rows, out_dim, device=dev)
acts = [torch.empty(rows, u, device=dev) for u in units]
chain.forward(x, heads, act_out=acts)
d_heads = torch.randn(rows, out_dim, generator=g).to(dev)
dzs = [torch.empty(rows, u, device=dev) for u in units]
nb = chain.num_blocks(rows, 1, 4)
parts = [torch.empty(nb * u, dtype=torch.float64, device=dev) for u in units]
dims = [in_dim] + units + [out_dim]
names = ['start', 'loss tile', 'prologue + barrier']
ideal = [0, 0, 0]
for L in range(3, 0, -1):
    KC, NOB = (dims[L + 1] + 31) // 32, (dims[L] + 15) // 16
    full, rem = NOB // 4, (NOB % 4) * 4
    names += [f'dZ{L - 1} units', f'dZ{L - 1} barrier']
    ideal += [-(-NOB // 4) * KC * 4 * 6 * 16, 0]
# (the fine stamps inside the two-block units of the last step exist only in a 4-wave build, RLG_BX_BWD_W=4: the 8-wave
#  kernel runs one block per unit and stamps the layers only; `ideal`: MFMA issue cycles of the wave with the most blocks at
#  3 products x 16 cycles per chunk and row group)
ideal = [0, 0, 0]
for L in range(3, 0, -1):
    KC, NOB = (dims[L + 1] + 31) // 32, (dims[L] + 15) // 16
    ideal += [-(-NOB // 8) * KC * 4 * 3 * 16, 0]
for rep in range(3):
    chain.backward(d_heads, acts, dzs, parts, groups=4)
dbg = torch.zeros(nb * 4 * 32, dtype=torch.int64, device=dev)
_lib.load().rlg_mlp_chain_debug_stamps(dbg.data_ptr())
chain.backward(d_heads, acts, dzs, parts, groups=4)
torch.cuda.synchronize()
_lib.load().rlg_mlp_chain_debug_stamps(None)
d = dbg.view(nb, 4, 32).cpu().double()
n = int((d[0, 0] != 0).sum())
for label, sel in (('first round', d[:256, :, :n]), ('last round', d[-256:, :, :n])):
    print(f'{label}: phase, mean s_memtime cycles (min..max over waves 0 - 3 and workgroups), ideal MFMA issue cycles of a wave')
    tot = 0
    for k in range(1, n):
        seg = sel[:, :, k] - sel[:, :, k - 1]
        tot += seg.mean().item()
        idl = ideal[k] if k < len(ideal) else 0
        print(f'   {names[k] if k < len(names) else k:24s} {seg.mean().item():9.0f}  ({seg.min().item():8.0f} .. {seg.max().item():8.0f})   ideal {idl:7d}   t = {tot:9.0f}')
span = (d[:, :, n - 1].max() - d[:, :, 0].min()).item()
print(f'   whole launch: {span:.0f} ticks from the first start stamp to the last end stamp')
starts = d[:, 0, 0] - d[:, 0, 0].min()
print('   start stamps of workgroups (ticks): first 256 max', starts[:256].max().item(), ' last 256 min', starts[-256:].min().item(), 'max', starts[-256:].max().item())
