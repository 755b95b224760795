"""Shader-clock phase stamps (s_memtime = core clock cycles) of the split-product forward (csrc/mlp_chain_bx_fwd.hip), humanoid
network; ideal = MFMA cycles at 16 per v_mfma_f32_16x16x32_bf16 for the wave with the most blocks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rl_games_amd import ops, _lib
dev = 'cuda:0'
in_dim, units, out_dim = 108, [400, 200, 100], 22
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
train = (sys.argv[2] != 'infer') if len(sys.argv) > 2 else True
g = torch.Generator().manual_seed(0)
dims = [in_dim] + units + [out_dim]
flat = torch.empty(sum(o * i + o for i, o in zip(dims[:-1], dims[1:])), device=dev)
layers, off = [], 0
for i, o in zip(dims[:-1], dims[1:]):
    w, b = flat[off:off + o * i].view(o, i), flat[off + o * i:off + o * i + o]
    w.copy_(torch.randn(o, i, generator=g) / i ** 0.5)
    b.copy_(0.1 * torch.randn(o, generator=g))
    off += o * i + o
    layers.append((w, b, 'elu'))
layers[-1] = (layers[-1][0], layers[-1][1], 'None')
chain = ops.MlpChain(layers, dev)
x = (3 * torch.randn(rows, in_dim, generator=g) + 1).to(dev)
mean = torch.zeros(in_dim, dtype=torch.float64, device=dev)
var = torch.ones(in_dim, dtype=torch.float64, device=dev)
heads = torch.empty(rows, out_dim, device=dev)
acts = [torch.empty(rows, u, device=dev) for u in units] if train else None
xn = torch.empty(rows, in_dim, device=dev) if train else None
nb = (rows + 63) // 64
# (round 6, fp16 planes: the humanoid network's tiles fit without a windowed layer - one 'units' + 'barrier' stamp pair per layer)
names = ['start', 'obs requested', 'stats written', 'stats barrier', 'prologue + barrier'] + [x for L in range(len(dims) - 1) for x in (f'L{L} units', f'L{L} barrier')]
for rep in range(3):
    chain.forward(x, heads, act_out=acts, rms=(mean, var), xn_out=xn)
dbg = torch.zeros(nb * 4 * 32, dtype=torch.int64, device=dev)
_lib.load().rlg_mlp_chain_debug_stamps(dbg.data_ptr())
chain.forward(x, heads, act_out=acts, rms=(mean, var), xn_out=xn)
torch.cuda.synchronize()
_lib.load().rlg_mlp_chain_debug_stamps(None)
d = dbg.view(nb, 4, 32).cpu().double()
n = int((d[0, 0] != 0).sum())
for label, sel in (('first round', d[:256, :, :n]), ('last round', d[-256:, :, :n])):
    print(f'{"train" if train else "infer"} rows {rows} {label}: phase, mean cycles (min..max over waves and workgroups)')
    tot = 0
    for k in range(1, n):
        seg = sel[:, :, k] - sel[:, :, k - 1]
        tot += seg.mean().item()
        print(f'   {names[k] if k < len(names) else k:22s} {seg.mean().item():9.0f}  ({seg.min().item():8.0f} .. {seg.max().item():8.0f})   t = {tot:9.0f}')
