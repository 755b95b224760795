"""A few launches of the split-bf16 forward (training form) and backward at 32,768 rows, humanoid network - the command
tools/gpu_r3_final2.sh runs under rocprofv3 --pmc."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rl_games_amd import ops
dev = 'cuda:0'
rows, dims = 32768, [108, 400, 200, 100, 22]
g = torch.Generator().manual_seed(0)
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
x = (3 * torch.randn(rows, dims[0], generator=g) + 1).to(dev)
mean = torch.zeros(dims[0], dtype=torch.float64, device=dev)
var = torch.ones(dims[0], dtype=torch.float64, device=dev)
heads = torch.empty(rows, dims[-1], device=dev)
acts = [torch.empty(rows, u, device=dev) for u in dims[1:-1]]
xn = torch.empty(rows, dims[0], device=dev)
d_heads = torch.randn(rows, dims[-1], generator=g).to(dev)
dzs = [torch.empty(rows, u, device=dev) for u in dims[1:-1]]
nb = chain.num_blocks(rows, 1)
parts = [torch.empty(nb * u, dtype=torch.float64, device=dev) for u in dims[1:-1]]
for _ in range(6):
    chain.forward(x, heads, act_out=acts, rms=(mean, var), xn_out=xn)
    chain.backward(d_heads, acts, dzs, parts)
torch.cuda.synchronize()
