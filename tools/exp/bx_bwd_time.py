"""Backward chain: split-bf16 kernel (csrc/mlp_chain_bx.hip) against the exact-product kernel, back-to-back launches.
    python tools/exp/bx_bwd_time.py [rows]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rl_games_amd import ops

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
dev = 'cuda:0'
g = torch.Generator().manual_seed(0)
in_dim, units, out_dim = 108, [400, 200, 100], 22
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
G = int(os.environ.get("BX_G", "4"))
nblk = chain.num_blocks(rows, 1, G)
parts = [torch.empty(nblk * u, dtype=torch.float64, device=dev) for u in units]


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


flops = 2 * rows * sum(w.shape[0] * w.shape[1] for w, _, _ in layers[1:])
planes = chain.pack_planes(1, d_heads)
t_pack = timeit(lambda: chain.pack_planes(1, d_heads))
t_exact = timeit(lambda: chain.backward(d_heads, acts, dzs, parts, groups=G, split_products=False))
t_bx = timeit(lambda: chain.backward(d_heads, acts, dzs, parts, groups=G))
# with the PPO loss tile in front (what the epoch runs)
A = 21
gg = torch.Generator().manual_seed(1)
logstd = (0.1 * torch.randn(A, generator=gg) - 0.3).to(dev)
dd = {'actions': torch.randn(rows, A, generator=gg), 'old_neglogp': 1.4 * A + torch.randn(rows, generator=gg),
      'adv': torch.randn(rows, generator=gg), 'old_values': torch.randn(rows, generator=gg), 'returns': torch.randn(rows, generator=gg),
      'old_mu': 0.3 * torch.randn(rows, A, generator=gg), 'old_sigma': 0.5 + torch.rand(rows, A, generator=gg)}
dd = {k: v.to(dev) for k, v in dd.items()}
partials = torch.empty(nblk, ops.ppo_loss_partials_per_block(A), dtype=torch.float64, device=dev)
largs = (heads[:, 1:], logstd, heads[:, 0], dd['actions'], dd['old_neglogp'], dd['adv'], dd['old_values'], dd['returns'],
         dd['old_mu'], dd['old_sigma'], d_heads[:, 1:], d_heads[:, 0], partials, 0.2, 2.0, 1e-4)
desc = ops.ppo_loss_desc(*largs)
t_loss_exact = timeit(lambda: chain.backward(d_heads, acts, dzs, parts, groups=G, ppo_loss=desc, split_products=False))
t_loss_bx = timeit(lambda: chain.backward(d_heads, acts, dzs, parts, groups=G, ppo_loss=desc))
print(f'rows {rows}: with the loss tile: exact-product {t_loss_exact:.1f} us | split-bf16 incl. pack {t_loss_bx:.1f} us')
print(f'rows {rows}: pack {t_pack:.1f} us | exact-product backward {t_exact:.1f} us ({flops / t_exact * 1e-6:.1f} TFLOP/s) | '
      f'split-bf16 backward incl. pack {t_bx:.1f} us ({flops / t_bx * 1e-6:.1f} TFLOP/s fp32-equivalent)')
