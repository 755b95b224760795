import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rl_games_amd import ops
DEV = 'cuda:0'
def net(in_dim, units, out_dim, seed):
    g = torch.Generator().manual_seed(seed)
    layers, last = [], in_dim
    for u in list(units) + [out_dim]:
        layers.append(((torch.randn(u, last, generator=g) / last ** 0.5).to(DEV), (0.1 * torch.randn(u, generator=g)).to(DEV), 'None'))
        last = u
    return layers, g
for (i, u, o) in [(48, [32], 7), (48, [32], 8), (48, [32], 16), (32, [32], 7), (48, [16], 7), (48, [64], 7), (60, [64, 64], 9), (13, [20, 36], 5), (108, [400, 200, 100], 22)]:
    for rows in (1, 16, 100):
        layers, g = net(i, u, o, 1)
        chain = ops.MlpChain(layers, DEV)
        x = torch.randn(rows, i, generator=g).to(DEV)
        heads = torch.zeros(rows, o, device=DEV)
        acts = [torch.zeros(rows, w, device=DEV) for w in u]
        chain.forward(x, heads, act_out=acts, groups=1)
        a = x.double()
        refs = []
        for w, b, _ in layers:
            a = torch.addmm(b.double(), a, w.double().t()); refs.append(a)
        errs = [(got.double() - r).abs().max(0).values for got, r in zip(acts + [heads], refs)]
        bad = [(k, (e > 1e-4).nonzero().flatten().tolist()[:12]) for k, e in enumerate(errs) if (e > 1e-4).any()]
        print(f'W={os.environ.get("RLG_CHAIN_WAVES")} net {i}-{u}-{o} rows {rows}: ' + ('ok' if not bad else f'BAD layers/cols {bad}'), flush=True)
