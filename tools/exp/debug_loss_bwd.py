import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rl_games_amd import ops
DEV = 'cuda:0'
def run(rows, units, A, in_dim, groups):
    V = 1
    g = torch.Generator().manual_seed(rows)
    layers, last = [], in_dim
    for u in list(units) + [V + A]:
        layers.append(((torch.randn(u, last, generator=g) / last ** 0.5).to(DEV), (0.1 * torch.randn(u, generator=g)).to(DEV), 'elu'))
        last = u
    layers[-1] = (layers[-1][0], layers[-1][1], 'None')
    chain = ops.MlpChain(layers, DEV)
    x = torch.randn(rows, in_dim, generator=g).to(DEV)
    logstd = (0.1 * torch.randn(A, generator=g) - 0.3).to(DEV)
    heads = torch.empty(rows, V + A, device=DEV)
    acts = [torch.empty(rows, u, device=DEV) for u in units]
    chain.forward(x, heads, act_out=acts, groups=groups)
    def data():
        gg = torch.Generator().manual_seed(rows + 1)
        d = {'actions': torch.randn(rows, A, generator=gg), 'old_neglogp': 25 + torch.randn(rows, generator=gg),
             'adv': torch.randn(rows, generator=gg), 'old_values': torch.randn(rows, generator=gg),
             'returns': torch.randn(rows, generator=gg), 'old_mu': 0.3 * torch.randn(rows, A, generator=gg),
             'old_sigma': 0.5 + torch.rand(rows, A, generator=gg)}
        return {k: v.to(DEV) for k, v in d.items()}
    out = {}
    for fused in (True, False):
        d = data()
        d_heads = torch.full((rows, V + A), float('nan'), device=DEV)
        dzs = [torch.full((rows, u), float('nan'), device=DEV) for u in units]
        nbw = chain.num_blocks(rows, 1, groups)
        parts = [torch.full((nbw * u,), float('nan'), dtype=torch.float64, device=DEV) for u in units]
        nblk = nbw if fused else ops.ppo_loss_blocks(rows)
        partials = torch.full((nblk, ops.ppo_loss_partials_per_block(A)), float('nan'), dtype=torch.float64, device=DEV)
        args = (heads[:, V:], logstd, heads[:, 0], d['actions'], d['old_neglogp'], d['adv'], d['old_values'],
                d['returns'], d['old_mu'], d['old_sigma'], d_heads[:, V:], d_heads[:, 0], partials, 0.2, 2.0, 1e-4)
        if fused:
            chain.backward(d_heads, acts, dzs, parts, groups=groups, ppo_loss=ops.ppo_loss_desc(*args))
        else:
            ops.ppo_loss_fused(*args)
            chain.backward(d_heads, acts, dzs, parts, groups=groups)
        torch.cuda.synchronize()
        out[fused] = [d_heads] + dzs + [p.view(nbw, -1) for p in parts] + [partials.sum(0)]
    names = ['d_heads'] + [f'dz{k}' for k in range(len(units))] + [f'bias_part{k}' for k in range(len(units))] + ['loss partial sums']
    for n, a, b in zip(names, out[True], out[False]):
        eq = torch.equal(a, b)
        md = (a.double() - b.double()).abs().max().item()
        print(f'rows {rows} units {units} G={groups}: {n:18s} equal {eq} maxdiff {md:.3e} finite {torch.isfinite(a).all().item()}')
for rows, units, A, in_dim, groups in [(512, [100, 52], 21, 12, 0), (512, [100, 52], 21, 12, 1), (4096, [400, 200, 100], 21, 108, 0), (1024, [32, 16], 3, 12, 0)]:
    for rep in range(2):
        run(rows, units, A, in_dim, groups)
