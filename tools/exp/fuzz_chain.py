"""Random-shape fuzz of the fused MLP kernels against fp64 torch (forward, backward with and without the
loss tile, weight gradients).  python tools/exp/fuzz_chain.py [cases] [seed] [only this case]"""
import os, sys, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rl_games_amd import ops
DEV = 'cuda:0'
ACT = {'elu': torch.nn.functional.elu, 'relu': torch.relu, 'tanh': torch.tanh, 'None': lambda t: t}
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1       # run this case alone
bad = 0
for case in range(cases):
    in_dim = rng.choice([1, 3, 7, 12, 16, 33, 60, 108, 130])
    units = [rng.choice([4, 8, 20, 32, 52, 64, 100, 112, 128, 200, 256]) for _ in range(rng.choice([1, 2, 3]))]
    A = rng.choice([1, 2, 8, 21, 33])
    V = 1
    act = rng.choice(['elu', 'relu', 'tanh', 'None'])
    rows = rng.choice([1, 5, 16, 17, 100, 512, 1000, 4096, 4100, 16384, 20000])
    groups = rng.choice([0, 0, 1, 2, 4])
    if only >= 0 and case != only:
        continue
    g = torch.Generator().manual_seed(case)
    layers, last = [], in_dim
    for u in units + [V + A]:
        layers.append(((torch.randn(u, last, generator=g) / last ** 0.5).to(DEV), (0.1 * torch.randn(u, generator=g)).to(DEV), act))
        last = u
    layers[-1] = (layers[-1][0], layers[-1][1], 'None')
    try:
        chain = ops.MlpChain(layers, DEV)
    except NotImplementedError:
        continue
    x = torch.randn(rows, in_dim, generator=g).to(DEV)
    heads = torch.full((rows, V + A), float('nan'), device=DEV)
    acts = [torch.full((rows, u), float('nan'), device=DEV) for u in units]
    chain.forward(x, heads, act_out=acts, groups=groups)
    a = x.double()
    msgs = []
    for (w, b, an), got in zip(layers, acts + [heads]):
        a = ACT[an](torch.addmm(b.double(), a, w.double().t()))
        err = (got.double() - a).abs().max().item()
        if not (err <= 2e-4 * max(1.0, a.abs().max().item())):
            msgs.append(f'fwd err {err:.2e}')
    # backward with the loss tile vs loss kernel + backward, and vs fp64
    logstd = (0.1 * torch.randn(A, generator=g) - 0.3).to(DEV)
    def data():
        gg = torch.Generator().manual_seed(case + 1)
        d = {'actions': torch.randn(rows, A, generator=gg), 'old_neglogp': 1.4 * A + torch.randn(rows, generator=gg),
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
        out[fused] = [d_heads] + dzs + [p.view(nbw, -1).sum(0) for p in parts]
    for k, (p, q) in enumerate(zip(out[True], out[False])):
        if not torch.isfinite(p).all() or not torch.equal(p, q):
            msgs.append(f'bwd fused!=unfused item {k} maxdiff {(p.double() - q.double()).abs().max().item():.2e}')
    # dZ against fp64
    dgrad = out[False][0].double()
    for l in range(len(units), 0, -1):
        w = layers[l][0].double()
        h = acts[l - 1].double()
        dh = dgrad @ w
        if act == 'elu':
            dgrad = dh * torch.where(h > 0, torch.ones_like(h), h + 1)
        elif act == 'relu':
            dgrad = dh * (h > 0).double()
        elif act == 'tanh':
            dgrad = dh * (1 - h * h)
        else:
            dgrad = dh
        got = out[False][l]
        err = (got.double() - dgrad).abs().max().item()
        if not (err <= 2e-4 * max(1e-3, dgrad.abs().max().item())):
            msgs.append(f'dz{l - 1} err {err:.2e}')
        dgrad = got.double()
    # weight gradients
    jobs = [(out[False][0], acts[-1], torch.empty_like(layers[-1][0]))]
    for l in range(len(units) - 1, -1, -1):
        jobs.append((out[False][1 + l], acts[l - 1] if l > 0 else x, torch.empty_like(layers[l][0])))
    try:
        ok_jobs = [j for j in jobs if j[2].shape[1] % 4 == 0 and j[2].shape[1] >= 4 and (j[2].numel() % 4 == 0)]
        if ok_jobs:
            plan = ops.MlpDwPlan([tuple(j[2].shape) for j in ok_jobs], rows, DEV)
            plan.launch(ok_jobs)
            for dz, xx, gr in ok_jobs:
                ref = dz.double().t() @ xx.double()
                err = (gr.double() - ref).abs().max().item()
                if not (err <= 2e-4 * max(1e-3, ref.abs().max().item())):
                    msgs.append(f'dW {tuple(gr.shape)} err {err:.2e}')
    except NotImplementedError:
        pass
    torch.cuda.synchronize()
    status = 'ok' if not msgs else 'BAD ' + '; '.join(msgs)
    bad += bool(msgs)
    print(f'case {case}: in {in_dim} units {units} A {A} act {act} rows {rows} G {groups}: {status}', flush=True)
print(f'{bad} bad of {cases}')
