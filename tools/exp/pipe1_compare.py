"""Pipelined 16-row chain kernels against the unit-structured ones and against fp64, on arena-packed networks at a
rank's sizes: forward (training and inference form, with the normaliser) and backward.  Two child processes
(RLG_CHAIN_PIPE1=1 / 0: the choice is a process-wide setting of the library), outputs compared by the parent.
    python tools/exp/pipe1_compare.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
NETS = {'ant': (60, [256, 128, 64], 9), 'humanoid': (108, [400, 200, 100], 22), 'odd': (13, [20, 36], 5)}
ROWS = [4096, 4100, 8192, 37]

def child(tag):
    from rl_games_amd import ops
    dev = 'cuda:0'
    out = {}
    for name, (in_dim, units, out_dim) in NETS.items():
        g = torch.Generator().manual_seed(3)
        shapes, last = [], in_dim
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
        for rows in ROWS:
            x = (3 * torch.randn(rows, in_dim, generator=g) + 1).to(dev)
            mean = torch.randn(in_dim, generator=g, dtype=torch.float64).to(dev)
            var = (torch.rand(in_dim, generator=g, dtype=torch.float64) * 4 + 0.1).to(dev)
            heads_t = torch.full((rows, out_dim), float('nan'), device=dev)
            acts = [torch.full((rows, u), float('nan'), device=dev) for u in units]
            xn = torch.full((rows, in_dim), float('nan'), device=dev)
            chain.forward(x, heads_t, act_out=acts, rms=(mean, var), xn_out=xn)
            heads_i = torch.full((rows, out_dim), float('nan'), device=dev)
            chain.forward(x, heads_i, rms=(mean, var))
            d_heads = torch.randn(rows, out_dim, generator=g).to(dev)
            dzs = [torch.full((rows, u), float('nan'), device=dev) for u in units]
            nblk = chain.num_blocks(rows, 1)
            parts = [torch.full((nblk * u,), float('nan'), dtype=torch.float64, device=dev) for u in units]
            chain.backward(d_heads, acts, dzs, parts)
            # fp64 reference
            a = torch.clamp((x.double() - mean) / torch.sqrt(var + 1e-5), -5, 5).float().double()
            ws = [w.double() for w, _, _ in layers]
            pre = []
            for (w, b, an), w64 in zip(layers, ws):
                z = torch.addmm(b.double(), a, w64.t()); z.requires_grad_(True); z.retain_grad(); pre.append(z)
                a = torch.nn.functional.elu(z) if an == 'elu' else z
            a.backward(d_heads.double())
            out[(name, rows)] = {'heads_t': heads_t.cpu(), 'heads_i': heads_i.cpu(), 'acts': [t.cpu() for t in acts],
                                 'dzs': [t.cpu() for t in dzs], 'parts': [t.cpu() for t in parts],
                                 'ref_heads': a.detach().cpu(), 'ref_dz': [z.grad.cpu() for z in pre[:-1]]}
    torch.save(out, f'/tmp/pipe1_compare_{tag}.pt')

if len(sys.argv) > 1:
    child(sys.argv[1])
    sys.exit(0)
for tag, env in (('pipe', {'RLG_CHAIN_PIPE1': '1'}), ('pipe8', {'RLG_CHAIN_PIPE1': '1', 'RLG_PIPE1_WAVES': '8'}), ('unit', {'RLG_CHAIN_PIPE1': '0'})):
    subprocess.run([sys.executable, os.path.abspath(__file__), tag], env=dict(os.environ, **env), check=True)
res = {t: torch.load(f'/tmp/pipe1_compare_{t}.pt') for t in ('pipe', 'pipe8', 'unit')}
def rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()
for key in res['unit']:
    u = res['unit'][key]
    for t in ('pipe', 'pipe8'):
        p = res[t][key]
        line = [f'{key[0]:9s} rows {key[1]:5d} {t:5s}:']
        line.append(f'heads train vs unit {rel(p["heads_t"], u["heads_t"]):.1e} vs fp64 {rel(p["heads_t"], u["ref_heads"]):.1e} (unit vs fp64 {rel(u["heads_t"], u["ref_heads"]):.1e})')
        line.append(f'infer==train {bool(torch.equal(p["heads_i"], p["heads_t"]))}')
        line.append('dZ vs fp64 ' + ' '.join(f'{rel(a, b):.1e}' for a, b in zip(p['dzs'], u['ref_dz'])) +
                    ' (unit ' + ' '.join(f'{rel(a, b):.1e}' for a, b in zip(u['dzs'], u['ref_dz'])) + ')')
        line.append('bias partial sums vs unit ' + ' '.join(f'{rel(a.view(-1, s.shape[1]).sum(0), b.view(-1, s.shape[1]).sum(0)):.1e}'
                                                          for a, b, s in zip(p['parts'], u['parts'], u['dzs'])))
        nonfinite = any(not torch.isfinite(t_).all() for t_ in [p['heads_t'], p['heads_i']] + p['acts'] + p['dzs'])
        line.append(f'non-finite {nonfinite}')
        print('  '.join(line))
