"""Repro of the wrong dZ that the 8-wave backward instance produces when its MFMA accumulators are NOT
pinned to AGPRs and no explicit s_nop precedes the epilogue: where are the wrong elements?

Build the variant next to the product library and point RLG_HIP_LIB at it:
    cd rl_games_amd/csrc
    sed 's/asm volatile("" : "+a"(acc\[f\]\[g\]));/;/; s/asm volatile("" : "+a"(acc_odd));/;/; s/asm volatile("s_nop 0" : "+a"(acc\[0\]\[0\]));/;/' \
        mlp_chain.hip > _v.hip
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -I. -c _v.hip -o /tmp/v.o
    hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v mlp_chain.o) /tmp/v.o -o ../../tools/exp/_build/librlg_nopin.so
    RLG_HIP_LIB=$PWD/../../tools/exp/_build/librlg_nopin.so python ../../tools/exp/chain_nopin_repro.py
(python tools/audit_mfma.py tools/exp/_build/librlg_nopin.so lists the reads inside the hazard window;
profiles/r2_mfma_hazard_probes.txt has the output of both.)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rl_games_amd import ops
DEV = 'cuda:0'
in_dim, units, A, rows = 7, [128, 32], 8, 512
g = torch.Generator().manual_seed(22)
layers, last = [], in_dim
for u in units + [1 + A]:
    layers.append(((torch.randn(u, last, generator=g) / last ** 0.5).to(DEV), (0.1 * torch.randn(u, generator=g)).to(DEV), 'relu'))
    last = u
layers[-1] = (layers[-1][0], layers[-1][1], 'None')
chain = ops.MlpChain(layers, DEV)
x = torch.randn(rows, in_dim, generator=g).to(DEV)
heads = torch.empty(rows, 1 + A, device=DEV)
acts = [torch.empty(rows, u, device=DEV) for u in units]
chain.forward(x, heads, act_out=acts, groups=0)
d_heads = torch.randn(rows, 1 + A, generator=g).to(DEV)
for rep in range(3):
    dzs = [torch.full((rows, u), float('nan'), device=DEV) for u in units]
    chain.backward(d_heads, acts, dzs, None, groups=0)
    torch.cuda.synchronize()
    dgrad = d_heads.double()
    for l in range(len(units), 0, -1):
        h = acts[l - 1].double()
        ref = (dgrad @ layers[l][0].double()) * (h > 0).double()
        got = dzs[l - 1].double()
        bad = (got - ref).abs() > 1e-4 * max(1e-3, ref.abs().max().item())
        print(f'rep {rep} dz{l - 1}: {int(bad.sum())} wrong of {bad.numel()}')
        if bad.any():
            r, c = bad.nonzero(as_tuple=True)
            print('   wrong rows (mod 16):', sorted(set((r % 16).tolist())), ' row blocks:', sorted(set((r // 16).tolist()))[:12])
            print('   wrong cols (mod 16):', sorted(set((c % 16).tolist())), ' col blocks:', sorted(set((c // 16).tolist())))
            k = 0
            print('   sample got/ref:', [(int(r[i]), int(c[i]), round(got[r[i], c[i]].item(), 4), round(ref[r[i], c[i]].item(), 4)) for i in range(0, min(len(r), 40), 8)])
        dgrad = got
