"""Feature-major MFMA forward (+bias+ELU, both layouts out) and dX (+act', both layouts) vs library."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_games_amd import gemm_tuning, ops
gemm_tuning.enable()
dev = 'cuda:0'
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768

def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

tot = [0, 0, 0, 0]
for (N, K, act) in [(400, 108, 1), (200, 400, 1), (100, 200, 1), (22, 100, 0)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    z = torch.empty(M, N, device=dev); h = torch.empty(M, N, device=dev)
    xt = x.t().contiguous(); wt = w.t().contiguous()
    zt = torch.empty(N, M, device=dev); ht = torch.empty(N, M, device=dev); hsm = torch.empty(M, N, device=dev)
    fl = 2.0 * M * N * K
    if act:
        t_lib = timeit(lambda: (torch.addmm(b, x, w.t(), out=z), torch.ops.aten.elu.out(z, out=h)))
    else:
        t_lib = timeit(lambda: torch.addmm(b, x, w.t(), out=z))
    t_mine = timeit(lambda: ops.mlp_fm_forward(wt, xt, b, zt if act else None, ht, hsm, act_kind=act))
    tot[0] += t_lib; tot[1] += t_mine
    line = f'[{N:3d}x{K:3d}] fwd lib(+elu) {t_lib:6.1f} us  fm-mfma (Z^T,H^T,H) {t_mine:6.1f} us ({fl/t_mine/1e6:5.1f} TF)'
    if K != 108:
        dz = torch.randn(M, N, device=dev); dzt = dz.t().contiguous()
        zp = torch.randn(M, K, device=dev); zpt = zp.t().contiguous()
        dp = torch.empty(M, K, device=dev); dpt = torch.empty(K, M, device=dev)
        nb = ops.act_bwd_blocks(M, K); part = torch.empty(nb * K, dtype=torch.float64, device=dev)
        bg = torch.empty(K, device=dev)
        t_lib2 = timeit(lambda: (torch.mm(dz, w, out=dp), ops.act_bwd_colsum(dp, zp, dp, 1, part, nb),
                                 ops.colsum_finalize(part, nb, K, bg)))
        t_mine2 = timeit(lambda: (ops.mlp_fm_backward(w, dzt, zpt, dpt, dp, act_kind=1), ops.fm_row_sum(dpt, bg)))
        tot[2] += t_lib2; tot[3] += t_mine2
        line += f' | dX lib+act_bwd+colsum {t_lib2:6.1f} us  fm-mfma(+rowsum) {t_mine2:6.1f} us ({fl/t_mine2/1e6:5.1f} TF)'
    print(line)
print('totals us: fwd lib %.1f fm %.1f | dX lib %.1f fm %.1f' % tuple(tot))
