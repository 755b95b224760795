"""LDS-free MFMA forward (Linear+bias+ELU) and dX (+act') kernels vs the tuned library + elementwise."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_games_amd import gemm_tuning, ops
gemm_tuning.enable()
dev = 'cuda:0'
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 32768

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
    x = torch.randn(rows, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    z = torch.empty(rows, N, device=dev); h = torch.empty(rows, N, device=dev)
    fl = 2.0 * rows * N * K
    if act:
        t_lib = timeit(lambda: (torch.addmm(b, x, w.t(), out=z), torch.ops.aten.elu.out(z, out=h)))
    else:
        t_lib = timeit(lambda: torch.addmm(b, x, w.t(), out=z))
    t_mine = timeit(lambda: ops.mlp_linear_act_forward(x, w, b, h, pre_act=z if act else None, act_kind=act))
    tot[0] += t_lib; tot[1] += t_mine
    line = f'[{N:3d}x{K:3d}] fwd lib(+elu) {t_lib:6.1f} us  mfma fused {t_mine:6.1f} us ({fl/t_mine/1e6:5.1f} TF)'
    if N % 4 == 0 and K != 108:
        dz = torch.randn(rows, N, device=dev); zp = torch.randn(rows, K, device=dev); dp = torch.empty(rows, K, device=dev)
        nb = ops.act_bwd_blocks(rows, K); part = torch.empty(nb * K, dtype=torch.float64, device=dev)
        t_lib2 = timeit(lambda: (torch.mm(dz, w, out=dp), ops.act_bwd_colsum(dp, zp, dp, 1, part, nb)))
        t_mine2 = timeit(lambda: ops.mlp_linear_act_backward(dz, w, zp, dp, act_kind=1))
        tot[2] += t_lib2; tot[3] += t_mine2
        line += f' | dX lib+act_bwd_colsum {t_lib2:6.1f} us  mfma fused {t_mine2:6.1f} us ({fl/t_mine2/1e6:5.1f} TF)'
    print(line)
print('totals us: fwd lib %.1f mfma %.1f | dX lib %.1f mfma %.1f' % tuple(tot))
