"""Correctness + timing of the fp32-MFMA MLP layer kernels against torch (rocBLAS/hipBLASLt)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_games_amd import ops, gemm_tuning
gemm_tuning.enable()
dev = 'cuda:0'

def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

shapes = [(32768, 400, 108), (32768, 200, 400), (32768, 100, 200), (32768, 22, 100), (65536, 400, 108),
          (4096, 400, 108), (4096, 200, 400), (1000, 100, 37), (300, 64, 3)]
for M, N, K in shapes:
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(dev); w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    z = torch.empty(M, N, device=dev); h = torch.empty(M, N, device=dev)
    ops.mlp_forward_layer(x, w, b, h, pre_act=z, act_kind=1)
    z_ref = torch.addmm(b, x, w.t()); h_ref = torch.nn.functional.elu(z_ref)
    z64 = torch.addmm(b.double(), x.double(), w.double().t())
    err = (z.double() - z64).abs().max().item(); err_ref = (z_ref.double() - z64).abs().max().item()
    ok = torch.allclose(h, h_ref, rtol=1e-4, atol=1e-5)
    zr = torch.empty_like(z_ref); hr = torch.empty_like(z_ref)
    t_ref = timeit(lambda: (torch.addmm(b, x, w.t(), out=zr), torch.ops.aten.elu.out(zr, out=hr)))
    t_mine = timeit(lambda: ops.mlp_forward_layer(x, w, b, h, pre_act=z, act_kind=1))
    fl = 2.0 * M * N * K
    print(f'fwd M={M:6d} N={N:4d} K={K:4d}  ok={ok}  err vs f64: mine {err:.2e} lib {err_ref:.2e}   '
          f'lib+elu {t_ref:7.1f} us ({fl/t_ref/1e6:6.1f} TF)   fused {t_mine:7.1f} us ({fl/t_mine/1e6:6.1f} TF)')
