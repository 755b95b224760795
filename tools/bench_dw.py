"""Library (tuned) timings of the manual engine's GEMMs at the 32,768-row minibatch, per shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_games_amd import gemm_tuning
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

tot = {'fwd': 0, 'dX': 0, 'dW': 0}
for (No, Mi) in [(400, 108), (200, 400), (100, 200), (22, 100)]:
    X = torch.randn(rows, Mi, device=dev); dY = torch.randn(rows, No, device=dev)
    W = torch.randn(No, Mi, device=dev); b = torch.randn(No, device=dev)
    G = torch.empty(No, Mi, device=dev); Y = torch.empty(rows, No, device=dev); dX = torch.empty(rows, Mi, device=dev)
    fl = 2.0 * rows * No * Mi
    t_f = timeit(lambda: torch.addmm(b, X, W.t(), out=Y))
    t_x = timeit(lambda: torch.mm(dY, W, out=dX))
    t_w = timeit(lambda: torch.mm(dY.t(), X, out=G))
    tot['fwd'] += t_f; tot['dX'] += t_x; tot['dW'] += t_w
    print(f'[{No:3d}x{Mi:3d}] rows {rows}: fwd {t_f:6.1f} us ({fl/t_f/1e6:5.1f} TF)  dX {t_x:6.1f} us ({fl/t_x/1e6:5.1f} TF)  '
          f'dW {t_w:6.1f} us ({fl/t_w/1e6:5.1f} TF)')
print('totals us:', {k: round(v, 1) for k, v in tot.items()})
