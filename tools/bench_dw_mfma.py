"""MFMA weight-gradient launch (all 4 layers) vs the tuned library GEMMs, 32,768-row minibatch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_games_amd import gemm_tuning, ops
gemm_tuning.enable()
dev = 'cuda:0'
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
targets = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [256, 512, 1024]

def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

shapes = [(400, 108), (200, 400), (100, 200), (22, 100)]
layers = [(torch.randn(rows, No, device=dev), torch.randn(rows, Mi, device=dev), torch.empty(No, Mi, device=dev))
          for No, Mi in shapes]
fl = sum(2.0 * rows * No * Mi for No, Mi in shapes)
t_lib = timeit(lambda: [torch.mm(dz.t(), x, out=g) for dz, x, g in layers])
print(f'library (tuned) 4 GEMMs: {t_lib:7.1f} us ({fl/t_lib/1e6:5.1f} TF)')
for tb in targets:
    plan = ops.MlpDwPlan(shapes, rows, dev, target_blocks=tb)
    t = timeit(lambda: plan.launch(layers))
    print(f'MFMA one launch, target_blocks {tb:5d}: {t:7.1f} us ({fl/t/1e6:5.1f} TF)  plans {[plan.plan(k) for k in range(4)]}')
    for k, (No, Mi) in enumerate(shapes):
        p1 = ops.MlpDwPlan([shapes[k]], rows, dev, target_blocks=tb)
        t1 = timeit(lambda: p1.launch([layers[k]]))
        print(f'    [{No}x{Mi}] alone {t1:6.1f} us ({2.0*rows*No*Mi/t1/1e6:5.1f} TF) plan {p1.plan(0)}')
