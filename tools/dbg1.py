import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_games_amd import ops
g = torch.Generator().manual_seed(0)
rows, C = 4096, 108
mean = (torch.randn(C, generator=g, dtype=torch.float64) * 2)
var = (torch.rand(C, generator=g, dtype=torch.float64) * 10 + 0.1)
x = torch.randn(rows, C, generator=g) * 4
m32 = mean.float(); 
d_ref = torch.sqrt(var.float() + 1e-5)
# exact: sqrt in f64 of the f32 sum
d_exact = torch.sqrt((var.float() + np.float32(1e-5)).double()).float()
print('cpu sqrt correctly rounded:', torch.equal(d_ref, d_exact))
num = (x - m32)
q_ref = num / d_ref
q_exact = (num.double() / d_ref.double()).float()
print('cpu div correctly rounded:', torch.equal(q_ref, q_exact), (q_ref != q_exact).sum().item())
y = ops.rms_apply(x.cuda(), mean.cuda(), var.cuda(), 1e-5, 2).cpu()   # norm_only: x / d
y_exact = (x.double() / d_ref.double()).float()
print('kernel x/d vs exact mismatches:', (y != y_exact).sum().item())
y0 = ops.rms_apply(x.cuda(), mean.cuda(), var.cuda(), 1e-5, 0).cpu()
y0_exact = torch.clamp(q_exact, -5, 5)
print('kernel mode0 vs exact mismatches:', (y0 != y0_exact).sum().item())
num_gpu = (x.cuda() - m32.cuda()).cpu()
print('sub equal', torch.equal(num_gpu, num))
