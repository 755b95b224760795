"""Phase times of the LDS-staged weight-gradient kernel (a -DRLG_DW_STAMPS=<workgroup> build, tools/exp/build_dw_stamps.sh):
s_memtime ticks (~ shader cycles on gfx950) per chunk spent waiting at barrier 1, splitting + storing,
issuing + waiting at barrier 2, and in the fragment reads + MFMAs - for one wave of one workgroup."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rl_games_amd import _lib, ops  # noqa: E402

dev = torch.device('cuda:0')
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
g = torch.Generator().manual_seed(7)
shapes = [(200, 400), (400, 108), (100, 200), (22, 100)]
jobs = []
for No, Mi in shapes:
    dz = (torch.randn(rows, No, generator=g) * 1e-4).to(dev)
    x = torch.nn.functional.elu(torch.randn(rows, Mi, generator=g)).to(dev)
    jobs.append((dz, x, torch.empty(No, Mi, device=dev)))
plan = ops.MlpDwPlan(shapes, rows, dev)
lib = _lib.load()
fn = lib.rlg_debug_dw_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
out = (ctypes.c_longlong * 8)()
for _ in range(3):
    plan.launch(jobs)
torch.cuda.synchronize()
fn(out, 1)
reps = 20
for _ in range(reps):
    plan.launch(jobs)
torch.cuda.synchronize()
fn(out, 1)
v = list(out)
chunks = max(v[0], 1)
names = ['barrier-1 wait', 'split + LDS stores', 'issue + barrier-2 wait', 'fragment reads + MFMAs']
tick = float(os.environ.get('TICK_CYCLES', '1'))        # gfx950: s_memtime ticks are ~ shader cycles (checked against the launch time)
print(f'chunks {chunks} over {reps} launches; per chunk (shader cycles at {tick} per tick):')
for k, n in enumerate(names, 1):
    print(f'  {n:28s} {v[k] / chunks * tick:8.0f}')
print(f'  {"sum":28s} {sum(v[1:5]) / chunks * tick:8.0f}')
