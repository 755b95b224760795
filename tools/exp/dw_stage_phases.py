"""Phase stamps of the staged weight-gradient kernel (csrc/mlp_dw.hip built with -DRLG_DW_STAMPS=<workgroup>
-DRLG_DW_STAMPS_WAVE=<wave>; tools/build_variant.sh): shader-clock cycles per chunk and phase of one wave.

    RLG_HIP_LIB=tools/exp/_build/dwstamps.so python tools/exp/dw_stage_phases.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rl_games_amd import _lib, ops  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    rows = 32768
    g = torch.Generator().manual_seed(7)
    shapes = [(400, 108), (200, 400), (100, 200), (22, 100)]
    jobs = [(torch.randn(rows, No, generator=g).to(dev), torch.randn(rows, Mi, generator=g).to(dev),
             torch.empty(No, Mi, device=dev)) for No, Mi in shapes]
    plan = ops.MlpDwPlan(shapes, rows, dev)
    print('plans', [plan.plan(k) for k in range(len(shapes))])
    lib = ctypes.CDLL(_lib.LIB_PATH if not os.environ.get('RLG_HIP_LIB') else os.environ['RLG_HIP_LIB'])
    out = (ctypes.c_longlong * 8)()
    for _ in range(3):
        plan.launch(jobs)
    torch.cuda.synchronize()
    lib.rlg_debug_dw_stamps(out, 1)
    reps = 10
    for _ in range(reps):
        plan.launch(jobs)
    torch.cuda.synchronize()
    lib.rlg_debug_dw_stamps(out, 1)
    chunks = out[0] / reps
    names = ['', 'fragment requests + first wait', 'four units (48 MFMAs + split)', 'plane stores + look-ahead issue', 'barrier']
    total = sum(out[k] for k in range(1, 5)) / reps
    print(f'chunks per launch {chunks:.0f}; cycles per chunk {total / chunks:.0f}')
    for k in range(1, 5):
        print(f'  {names[k]:36s} {out[k] / reps / chunks:8.0f}')


if __name__ == '__main__':
    main()
