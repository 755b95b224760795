"""Accuracy and time of the weight-gradient launch under RLG_DW_BF16 (experiment: split-bf16 products,
DESIGN.md section 9).  The mode is read once by the library, so run one process per mode:

    RLG_DW_BF16=13 python tools/exp/dw_bf16_check.py [--reps 50]

Prints, per layer of the BASELINE MLP at 32,768 rows, the error against an fp64 product relative to
|dZ|^T |X| (max and rms) beside the library's fp32 result, then the time of the launch pair."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rl_games_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=32768)
    ap.add_argument('--reps', type=int, default=50)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    mode = os.environ.get('RLG_DW_BF16', '0')
    g = torch.Generator().manual_seed(7)
    shapes = [(200, 400), (400, 108), (100, 200), (22, 100)]
    jobs = []
    for No, Mi in shapes:
        # gradient-like left operand (small, wide dynamic range), activation-like right operand
        dz = (torch.randn(args.rows, No, generator=g) * torch.exp(2.0 * torch.randn(args.rows, 1, generator=g)) * 1e-4).to(dev)
        x = torch.nn.functional.elu(torch.randn(args.rows, Mi, generator=g)).to(dev)
        jobs.append((dz, x, torch.empty(No, Mi, device=dev)))
    # (RLG_DW_F16=1: the launch scales its operands by powers of two taken from these bounds)
    os.environ.setdefault('RLG_DW_F16_AMAX_DZ', repr(max(j[0].abs().max().item() for j in jobs)))
    os.environ.setdefault('RLG_DW_F16_AMAX_X', repr(max(j[1].abs().max().item() for j in jobs)))
    plan = ops.MlpDwPlan(shapes, args.rows, dev)
    plan.launch(jobs)
    torch.cuda.synchronize()
    print(f'RLG_DW_BF16={mode} RLG_DW_F16={os.environ.get("RLG_DW_F16", "0")} rows {args.rows}')
    for dz, x, grad in jobs:
        t64 = dz.double().t() @ x.double()
        scale = dz.double().abs().t() @ x.double().abs()
        lib = dz.t() @ x
        e = ((grad.double() - t64).abs() / scale)
        el = ((lib.double() - t64).abs() / scale)
        rel = ((grad.double() - t64).abs().max() / t64.abs().max()).item()
        print(f'  {tuple(grad.shape)!s:>12}: kernel max {e.max().item():.2e} rms {e.pow(2).mean().sqrt().item():.2e} | '
              f'library max {el.max().item():.2e} rms {el.pow(2).mean().sqrt().item():.2e} | max err / max |G| {rel:.2e}')
    for _ in range(5):
        plan.launch(jobs)
    torch.cuda.synchronize()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = []
    for _ in range(3):
        start.record()
        for _ in range(args.reps):
            plan.launch(jobs)
        stop.record()
        torch.cuda.synchronize()
        best.append(start.elapsed_time(stop) * 1e3 / args.reps)
    macs = sum(a * b for a, b in shapes)
    t = min(best)
    print(f'  dW + finalise: {t:.1f} us  ({2e-6 * args.rows * macs / t:.1f} fp32-equivalent TFLOP/s); runs {["%.1f" % b for b in best]}')


if __name__ == '__main__':
    main()
