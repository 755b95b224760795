"""Numerics of split-bf16 products - CPU study, no GPU needed.

x = x0 + x1 + x2 with each plane a bf16 value (8 significant bits each, 24 in total), the same for y.
A product x*y is then the sum of 9 plane products, each EXACT in fp32 (8 x 8 bits); summed into an fp32
accumulator by v_mfma_f32_16x16x32_bf16.  Keeping the 6 products of weight >= 2^-16 ("bf16x6") leaves a
truncation error of about 3 * 2^-24 |x||y| - the same order as one fp32 rounding.  This script measures the
error of whole layer products of config #3 against an fp64 result, next to the plain fp32 product."""
import sys
import torch

torch.manual_seed(0)


def planes(t, n=3):
    out, r = [], t.clone()
    for _ in range(n):
        p = r.bfloat16().float()
        out.append(p)
        r = r - p
    return out


def split_matmul(a, b, keep):
    """keep: list of (i, j) plane pairs, summed smallest first the way an accumulator chain would not -
    the order is irrelevant at this level (every partial is an fp32 matmul)."""
    pa, pb = planes(a), planes(b)
    acc = torch.zeros(a.shape[0], b.shape[1])
    for i, j in sorted(keep, key=lambda ij: -(ij[0] + ij[1])):
        acc += pa[i] @ pb[j]
    return acc


SETS = {
    'bf16x3': [(0, 0), (0, 1), (1, 0)],
    'bf16x6': [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)],
    'bf16x9': [(i, j) for i in range(3) for j in range(3)],
}


def report(name, rows, k, n, scale_a=1.0, scale_b=None):
    a = torch.randn(rows, k) * scale_a
    b = torch.randn(k, n) * (scale_b if scale_b is not None else (2.0 / k) ** 0.5)
    ref = a.double() @ b.double()
    denom = (a.double().abs() @ b.double().abs())        # the error scale of any summation of the products
    line = f'{name:>22} K={k:<6}'
    res = {'fp32': a @ b}
    for key, keep in SETS.items():
        res[key] = split_matmul(a, b, keep)
    for key, val in res.items():
        err = (val.double() - ref).abs()
        line += f' | {key} max {float((err / denom).max()):.2e} rms {float((err / denom).pow(2).mean().sqrt()):.2e}'
    print(line)


if __name__ == '__main__':
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    print('error / (|a| @ |b|), against fp64')
    report('forward 108->400', rows, 108, 400)
    report('forward 400->200', rows, 400, 200)
    report('forward 200->100', rows, 200, 100)
    report('forward 100->22', rows, 100, 22)
    report('dW 400x108 (K=rows)', 400, 32768, 108, scale_a=1e-3, scale_b=1.0)
    report('dW 200x400 (K=rows)', 200, 32768, 400, scale_a=1e-3, scale_b=1.0)
