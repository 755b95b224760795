"""CPU emulation (numpy): products of fp32 operands as (a) six exact bf16 plane products (three planes per operand, the
shipped split), (b) three exact fp16 plane products (two planes per operand, operands pre-scaled by a power of two so that
the largest sits near 2^13), both accumulated in fp32 per 32-wide chunk the way the MFMA does (chunk sums formed wide, one
fp32 rounding per chunk and accumulate), against fp64 and against plain fp32 products accumulated the same way.

    python tools/exp/split_f16_numerics.py
"""
import numpy as np


def bf16_round(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32)


def split_bf16(x):
    p0 = bf16_round(x)
    r = (x - p0).astype(np.float32)
    p1 = bf16_round(r)
    r = (r - p1).astype(np.float32)
    p2 = bf16_round(r)
    return p0, p1, p2


def split_f16(x, scale):
    xs = (x * np.float32(scale)).astype(np.float32)
    p0 = xs.astype(np.float16)
    assert np.isfinite(p0).all(), 'fp16 overflow'
    r = (xs - p0.astype(np.float32)).astype(np.float32)
    p1 = r.astype(np.float16)
    return p0.astype(np.float32), p1.astype(np.float32)


def chunked_dot(pairs, K, chunk=32):
    """pairs: list of (A [M,K], B [K,N]) whose products are summed; one fp32 rounding per chunk and pair, in order."""
    acc = None
    for c in range(0, K, chunk):
        for A, B in pairs:
            part = (A[:, c:c + chunk].astype(np.float64) @ B[c:c + chunk].astype(np.float64))
            acc = part.astype(np.float32) if acc is None else (acc.astype(np.float64) + part).astype(np.float32)
    return acc


def pow2_scale(x, target=2.0 ** 13):
    m = float(np.abs(x).max())
    return 2.0 ** np.floor(np.log2(target / m)) if m > 0 else 1.0


def run(name, x, w):
    M, K = x.shape
    truth = x.astype(np.float64) @ w.astype(np.float64)
    norm = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64)
    exact = chunked_dot([(x, w)], K)
    x0, x1, x2 = split_bf16(x)
    w0, w1, w2 = split_bf16(w)
    b6 = chunked_dot([(x0, w0), (x0, w1), (x1, w0), (x1, w1), (x0, w2), (x2, w0)], K)
    sx, sw = pow2_scale(x), pow2_scale(w)
    a0, a1 = split_f16(x, sx)
    b0, b1 = split_f16(w, sw)
    f3 = chunked_dot([(a0, b0), (a0, b1), (a1, b0)], K) * np.float32(1.0 / (sx * sw))
    f4 = chunked_dot([(a0, b0), (a0, b1), (a1, b0), (a1, b1)], K) * np.float32(1.0 / (sx * sw))
    print(f'{name}: scales 2^{int(np.log2(sx))} 2^{int(np.log2(sw))}')
    for label, got in (('exact fp32 products', exact), ('bf16 x 6', b6), ('fp16 x 3', f3), ('fp16 x 4', f4)):
        e = np.abs(got.astype(np.float64) - truth)
        print(f'   {label:22s} max err / (|x|.|w|) {np.max(e / norm):.3e}   rms {np.sqrt(np.mean((e / norm) ** 2)):.3e}'
              f'   max err / max|y| {e.max() / np.abs(truth).max():.3e}')


def main():
    rng = np.random.default_rng(0)
    M, K, N = 256, 400, 200
    pre = rng.standard_normal((M, K)).astype(np.float32)
    x = np.where(pre > 0, pre, np.expm1(pre)).astype(np.float32)            # ELU outputs
    w = (rng.standard_normal((K, N)) * 0.05).astype(np.float32)
    run('activations x weights (400 -> 200)', x, w)
    # a gradient-like operand: wide dynamic range (rows scaled log-uniformly over 2^-12 .. 1), tiny overall
    dz = (rng.standard_normal((M, K)) * np.exp2(rng.uniform(-12, 0, (M, 1))) * 3e-5).astype(np.float32)
    run('dZ (wide range, ~1e-5) x weights', dz, w)
    # weight-gradient shape: contraction over the rows
    h = x[:, :200]
    run('dZ^T x H (contraction over 256 rows)', np.ascontiguousarray(dz[:, :100].T), h)
    # heavy-tailed operand: a few elements 1e4 times the rest (scale set by the outliers)
    y = x.copy()
    y[rng.integers(0, M, 20), rng.integers(0, K, 20)] *= 1e4
    run('activations with 20 outliers x 1e4', y, w)


if __name__ == '__main__':
    main()
