"""CPU: the arithmetic of the split-product launches restated with torch - the bf16 form (csrc/split_bf16.hpp: three exact
bf16 planes per operand, six plane products per fp32 product) and the fp16 form (csrc/split_f16.hpp, round 6: two fp16
planes of the operand times a power of two, three plane products) - both with fp32 accumulation: the truncated sums are as
accurate against fp64 as a plain fp32 product.  (The kernels themselves are checked against fp64 and the library on the
GPU: tests/test_ops_gpu.py, tests/test_mlp_chain_gpu.py, tools/exp/dw_bf16_check.py.)"""
import pytest
import torch


def _planes(t):
    """x -> (x0, x1, x2): bf16 values (as fp32) by round-to-nearest-even of the running residual."""
    out, r = [], t.clone()
    for _ in range(3):
        p = r.bfloat16().float()
        out.append(p)
        r = r - p                      # exact in fp32: the residual has <= 16 (8) significant bits
    return out, r


KEPT = [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]      # order of the kernel: small terms first


def _split_matmul(a, b):
    (pa, _), (pb, _) = _planes(a), _planes(b)
    acc = torch.zeros(a.shape[0], b.shape[1])
    for i, j in KEPT:
        acc += pa[i] @ pb[j]           # products of bf16 values are exact in fp32; fp32 accumulation
    return acc


@pytest.mark.parametrize('scale', [1.0, 1e-4, 3e3])
def test_three_bf16_planes_reproduce_an_fp32_value_exactly(scale):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(100000, generator=g) * scale * torch.exp(3.0 * torch.randn(100000, generator=g))
    x[:4] = torch.tensor([0.0, -0.0, 1.0, -3.0e-20])
    planes, rest = _planes(x)
    assert torch.equal(planes[0] + planes[1] + planes[2], x)
    assert torch.count_nonzero(rest) == 0
    # every plane is a bf16 value: 8 significant bits, the low 16 bits of its fp32 pattern are zero
    for p in planes:
        assert torch.all((p.view(torch.int32) & 0xFFFF) == 0)
    # each plane is at most half an ulp (2^-8 relative) of the one above it
    assert torch.all(planes[1].abs() <= planes[0].abs() * 2.0 ** -8 + 1e-45)
    assert torch.all(planes[2].abs() <= planes[1].abs() * 2.0 ** -8 + 1e-45)


@pytest.mark.parametrize('rows,No,Mi', [(4096, 200, 400), (4096, 22, 100), (512, 64, 60)])
def test_six_plane_products_are_as_accurate_as_an_fp32_product(rows, No, Mi):
    """dW = dZ^T X on gradient-like / activation-like operands: error against fp64 relative to |dZ|^T |X|.
    The dropped terms (x1 y2, x2 y1, x2 y2) are <= 3 * 2^-24 |x||y| per product - one fp32 rounding."""
    g = torch.Generator().manual_seed(rows + No)
    dz = torch.randn(rows, No, generator=g) * torch.exp(2.0 * torch.randn(rows, 1, generator=g)) * 1e-4
    x = torch.nn.functional.elu(torch.randn(rows, Mi, generator=g))
    ref = dz.double().t() @ x.double()
    scale = dz.double().abs().t() @ x.double().abs()
    err_split = ((_split_matmul(dz.t().contiguous(), x).double() - ref).abs() / scale)
    err_fp32 = (((dz.t() @ x).double() - ref).abs() / scale)
    # (absolute level: the fp32 ACCUMULATION of a few thousand heavy-tailed rows, common to both forms)
    assert err_split.max() < 2e-6 and err_split.pow(2).mean().sqrt() < 3e-7
    assert err_split.max() <= 1.5 * err_fp32.max() + 1e-8
    assert err_split.pow(2).mean().sqrt() <= 1.5 * err_fp32.pow(2).mean().sqrt() + 1e-9
    # three products only (x0 y0, x0 y1, x1 y0) would NOT do: an order of magnitude worse
    (pa, _), (pb, _) = _planes(dz.t().contiguous()), _planes(x)
    three = pa[0] @ pb[0] + pa[0] @ pb[1] + pa[1] @ pb[0]
    assert ((three.double() - ref).abs() / scale).pow(2).mean().sqrt() > 5 * err_split.pow(2).mean().sqrt()


# ------------------------------------------------------------------ the fp16 form (csrc/split_f16.hpp)

def _f16_scale(t, target=13):
    """power of two that puts the largest magnitude into [2^13, 2^14) (f16_scale_for)"""
    m = float(t.abs().max())
    return 1.0 if m == 0.0 else 2.0 ** (target - int(torch.floor(torch.log2(torch.tensor(m))).item()))


def _planes_f16(t, scale):
    xs = t * scale                                  # exact: a power of two
    h0 = xs.half().float()
    assert torch.isfinite(h0).all()
    h1 = (xs - h0).half().float()                   # the residual is exact in fp32; fp16 keeps 11 of its <= 13 bits
    return h0, h1, xs - h0 - h1


@pytest.mark.parametrize('scale', [1.0, 1e-4, 3e3])
def test_two_fp16_planes_carry_an_fp32_value_to_22_bits(scale):
    """x S = h0 + h1 + e with |e| <= 2^-22 |x S| in the worst case (two planes of 11 significant bits; 2^-23 and less for
    most values) down to the fp16 normal range, |e| <= 2^-25 absolute below it (gradual underflow: 2^-38 of the largest
    element) - the representation error of the fp16 form."""
    g = torch.Generator().manual_seed(4)
    x = torch.randn(100000, generator=g) * scale * torch.exp(1.5 * torch.randn(100000, generator=g))
    S = _f16_scale(x)
    assert 2.0 ** 13 <= float(x.abs().max()) * S < 2.0 ** 14
    h0, h1, e = _planes_f16(x, S)
    xs = (x * S).abs()
    assert torch.all(e.abs() <= torch.maximum(xs * 2.0 ** -22, torch.tensor(2.0 ** -25)))
    assert (e.abs() / xs.clamp_min(1.0)).pow(2).mean().sqrt() < 2.0 ** -24          # rms: below one fp32 rounding
    assert torch.all(h1.abs() <= h0.abs() * 2.0 ** -11 + 2.0 ** -24)


@pytest.mark.parametrize('rows,No,Mi', [(4096, 200, 400), (4096, 22, 100), (512, 64, 60)])
def test_three_fp16_plane_products_are_as_accurate_as_an_fp32_product(rows, No, Mi):
    """The same operands as the bf16 test: h0 g0 + h0 g1 + h1 g0 (each exact in fp32), un-scaled behind the sum.  The
    dropped h1 g1 and the plane rounding are <= 3 * 2^-22 |x||y| per product in the worst case, ~2^-24 rms - below the
    roundings of the fp32 accumulation, which is why the sums come out as accurate as with the bf16 form or plain fp32."""
    g = torch.Generator().manual_seed(rows + No)
    dz = torch.randn(rows, No, generator=g) * torch.exp(2.0 * torch.randn(rows, 1, generator=g)) * 1e-4
    x = torch.nn.functional.elu(torch.randn(rows, Mi, generator=g))
    ref = dz.double().t() @ x.double()
    scale = dz.double().abs().t() @ x.double().abs()
    sa, sb = _f16_scale(dz), _f16_scale(x)
    a0, a1, _ = _planes_f16(dz.t().contiguous(), sa)
    b0, b1, _ = _planes_f16(x, sb)
    acc = torch.zeros(No, Mi)
    for pa, pb in ((a1, b0), (a0, b1), (a0, b0)):           # the kernel's order: small terms first
        acc += pa @ pb
    got = acc * (1.0 / (sa * sb))
    err = ((got.double() - ref).abs() / scale)
    err_fp32 = (((dz.t() @ x).double() - ref).abs() / scale)
    err_bf16 = ((_split_matmul(dz.t().contiguous(), x).double() - ref).abs() / scale)
    assert err.max() < 2e-6 and err.pow(2).mean().sqrt() < 3e-7
    assert err.max() <= 1.5 * err_fp32.max() + 1e-8
    assert err.pow(2).mean().sqrt() <= 1.5 * err_fp32.pow(2).mean().sqrt() + 1e-9
    assert err.pow(2).mean().sqrt() <= 1.5 * err_bf16.pow(2).mean().sqrt() + 1e-9
    # the top planes alone (one product) are three orders of magnitude worse
    one = (a0 @ b0) * (1.0 / (sa * sb))
    assert ((one.double() - ref).abs() / scale).pow(2).mean().sqrt() > 100 * err.pow(2).mean().sqrt()


# ------------------------------------------------------------------ host side of the split-product chain (no GPU needed)

def _arr(vals):
    import ctypes
    return (ctypes.c_int * len(vals))(*vals)


def test_chain_plane_buffer_sizes_and_envelope_of_the_split_kernels():
    """rlg_mlp_chain_planes_bytes / _offset / _bx_supported are host logic of the C ABI: fragment counts (16-row blocks
    x 32-column chunks x the planes of 1 KiB - two fp16 planes, or three bf16 planes in a -DRLG_BX_F16=0 build; the
    backward needs none for layer 0), the combined buffer of both
    directions, and the envelope - minibatches of >= 16,384 rows; the forward's LDS plan must fit (a windowed
    400- or 512-wide tile is fine, a windowed tile whose consumer has more than 256 outputs is not)."""
    from rl_games_amd import _lib
    lib = _lib.load()
    ins, outs = [108, 400, 200, 100], [400, 200, 100, 22]
    n = len(ins)
    blocks = lambda v: -(-v // 16)
    chunks = lambda v: -(-v // 32)
    products = lib.rlg_mlp_chain_split_products()
    assert products in (3, 6)
    chunk = 1024 * {3: 2, 6: 3}[products]
    fwd = sum(blocks(o) * chunks(i) * chunk for i, o in zip(ins, outs))
    bwd = sum(blocks(i) * chunks(o) * chunk for i, o in list(zip(ins, outs))[1:])
    assert lib.rlg_mlp_chain_planes_bytes(n, _arr(ins), _arr(outs), 0) == fwd
    assert lib.rlg_mlp_chain_planes_bytes(n, _arr(ins), _arr(outs), 1) == bwd
    off = lib.rlg_mlp_chain_planes_offset(n, _arr(ins), _arr(outs), 1)
    assert off >= fwd and off % 256 == 0 and lib.rlg_mlp_chain_planes_offset(n, _arr(ins), _arr(outs), 0) == 0
    assert lib.rlg_mlp_chain_planes_bytes(n, _arr(ins), _arr(outs), 2) == off + bwd
    assert lib.rlg_mlp_chain_planes_bytes(n, _arr(ins), _arr(outs), 3) == -1

    def supported(i, o, rows, direction, groups=0):
        return lib.rlg_mlp_chain_bx_supported(len(i), _arr(i), _arr(o), rows, groups, direction)
    for direction in (0, 1):
        assert supported(ins, outs, 32768, direction) == 1            # the benchmarked update shape
        assert supported(ins, outs, 16384, direction) == 1
        assert supported(ins, outs, 4096, direction) == 0             # a data-parallel rank's minibatch: exact products
        assert supported([60, 256, 128, 64], [256, 128, 64, 9], 32768, direction) == 1
    assert supported(ins, outs, 65536, 0) == 1                        # the rollout forward
    assert supported([33, 512, 64], [512, 64, 8], 16384, 0) == 1      # windowed 512-wide tile, 4 consumer blocks
    assert supported([33, 1024, 512], [1024, 512, 8], 16384, 0) == 0  # its consumer would need 32 blocks in registers
    assert supported([108], [22], 32768, 1) == 0                      # a single layer has no dX chain
