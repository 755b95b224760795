"""GPU parity tests of the vertically fused MLP kernels (csrc/mlp_chain.hip) and the 16x16x4 MFMA
weight-gradient launch (csrc/mlp_dw.hip), through the C ABI.

Reference = the same op chain the reference network runs (rl_games/algos_torch/network_builder.py:
447-512 actor_mlp + value/mu heads, norm_obs rl_games/algos_torch/models.py:54-56) evaluated in fp64
with torch; the kernels compute exact fp32 products with fp32 accumulation, so they must be as
accurate as the fp32 library path (tolerance: a small multiple of the fp32 library's own error
against fp64, floor 1e-6 of the tensor scale) - well inside the 1e-5 the losses are held to.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'
ACT = {'elu': torch.nn.functional.elu, 'relu': torch.relu, 'tanh': torch.tanh, 'None': lambda t: t}


def _net(in_dim, units, out_dim, act, seed, packed=True):
    """packed: the parameters are views into ONE flat tensor, every weight matrix followed by its bias - the layout of
    the optimizer's arena (rl_games_amd/flat_optim.py), which the pipelined forward relies on (a row's last chunk may
    read up to 48 bytes past a matrix whose width is not a multiple of 16: chain_pipe_fill in csrc/mlp_chain.hip
    checks that this lands in another array of the network).  packed=False: separately allocated tensors - the
    launch must then fall back to the unit-structured kernels for such widths instead of reading foreign memory."""
    g = torch.Generator().manual_seed(seed)
    shapes, last = [], in_dim
    for u in list(units) + [out_dim]:
        shapes.append((u, last))
        last = u
    flat = torch.empty(sum(u * i + u for u, i in shapes), device=DEV) if packed else None
    layers, off = [], 0
    for u, i in shapes:
        w = torch.randn(u, i, generator=g) / i ** 0.5
        b = 0.1 * torch.randn(u, generator=g)
        if packed:
            wv, bv = flat[off:off + u * i].view(u, i), flat[off + u * i:off + u * i + u]
            wv.copy_(w)
            bv.copy_(b)
            off += u * i + u
            layers.append([wv, bv, act])
        else:
            layers.append([w.to(DEV), b.to(DEV), act])
    layers[-1][2] = 'None'
    return [tuple(l) for l in layers], g


def _ref_forward(layers, x, dtype):
    hs, a = [], x.to(dtype)
    for w, b, act in layers:
        a = ACT[act](torch.addmm(b.to(dtype), a, w.to(dtype).t()))
        hs.append(a)
    return hs


def _close(got, ref64, lib32=None, mult=4.0, floor=1e-6):
    err = (got.double() - ref64).abs().max().item()
    scale = max(ref64.abs().max().item(), 1e-30)
    bound = floor * scale
    if lib32 is not None:
        bound = max(bound, mult * (lib32.double() - ref64).abs().max().item())
    assert err <= bound, (err, bound, scale)


SHAPES = [
    (108, [400, 200, 100], 22, 'elu'),      # BASELINE config #3 / #4
    (60, [256, 128, 64], 9, 'elu'),         # BASELINE config #2
    (13, [20, 36], 5, 'tanh'),              # nothing a multiple of 16, input not a multiple of 4
    (3, [64, 64], 2, 'relu'),               # Pendulum-sized observations
    (48, [32], 7, 'None'),
    (12, [100, 52], 22, 'elu'),             # 7 output blocks of the first layer: all of them remainder blocks of an
                                            # 8-wave workgroup (the backward stages them in LDS: sized for 7, not 3)
]


@pytest.mark.parametrize('in_dim,units,out_dim,act', SHAPES)
@pytest.mark.parametrize('rows,groups,packed', [(1, 0, True), (63, 1, True), (64, 2, True), (1000, 4, True), (4113, 0, True),
                                                (16384, 4, True), (64, 2, False), (16384, 0, False), (20000, 4, False)])
def test_chain_forward_matches_fp64(in_dim, units, out_dim, act, rows, groups, packed):
    from rl_games_amd import ops
    layers, g = _net(in_dim, units, out_dim, act, seed=rows + in_dim, packed=packed)
    chain = ops.MlpChain(layers, DEV)
    x = (3 * torch.randn(rows, in_dim, generator=g) + 1).to(DEV)
    mean = torch.randn(in_dim, generator=g, dtype=torch.float64).to(DEV)
    var = (torch.rand(in_dim, generator=g, dtype=torch.float64) * 4 + 0.1).to(DEV)
    for rms in (None, (mean, var)):
        heads = torch.full((rows, out_dim), float('nan'), device=DEV)
        acts = [torch.full((rows, u), float('nan'), device=DEV) for u in units]
        xn = torch.full((rows, in_dim), float('nan'), device=DEV) if rms is not None else None
        chain.forward(x, heads, act_out=acts, rms=rms, eps=1e-5, xn_out=xn, groups=groups)
        if rms is not None:
            xn_ref = ops.rms_apply(x, mean, var, 1e-5, 0)           # the stand-alone normaliser kernel
            assert torch.equal(xn, xn_ref)
            xin = xn_ref
        else:
            xin = x
        ref64 = _ref_forward(layers, xin, torch.float64)
        lib32 = _ref_forward(layers, xin, torch.float32)
        for got, r64, r32 in zip(acts + [heads], ref64, lib32):
            assert torch.isfinite(got).all()
            _close(got, r64, r32)
        # inference form: nothing but the heads is written, same bits
        heads2 = torch.full((rows, out_dim), float('nan'), device=DEV)
        chain.forward(x, heads2, rms=rms, eps=1e-5, groups=groups)
        assert torch.equal(heads2, heads)


@pytest.mark.parametrize('in_dim,units,out_dim,act', SHAPES)
@pytest.mark.parametrize('rows,groups', [(1, 0), (63, 1), (64, 2), (1000, 4), (4113, 0), (16384, 4)])
def test_chain_backward_matches_autograd_fp64(in_dim, units, out_dim, act, rows, groups):
    from rl_games_amd import ops
    layers, g = _net(in_dim, units, out_dim, act, seed=7 * rows + in_dim)
    chain = ops.MlpChain(layers, DEV)
    x = torch.randn(rows, in_dim, generator=g).to(DEV)
    heads = torch.empty(rows, out_dim, device=DEV)
    acts = [torch.empty(rows, u, device=DEV) for u in units]
    chain.forward(x, heads, act_out=acts, groups=groups)
    d_heads = torch.randn(rows, out_dim, generator=g).to(DEV)
    dzs = [torch.full((rows, u), float('nan'), device=DEV) for u in units]
    nblk = chain.num_blocks(rows, 1, groups)
    parts = [torch.full((nblk * u,), float('nan'), dtype=torch.float64, device=DEV) for u in units]
    chain.backward(d_heads, acts, dzs, parts, groups=groups)

    def ref(dtype):
        # backward from the kernel's own activations (act' is taken from the layer output, like
        # aten's in-place activation backward): dZ_{l-1} = (dZ_l W_l) * act'(H_{l-1})
        out, d = [None] * len(units), d_heads.to(dtype)
        for l in range(len(units), 0, -1):
            w = layers[l][0].to(dtype)
            h = acts[l - 1].to(dtype)
            dh = d @ w
            if act == 'elu':
                d = dh * torch.where(h > 0, torch.ones_like(h), h + 1)
            elif act == 'relu':
                d = dh * (h > 0).to(dtype)
            elif act == 'tanh':
                d = dh * (1 - h * h)
            else:
                d = dh
            out[l - 1] = d
        return out
    r64, r32 = ref(torch.float64), ref(torch.float32)
    for l, u in enumerate(units):
        assert torch.isfinite(dzs[l]).all()
        _close(dzs[l], r64[l], r32[l])
        colsum = parts[l].view(nblk, u).sum(0)
        want = dzs[l].double().sum(0)                        # sums of the kernel's own dZ
        assert torch.allclose(colsum, want, rtol=1e-5, atol=1e-5 * max(1.0, want.abs().max().item()))
    # autograd through the same network in fp64 agrees as well (end-to-end check of the chain)
    ws = [w.double().requires_grad_(True) for w, _, _ in layers]
    a = x.double()
    pre = []
    for (w, b, actn), w64 in zip(layers, ws):
        z = torch.addmm(b.double(), a, w64.t())
        z.retain_grad()
        pre.append(z)
        a = ACT[actn](z)
    a.backward(d_heads.double())
    for l, u in enumerate(units):
        _close(dzs[l], pre[l].grad, None, floor=2e-5)
    # deterministic
    again = [d.clone() for d in dzs]
    chain.backward(d_heads, acts, dzs, parts, groups=groups)
    assert all(torch.equal(p, q) for p, q in zip(again, dzs))


@pytest.mark.parametrize('rows', [32768, 4096, 1000, 37, 2])
def test_dw_odd_shapes_match_fp64(rows):
    """Weight gradients for widths that are no multiples of 16 / 4 and a 22-wide dZ (row stride 88 B)."""
    from rl_games_amd import ops
    g = torch.Generator().manual_seed(rows)
    shapes = [(22, 100), (36, 20), (20, 12), (8, 6), (64, 60), (130, 260)]
    layers, refs = [], []
    for No, Mi in shapes:
        dz = torch.randn(rows, No, generator=g).to(DEV)
        x = torch.randn(rows, Mi, generator=g).to(DEV)
        grad = torch.full((No, Mi), float('nan'), device=DEV)
        layers.append((dz, x, grad))
        refs.append((dz.t() @ x, dz.double().t() @ x.double()))
    plan = ops.MlpDwPlan(shapes, rows, DEV)
    plan.launch(layers)
    for (dz, x, grad), (lib32, t64) in zip(layers, refs):
        assert torch.isfinite(grad).all(), tuple(grad.shape)
        _close(grad, t64, lib32)


@pytest.mark.parametrize('nblocks', [1, 17, 128, 300, 513])
def test_dw_finalize_bias_column_sums(nblocks):
    """Bias gradients ride along in the finalise launch: out[c] = sum over the per-block fp64 partial
    rows the backward kernels leave behind (any row count, widths that are no multiples of 16)."""
    from rl_games_amd import ops
    g = torch.Generator().manual_seed(nblocks)
    rows = 64
    shapes = [(20, 12)]
    dz = torch.randn(rows, 20, generator=g).to(DEV)
    x = torch.randn(rows, 12, generator=g).to(DEV)
    grad = torch.empty(20, 12, device=DEV)
    widths = [400, 100, 22, 7]
    parts = [torch.randn(nblocks * w, generator=g, dtype=torch.float64).to(DEV) for w in widths]
    outs = [torch.full((w,), float('nan'), device=DEV) for w in widths]
    plan = ops.MlpDwPlan(shapes, rows, DEV)
    plan.launch([(dz, x, grad)], [(parts[k], nblocks, widths[k], outs[k]) for k in range(len(widths))])
    assert torch.allclose(grad.double(), dz.double().t() @ x.double(), rtol=1e-5, atol=1e-5)
    for p, w, o in zip(parts, widths, outs):
        want = p.view(nblocks, w).sum(0)
        assert torch.allclose(o.double(), want, rtol=1e-6, atol=1e-6 * max(1.0, want.abs().max().item()))


@pytest.mark.parametrize('in_dim,mb_rows,nmb', [(108, 4096, 3), (60, 1000, 4), (13, 37, 5),
                                                (108, 16384, 2), (13, 20000, 2)])      # (the last two: the split-bf16 forward)
def test_forward_folds_minibatch_moments_like_running_mean_std(in_dim, mb_rows, nmb):
    """Training-mode RunningMeanStd.forward = update, then normalise (running_mean_std.py:69-84).  The
    fused forward folds the epoch's precomputed minibatch moments in its prologue; the state and the
    normalised observations must follow the stand-alone update + apply kernels step by step."""
    from rl_games_amd import ops
    from rl_games_amd.normalizers import RunningMeanStd
    layers, g = _net(in_dim, [32], 5, 'elu', seed=in_dim)
    chain = ops.MlpChain(layers, DEV)
    obs = (2.5 * torch.randn(nmb * mb_rows, in_dim, generator=g) + 0.7).to(DEV)
    a, b = RunningMeanStd(in_dim).to(DEV), RunningMeanStd(in_dim).to(DEV)
    for m in (a, b):
        m.running_mean.copy_(torch.linspace(-1, 1, in_dim, dtype=torch.float64))
        m.running_var.copy_(torch.linspace(0.5, 2, in_dim, dtype=torch.float64))
        m.count.fill_(777)
    table = a.precompute_minibatch_moments(obs, mb_rows)
    want = torch.cat([obs.double().view(nmb, mb_rows, in_dim).sum(1),
                      (obs.double() ** 2).view(nmb, mb_rows, in_dim).sum(1),
                      torch.full((nmb, 1), float(mb_rows), dtype=torch.float64, device=DEV)], dim=1)
    assert torch.allclose(table, want, rtol=1e-12, atol=1e-9)
    for i in range(nmb):
        x = obs[i * mb_rows:(i + 1) * mb_rows]
        rms, fold = a.fold_buffers(i)
        heads = torch.empty(mb_rows, 5, device=DEV)
        xn = torch.empty(mb_rows, in_dim, device=DEV)
        chain.forward(x, heads, rms=rms, eps=a.epsilon, xn_out=xn, rms_fold=fold)
        b.train()
        xn_ref = b(x)                                        # update + apply kernels
        heads_ref = torch.empty(mb_rows, 5, device=DEV)
        chain.forward(xn_ref, heads_ref)
        new_mean, new_var, new_count = fold[2], fold[3], fold[4]
        assert new_count.item() == b.count.item() == 777 + (i + 1) * mb_rows
        assert torch.allclose(new_mean, b.running_mean, rtol=1e-12, atol=1e-13)
        assert torch.allclose(new_var, b.running_var, rtol=1e-12, atol=1e-13)
        # same fp32 mean / denominator unless the 1e-12 state difference crosses a rounding boundary
        assert torch.allclose(xn, xn_ref, rtol=0, atol=1e-6)
        assert (xn != xn_ref).float().mean().item() < 1e-3
        assert torch.allclose(heads, heads_ref, rtol=1e-5, atol=1e-5)
    a.fold_sync(nmb)
    assert torch.allclose(a.running_mean, b.running_mean, rtol=1e-12, atol=1e-13)
    assert torch.allclose(a.running_var, b.running_var, rtol=1e-12, atol=1e-13)
    assert a.count.item() == b.count.item()
    # the fold needs a second buffer set
    with pytest.raises(RuntimeError):
        chain.forward(obs[:mb_rows], heads, rms=(a.running_mean, a.running_var),
                      rms_fold=(table[0], a.count, a.running_mean, a.running_var, a.count))


@pytest.mark.parametrize('nblocks,A', [(512, 21), (64, 8), (1, 1), (300, 33), (77, 0)])
def test_dw_finalize_folds_loss_partials_like_ppo_loss_finalize(nblocks, A):
    """The weight-gradient finalise launch can fold the PPO loss partials (rlg_loss_finalize_desc): same
    scalars, KL, d logstd and head-bias gradients as the stand-alone rlg_ppo_loss_finalize."""
    from rl_games_amd import ops
    g = torch.Generator().manual_seed(nblocks + A)
    W = 7 + 2 * A
    partials = torch.randn(nblocks, W, generator=g, dtype=torch.float64)
    partials[:, 5] = torch.rand(nblocks, generator=g, dtype=torch.float64) * 64     # mask sums
    partials = partials.to(DEV)
    mb = nblocks * 64
    dz = torch.randn(64, 20, generator=g).to(DEV)
    x = torch.randn(64, 12, generator=g).to(DEV)
    grad = torch.empty(20, 12, device=DEV)
    plan = ops.MlpDwPlan([(20, 12)], 64, DEV)
    out = {}
    for masked in (False, True):
        for fused in (True, False):
            scalars = torch.full((8,), float('nan'), device=DEV)
            d_logstd = torch.full((max(A, 1),), float('nan'), device=DEV)
            kl = torch.full((1,), float('nan'), device=DEV)
            dmb = torch.full((max(A, 1),), float('nan'), device=DEV)
            dvb = torch.full((1,), float('nan'), device=DEV)
            args = (partials, nblocks, A, mb, masked, 2.0, 0.01, 1e-4, scalars, d_logstd, kl, dmb, dvb)
            if fused:
                plan.launch([(dz, x, grad)], loss_finalize=ops.loss_finalize_desc(*args))
            else:
                ops.ppo_loss_finalize(*args)
            out[fused] = (scalars, d_logstd[:A], kl, dmb[:A], dvb)
        for got, ref in zip(out[True], out[False]):
            assert torch.isfinite(got).all()
            assert torch.allclose(got, ref, rtol=1e-6, atol=1e-7 * max(1.0, ref.abs().max().item() if ref.numel() else 1.0))
    assert torch.allclose(grad.double(), dz.double().t() @ x.double(), rtol=1e-5, atol=1e-5)
    # by-product: per-block sums of (g * scale)^2 over every element the launch wrote + the step counter
    cparts = torch.randn(5 * 36, generator=g, dtype=torch.float64).to(DEV)
    cout = torch.empty(36, device=DEV)
    colsums = [(cparts, 5, 36, cout)]
    desc = ops.loss_finalize_desc(*args)
    norm_partials = torch.full((plan.finalize_blocks(colsums, desc) + 3,), float('nan'), dtype=torch.float64, device=DEV)
    counter = torch.tensor([41], dtype=torch.int64, device=DEV)
    nb = plan.launch([(dz, x, grad)], colsums, desc, norm=(norm_partials, 0.5, counter))
    assert nb == plan.finalize_blocks(colsums, desc) and counter.item() == 42
    written = [grad.reshape(-1), cout, d_logstd[:A], dmb[:A], dvb]
    want = sum(((0.5 * t).double() ** 2).sum() for t in written)
    assert torch.isfinite(norm_partials[:nb]).all() and torch.isnan(norm_partials[nb:]).all()
    assert torch.allclose(norm_partials[:nb].sum(), want, rtol=1e-12)


@pytest.mark.parametrize('rows,groups', [(32768, 0), (16384, 2), (1000, 4), (100, 2), (7, 1)])
@pytest.mark.parametrize('variant', ['plain', 'smooth_reg_noclip'])
def test_backward_evaluates_ppo_loss_like_the_loss_kernel(rows, groups, variant):
    """Backward launch with a loss descriptor = rlg_ppo_loss_fused, then backward: the same per-row
    arithmetic on the backward's 16 / 32 / 64-row tiles -> d heads, every dZ and the mu/sigma write-back
    bit-identical, the reduced scalars and column sums equal up to fp64 summation order."""
    from rl_games_amd import ops
    A, V = 21, 1
    layers, g = _net(108, [64, 32], V + A, 'elu', seed=rows)
    chain = ops.MlpChain(layers, DEV)
    x = torch.randn(rows, 108, generator=g).to(DEV)
    logstd = (0.1 * torch.randn(A, generator=g) - 0.3).to(DEV)
    heads = torch.empty(rows, V + A, device=DEV)
    acts = [torch.empty(rows, u, device=DEV) for u in (64, 32)]
    chain.forward(x, heads, act_out=acts, groups=groups)

    def data():
        gg = torch.Generator().manual_seed(rows + 1)
        d = {'actions': torch.randn(rows, A, generator=gg), 'old_neglogp': 25 + torch.randn(rows, generator=gg),
             'adv': torch.randn(rows, generator=gg), 'old_values': torch.randn(rows, generator=gg),
             'returns': torch.randn(rows, generator=gg), 'old_mu': 0.3 * torch.randn(rows, A, generator=gg),
             'old_sigma': 0.5 + torch.rand(rows, A, generator=gg)}
        return {k: v.to(DEV) for k, v in d.items()}
    kw = dict(clip_value=True, smooth=False, bound_kind=1)
    if variant != 'plain':
        kw = dict(clip_value=False, smooth=True, bound_kind=2)
    out = {}
    for fused in (True, False):
        d = data()
        d_heads = torch.full((rows, V + A), float('nan'), device=DEV)
        dzs = [torch.full((rows, u), float('nan'), device=DEV) for u in (64, 32)]
        nbw = chain.num_blocks(rows, 1, groups)
        parts = [torch.empty(nbw * u, dtype=torch.float64, device=DEV) for u in (64, 32)]
        nblk = nbw if fused else ops.ppo_loss_blocks(rows)
        partials = torch.full((nblk, ops.ppo_loss_partials_per_block(A)), float('nan'), dtype=torch.float64, device=DEV)
        args = (heads[:, V:], logstd, heads[:, 0], d['actions'], d['old_neglogp'], d['adv'], d['old_values'],
                d['returns'], d['old_mu'], d['old_sigma'], d_heads[:, V:], d_heads[:, 0], partials, 0.2, 2.0, 1e-4)
        if fused:
            chain.backward(d_heads, acts, dzs, parts, groups=groups, ppo_loss=ops.ppo_loss_desc(*args, **kw))
        else:
            ops.ppo_loss_fused(*args, **kw)
            chain.backward(d_heads, acts, dzs, parts, groups=groups)
        scalars = torch.zeros(8, device=DEV)
        d_logstd, kl = torch.zeros(A, device=DEV), torch.zeros(1, device=DEV)
        d_mu_bias, d_v_bias = torch.zeros(A, device=DEV), torch.zeros(1, device=DEV)
        ops.ppo_loss_finalize(partials, nblk, A, rows, False, 2.0, 0.0, 1e-4, scalars, d_logstd, kl, d_mu_bias, d_v_bias)
        out[fused] = (d_heads, dzs[0], dzs[1], d['old_mu'], d['old_sigma'], scalars, d_logstd, d_mu_bias, d_v_bias)
    for k in range(5):
        assert torch.isfinite(out[True][k]).all()
        assert torch.equal(out[True][k], out[False][k]), k
    for k in range(5, 9):
        ref = out[False][k]
        assert torch.allclose(out[True][k], ref, rtol=1e-6, atol=1e-7 * max(1.0, ref.abs().max().item())), k


@pytest.mark.parametrize('rows', [4096, 1000, 37])
@pytest.mark.parametrize('in_dim,units,A', [(108, [400, 200, 100], 21), (60, [256, 128, 64], 8), (12, [100, 52], 11)])
def test_one_launch_step_equals_forward_then_backward(rows, in_dim, units, A):
    """Round 4: forward + PPO loss + backward of a minibatch below 16,384 rows as ONE launch (rlg_mlp_chain_step,
    mlp_chain_step_pipe_kernel: the loss tile's inputs requested before the forward, no launch boundary between the
    halves) against rlg_mlp_chain_forward followed by rlg_mlp_chain_backward with the loss descriptor - the same device
    code, so everything is bit-identical: heads, activations, normalised observations, the folded RunningMeanStd state,
    d heads, every dZ, the bias partials, the loss partials and the mu / sigma write-back."""
    from rl_games_amd import ops
    V = 1
    layers, g = _net(in_dim, units, V + A, 'elu', seed=rows + A)
    chain = ops.MlpChain(layers, DEV)
    x = (2 * torch.randn(rows, in_dim, generator=g) + 0.5).to(DEV)
    logstd = (0.1 * torch.randn(A, generator=g) - 0.3).to(DEV)
    gg = torch.Generator().manual_seed(rows + 1)
    base = {'actions': torch.randn(rows, A, generator=gg), 'old_neglogp': 25 + torch.randn(rows, generator=gg),
            'adv': torch.randn(rows, generator=gg), 'old_values': torch.randn(rows, generator=gg),
            'returns': torch.randn(rows, generator=gg), 'old_mu': 0.3 * torch.randn(rows, A, generator=gg),
            'old_sigma': 0.5 + torch.rand(rows, A, generator=gg)}
    # training-mode statistics fold in the prologue: moments of this minibatch + a state to fold them into
    moments, _ = ops.column_moments_segments(x, rows)
    out = {}
    for fused in (True, False):
        d = {k: v.clone().to(DEV) for k, v in base.items()}
        mean = torch.zeros(in_dim, dtype=torch.float64, device=DEV) + 0.25
        var = torch.ones(in_dim, dtype=torch.float64, device=DEV) * 3.0
        count = torch.tensor([1000], dtype=torch.int64, device=DEV)
        mean2, var2, count2 = torch.zeros_like(mean), torch.zeros_like(var), torch.zeros_like(count)
        heads = torch.full((rows, V + A), float('nan'), device=DEV)
        acts = [torch.full((rows, u), float('nan'), device=DEV) for u in units]
        xn = torch.full((rows, in_dim), float('nan'), device=DEV)
        d_heads = torch.full((rows, V + A), float('nan'), device=DEV)
        dzs = [torch.full((rows, u), float('nan'), device=DEV) for u in units]
        nbw = chain.num_blocks(rows, 1)
        parts = [torch.full((nbw * u,), float('nan'), dtype=torch.float64, device=DEV) for u in units]
        partials = torch.full((nbw, ops.ppo_loss_partials_per_block(A)), float('nan'), dtype=torch.float64, device=DEV)
        desc = ops.ppo_loss_desc(heads[:, V:], logstd, heads[:, 0], d['actions'], d['old_neglogp'], d['adv'], d['old_values'],
                                 d['returns'], d['old_mu'], d['old_sigma'], d_heads[:, V:], d_heads[:, 0], partials, 0.2, 2.0,
                                 1e-4, clip_value=True, smooth=False, bound_kind=1)
        fold = (moments[0], count, mean2, var2, count2)
        if fused:
            assert chain.step(x, heads, acts, d_heads, dzs, parts, desc, rms=(mean, var), eps=1e-5, xn_out=xn, rms_fold=fold)
        else:
            chain.forward(x, heads, act_out=acts, rms=(mean, var), eps=1e-5, xn_out=xn, rms_fold=fold)
            chain.backward(d_heads, acts, dzs, parts, ppo_loss=desc)
        torch.cuda.synchronize()
        out[fused] = [heads, xn, d_heads, d['old_mu'], d['old_sigma'], partials, mean2, var2, count2] + acts + dzs + parts
    for k, (a, b) in enumerate(zip(out[True], out[False])):
        assert torch.isfinite(a.double()).all(), k
        assert torch.equal(a, b), k


def test_one_launch_step_declines_what_it_does_not_cover():
    """rlg_mlp_chain_step / rlg_mlp_chain_step_lean return hipErrorNotSupported (MlpChain.step -> False, nothing launched)
    for minibatches that run the split-bf16 kernels or need more than one round of workgroups (> 16 rows x CUs: two
    launches with two workgroups per CU are faster there) and without a loss descriptor; the pipelined form also for
    weights outside one arena."""
    from rl_games_amd import ops
    layers, g = _net(60, [64, 32], 9, 'elu', seed=1)
    chain = ops.MlpChain(layers, DEV)
    rows = 16384
    x = torch.randn(rows, 60, generator=g).to(DEV)
    heads = torch.empty(rows, 9, device=DEV)
    acts = [torch.empty(rows, u, device=DEV) for u in (64, 32)]
    assert chain.step(x, heads, acts, torch.empty(rows, 9, device=DEV), [torch.empty_like(a) for a in acts], None, None) is False
    # (52 x 60 floats = 12,480 bytes: the allocator rounds the block to 12,800, so nothing of the network can lie within
    #  16 bytes behind the first matrix - with 64 x 60 = 30 x 512 bytes the bias may follow it and the launch is covered)
    loose, g = _net(60, [52, 32], 9, 'elu', seed=1, packed=False)
    chain2 = ops.MlpChain(loose, DEV)
    chain2._lean = False                  # (the lean form reads packed fragments: separately allocated weights are fine there)
    rows = 256
    x = torch.randn(rows, 60, generator=g).to(DEV)
    heads = torch.full((rows, 9), float('nan'), device=DEV)
    acts = [torch.empty(rows, u, device=DEV) for u in (52, 32)]
    d_heads = torch.empty(rows, 9, device=DEV)
    partials = torch.empty(chain2.num_blocks(rows, 1), ops.ppo_loss_partials_per_block(8), dtype=torch.float64, device=DEV)
    z = lambda *s: torch.zeros(*s, device=DEV)
    desc = ops.ppo_loss_desc(heads[:, 1:], z(8), heads[:, 0], z(rows, 8), z(rows), z(rows), z(rows), z(rows), z(rows, 8),
                             z(rows, 8) + 1, d_heads[:, 1:], d_heads[:, 0], partials, 0.2, 2.0, 1e-4, clip_value=True,
                             smooth=False, bound_kind=1)
    parts = [torch.empty(chain2.num_blocks(rows, 1) * u, dtype=torch.float64, device=DEV) for u in (52, 32)]
    assert chain2.step(x, heads, acts, d_heads, [torch.empty_like(a) for a in acts], parts, desc) is False
    assert torch.isnan(heads).all()                    # nothing ran


def test_engine_fused_chain_equals_per_layer_engine():
    """ManualMLP with the fused chain vs the per-layer (library GEMM) engine: same heads, same
    gradients in the arena, on a BASELINE config #2 shaped network."""
    from rl_games_amd import configs
    from rl_games_amd.agent import A2CAgent
    outs = {}
    for fused in (True, False):
        torch.manual_seed(3)
        params = configs.ant_4096(num_actors=256, minibatch_size=2048, fused_mlp=fused, hip_graphs=False)
        agent = A2CAgent('t', params)
        eng = agent._engine
        assert (eng.chain is not None) == fused
        g = torch.Generator().manual_seed(5)
        obs = (2 * torch.randn(2048, 60, generator=g) + 0.5).to(DEV)
        m = agent.model.running_mean_std
        m.running_mean.copy_(torch.randn(60, generator=g, dtype=torch.float64))
        m.running_var.copy_(torch.rand(60, generator=g, dtype=torch.float64) + 0.5)
        if fused:
            heads = eng.forward_obs(obs, (m.running_mean, m.running_var), m.epsilon).clone()
        else:
            from rl_games_amd import ops
            heads = eng.forward(ops.rms_apply(obs, m.running_mean, m.running_var, m.epsilon, 0)).clone()
        d_heads = eng.d_heads[:2048]
        d_heads.copy_(torch.randn(2048, 9, generator=g).to(DEV))
        agent.optimizer.flat_grads.fill_(float('nan'))
        eng.backward(d_heads)
        torch.cuda.synchronize()
        grads = {n: p.grad.clone() for n, p in agent.model.a2c_network.named_parameters()
                 if p.grad is not None and 'bias' not in n.split('.')[-1] or n.startswith('actor_mlp')}
        outs[fused] = (heads, grads)
    assert torch.allclose(outs[True][0], outs[False][0], rtol=2e-5, atol=2e-6)
    for n, gref in outs[False][1].items():
        got = outs[True][1][n]
        if not torch.isfinite(gref).all():
            continue                                     # head biases come from the loss kernel
        assert torch.allclose(got, gref, rtol=1e-4, atol=1e-5 * max(1.0, gref.abs().max().item())), n


def test_full_size_rows_are_computed_independently_of_their_position():
    """Size-independent properties at the BASELINE size (65,536 rows, 108 -> [400, 200, 100] -> 22): a row's
    forward / backward result does not depend on where the row sits (bit-exact under a row permutation: the
    same MFMA sequence per row wherever its tile is), the weight gradients are a sum over rows (equal under
    the permutation up to summation order, and linear in dZ), and every launch is deterministic."""
    from rl_games_amd import ops
    rows = 65536
    layers, g = _net(108, [400, 200, 100], 22, 'elu', seed=65536)
    chain = ops.MlpChain(layers, DEV)
    x = torch.randn(rows, 108, generator=g).to(DEV)
    d_heads = torch.randn(rows, 22, generator=g).to(DEV)
    perm = torch.randperm(rows, generator=g).to(DEV)

    def run(xx, dd):
        heads = torch.empty(rows, 22, device=DEV)
        acts = [torch.empty(rows, u, device=DEV) for u in (400, 200, 100)]
        dzs = [torch.empty(rows, u, device=DEV) for u in (400, 200, 100)]
        nb = chain.num_blocks(rows, 1)
        parts = [torch.empty(nb * u, dtype=torch.float64, device=DEV) for u in (400, 200, 100)]
        chain.forward(xx, heads, act_out=acts)
        chain.backward(dd, acts, dzs, parts)
        jobs = [(dd, acts[2], torch.empty(22, 100, device=DEV)), (dzs[2], acts[1], torch.empty(100, 200, device=DEV)),
                (dzs[1], acts[0], torch.empty(200, 400, device=DEV)), (dzs[0], xx, torch.empty(400, 108, device=DEV))]
        plan = ops.MlpDwPlan([tuple(j[2].shape) for j in jobs], rows, DEV)
        plan.launch(jobs)
        return heads, acts, dzs, [j[2] for j in jobs]
    h0, a0, z0, w0 = run(x, d_heads)
    h1, a1, z1, w1 = run(x[perm].contiguous(), d_heads[perm].contiguous())
    assert torch.equal(h1, h0[perm])
    for p, q in zip(a1 + z1, a0 + z0):
        assert torch.equal(p, q[perm])
    for p, q in zip(w1, w0):
        assert torch.allclose(p, q, rtol=1e-4, atol=1e-4 * q.abs().max().item())
    h2, a2, z2, w2 = run(x, d_heads)                       # deterministic
    assert torch.equal(h2, h0) and all(torch.equal(p, q) for p, q in zip(z2 + w2, z0 + w0))
    # dW is linear in dZ: G(2 dZ) = 2 G(dZ) exactly (power-of-two scaling commutes with every rounding)
    _, _, _, w3 = run(x, 2.0 * d_heads)
    for p, q in zip(w3, w0):
        assert torch.equal(p, 2.0 * q)


def test_random_network_shapes_fuzz():
    """tools/exp/fuzz_chain.py: 60 random (observation width, hidden widths, action count, activation, rows,
    row groups) combinations - forward, backward with and without the loss tile, weight gradients - against
    fp64 torch.  Guards the shape-dependent paths (remainder blocks of 4- and 8-wave workgroups, LDS region
    sizes, ragged tiles) that the fixed shape list above cannot enumerate."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, 'tools', 'exp', 'fuzz_chain.py'), '60', '3'],
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert res.returncode == 0, res.stderr[-2000:]
    assert '0 bad of 60' in res.stdout, [l for l in res.stdout.splitlines() if 'BAD' in l][:5]


def test_dw_exact_f32_product_form_passes_the_same_tests():
    """The weight gradients default to split-bf16 products (csrc/mlp_dw.hip); RLG_DW_BF16=0 selects the
    exact-f32 kernel.  The library reads the variable once per process, so the dW tests of this file and of
    test_ops_gpu.py are re-run in a child process with it set."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RLG_DW_BF16='0')
    res = subprocess.run([sys.executable, '-m', 'pytest', 'tests/test_ops_gpu.py', 'tests/test_mlp_chain_gpu.py', '-q', '-m', 'gpu',
                          '-k', 'dw and not exact_f32', '-p', 'no:cacheprovider'],
                         capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-1000:]
    assert ' passed' in res.stdout and 'failed' not in res.stdout, res.stdout[-500:]


# ------------------------------------------------------------------ split-bf16 chain (csrc/mlp_chain_bx.hip)

def _bf16_bits_to_f64(u16):
    return (u16.to(torch.int32) << 16).view(torch.float32).double()


def _split_form():
    """(planes per operand, plane bits -> fp64, weight scale) of this build's split-product chain kernels."""
    from rl_games_amd import ops
    products, ptype = ops.chain_split_form()
    if ptype == 'fp16':
        return 2, (lambda u16: u16.view(torch.float16).double()), 64.0           # csrc/bx_form.hpp kBxScaleW
    return 3, _bf16_bits_to_f64, 1.0


@pytest.mark.parametrize('direction', [0, 1])
@pytest.mark.parametrize('in_dim,units,out_dim,act', SHAPES)
def test_weight_planes_are_a_split_in_fragment_order(in_dim, units, out_dim, act, direction):
    """rlg_mlp_chain_pack_planes: fragment (block, chunk, plane) = 64 lanes x 8 half-width values; lane l, element e holds
    A[16 block + (l & 15)][32 chunk + (e < 4 ? 4 (l >> 4) + e : 16 + 4 (l >> 4) + e - 4)], A = W (forward) or W^T
    (backward), zero outside the matrix.  fp16 form (the default build): two planes of 64 W, h0 = RN16(64 W) and
    h1 = RN16(64 W - h0) bit for bit, their sum within 2^-22 of 64 W (2^-25 absolute below the fp16 normal range);
    bf16 form: the three planes add up to the fp32 weight EXACTLY."""
    from rl_games_amd import ops
    layers, _ = _net(in_dim, units, out_dim, act, seed=5 + in_dim)
    chain = ops.MlpChain(layers, DEV)
    planes = chain.pack_planes(direction, layers[0][0])
    torch.cuda.synchronize()
    np_, to64, wscale = _split_form()
    raw = planes.cpu().view(torch.int16)
    off = 0
    lane = torch.arange(64)
    e = torch.arange(8)
    for L, (w, _, _) in enumerate(layers):
        if direction == 1 and L == 0:
            continue
        A = w.cpu() if direction == 0 else w.cpu().t()
        I, K = A.shape
        nb, kc = (I + 15) // 16, (K + 31) // 32
        frag = raw[off // 2: off // 2 + nb * kc * np_ * 512].view(nb, kc, np_, 64, 8)
        off += nb * kc * np_ * 1024
        i = (torch.arange(nb)[:, None, None, None] * 16 + (lane & 15)[None, None, :, None]).expand(nb, kc, 64, 8)
        q4 = 4 * (lane >> 4)[None, None, :, None]
        k = torch.arange(kc)[None, :, None, None] * 32 + torch.where(e < 4, q4 + e, 16 + q4 + e - 4)
        k = k.expand(nb, kc, 64, 8)
        inside = (i < I) & (k < K)
        want = torch.zeros(nb, kc, 64, 8, dtype=torch.float32)
        want[inside] = A[i[inside], k[inside]] * wscale
        if np_ == 3:
            total = sum(to64(frag[:, :, p]) for p in range(3))          # [nb, kc, 64, 8]
            assert torch.equal(total, want.double())
        else:
            h0 = want.half()
            h1 = (want - h0.float()).half()
            assert torch.equal(frag[:, :, 0].contiguous().view(torch.float16), h0)
            assert torch.equal(frag[:, :, 1].contiguous().view(torch.float16), h1)
            err = (h0.double() + h1.double() - want.double()).abs()
            assert torch.all(err <= torch.maximum(want.double().abs() * 2.0 ** -22, torch.tensor(2.0 ** -25, dtype=torch.float64)))
    assert off == planes.numel()
    if direction == 1:
        # one launch for both directions leaves the same fragments (the backward ones behind the forward ones)
        both = chain.pack_planes(2, layers[0][0]).cpu()
        fwd = chain.pack_planes(0, layers[0][0]).cpu()
        assert torch.equal(both[:fwd.numel()], fwd) and torch.equal(both[chain._bwd_offset:chain._bwd_offset + planes.numel()], planes.cpu())


@pytest.mark.parametrize('in_dim,units,out_dim,act', SHAPES)
@pytest.mark.parametrize('rows', [64, 1000, 16384, 32768 + 17])
def test_split_bf16_backward_agrees_with_the_exact_product_kernel(in_dim, units, out_dim, act, rows):
    """The split-bf16 backward (64-row tiles, weight planes) against the exact-f32-product kernel on the same inputs:
    not the same bits (that would mean the split kernel did not run), but within 1e-6 of the tensor scale - each
    product differs by at most 3 * 2^-24 of its magnitude - and the bias-gradient partial sums add up to the same
    column sums.  Tolerance of the chain against fp64: test_chain_backward_matches_autograd_fp64 (same bound as the
    exact kernel)."""
    from rl_games_amd import ops
    layers, g = _net(in_dim, units, out_dim, act, seed=3 * rows + in_dim)
    chain = ops.MlpChain(layers, DEV)
    assert chain.split_products(rows, 1, 4)
    x = torch.randn(rows, in_dim, generator=g).to(DEV)
    heads = torch.empty(rows, out_dim, device=DEV)
    acts = [torch.empty(rows, u, device=DEV) for u in units]
    chain.forward(x, heads, act_out=acts)
    d_heads = torch.randn(rows, out_dim, generator=g).to(DEV)
    nblk = chain.num_blocks(rows, 1, 4)
    out = {}
    for split in (False, None):
        dzs = [torch.full((rows, u), float('nan'), device=DEV) for u in units]
        parts = [torch.full((nblk * u,), float('nan'), dtype=torch.float64, device=DEV) for u in units]
        chain.backward(d_heads, acts, dzs, parts, groups=4, split_products=split)
        out[split] = (dzs, parts)
    differs = False
    for l, u in enumerate(units):
        exact, got = out[False][0][l], out[None][0][l]
        assert torch.isfinite(got).all()
        scale = exact.abs().max().item()
        assert (got - exact).abs().max().item() <= 1e-6 * scale
        differs = differs or not torch.equal(got, exact)
        cs_exact = out[False][1][l].view(nblk, u).sum(0)
        cs_got = out[None][1][l].view(nblk, u).sum(0)
        # (sums over the rows of per-element differences of <= 1e-6 of the scale)
        assert torch.allclose(cs_got, cs_exact, rtol=1e-5, atol=1e-6 * scale * rows ** 0.5 + 1e-6)
        want = got.double().sum(0)
        assert torch.allclose(cs_got, want, rtol=1e-5, atol=1e-5 * max(1.0, want.abs().max().item()))
    assert differs or len(units) == 1 and act == 'None'


@pytest.mark.parametrize('in_dim,units,out_dim,act', SHAPES + [(130, [256, 256, 128], 34, 'elu'), (33, [512, 64], 8, 'relu'),
                                                       (16, [64, 512, 64], 8, 'tanh'), (24, [32, 96, 448, 208], 6, 'elu')])
@pytest.mark.parametrize('rows', [16384, 32768 + 17])
def test_split_bf16_forward_agrees_with_the_exact_product_kernel(in_dim, units, out_dim, act, rows):
    """The split-bf16 forward (64-row tiles, weight planes; the 400-, 448- and 512-wide layers run windowed - as the
    second, third or fourth tile of the chain) against the
    exact-f32-product kernels on the same inputs: every activation and the heads within 4e-6 of the tensor scale (a
    product differs by at most 3 * 2^-24 of its magnitude, layer by layer; measured up to 2.1e-6 behind four layers), the normalised observations bit for bit,
    the inference form the same bits as the training form.  Tolerance against fp64: test_chain_forward_matches_fp64
    runs the same kernel (rows >= 16,384) under the exact kernels' bound."""
    from rl_games_amd import ops
    layers, g = _net(in_dim, units, out_dim, act, seed=11 * rows + in_dim)
    chain = ops.MlpChain(layers, DEV)
    assert chain.split_products(rows, 0)
    x = (3 * torch.randn(rows, in_dim, generator=g) + 1).to(DEV)
    mean = torch.randn(in_dim, generator=g, dtype=torch.float64).to(DEV)
    var = (torch.rand(in_dim, generator=g, dtype=torch.float64) * 4 + 0.1).to(DEV)
    out = {}
    for split in (False, None):
        heads = torch.full((rows, out_dim), float('nan'), device=DEV)
        acts = [torch.full((rows, u), float('nan'), device=DEV) for u in units]
        xn = torch.full((rows, in_dim), float('nan'), device=DEV)
        chain.forward(x, heads, act_out=acts, rms=(mean, var), eps=1e-5, xn_out=xn, split_products=split)
        out[split] = (acts + [heads], xn)
    assert torch.equal(out[None][1], out[False][1])
    differs = False
    for got, exact in zip(out[None][0], out[False][0]):
        assert torch.isfinite(got).all()
        assert (got - exact).abs().max().item() <= 4e-6 * exact.abs().max().item()
        differs = differs or not torch.equal(got, exact)
    assert differs
    heads2 = torch.full((rows, out_dim), float('nan'), device=DEV)
    chain.forward(x, heads2, rms=(mean, var), eps=1e-5)
    assert torch.equal(heads2, out[None][0][-1])


def test_split_forward_scales_raw_observation_rows_by_their_own_maxima():
    """Raw (un-normalised) observations have no bound: the fp16 form scales every ROW by the power of two its largest
    magnitude asks for.  (1) Rows of very different magnitude in one tile do not disturb one another: a row gives the SAME
    BITS whatever its neighbours hold.  (2) Observations times 2^-e with first-layer weights times 2^e are the same
    network: the outputs agree to the split kernels' tolerance (not bit for bit - the WEIGHT scale is fixed at 2^6, and
    the low plane of a small weight sits in fp16's gradual underflow or not depending on e; the bf16 form of rounds 3 - 5,
    with fp32's exponent range, reproduced the bits for any e).  Documented limit of both forms: a +-Inf operand becomes
    NaN in the split (Inf - Inf in the residual), where an exact product would propagate the infinity."""
    from rl_games_amd import ops
    rows = 16384
    base, g = _net(60, [256, 128], 9, 'elu', seed=77)
    x = (2 * torch.randn(rows, 60, generator=g)).to(DEV)

    def run(scale_x, scale_w, xin=None):
        layers = [(w.clone(), b.clone(), a) for w, b, a in base]
        layers[0][0].mul_(scale_w)
        chain = ops.MlpChain(layers, DEV)
        assert chain.split_products(rows, 0)
        heads = torch.empty(rows, 9, device=DEV)
        acts = [torch.empty(rows, 256, device=DEV), torch.empty(rows, 128, device=DEV)]
        chain.forward((x if xin is None else xin) * scale_x, heads, act_out=acts)
        return acts + [heads]
    ref = run(1.0, 1.0)
    for e in (3, 7):
        got = run(2.0 ** -e, 2.0 ** e)
        for p, q in zip(got, ref):
            assert (p - q).abs().max().item() <= 4e-6 * q.abs().max().item(), e
    # every third row a million times smaller: its outputs are what it gives inside a tile of rows like itself
    small = x.clone()
    small[::3] *= 2.0 ** -20
    mixed = run(1.0, 1.0, small)
    alone = run(1.0, 1.0, x * 2.0 ** -20)
    for m, a_ in zip(mixed, alone):
        assert torch.equal(m[::3], a_[::3])


# ----------------------------------------------------------------------------- non-finite inputs (round 4)

@pytest.mark.parametrize('rows', [16384, 4096])
def test_non_finite_inputs_stay_non_finite_and_stay_in_their_rows(rows):
    """+-Inf / NaN in an observation row (no normaliser in front: RunningMeanStd would clamp +-Inf to +-5 like the
    reference does) must come out of the forward as NON-FINITE heads of THAT row - never as a finite wrong value - and
    must not touch any other row; the same for d heads -> dZ in the backward, and for the weight gradients (the columns
    / rows a non-finite operand element feeds).  rows = 16,384 runs the split-bf16 kernels (an Inf operand becomes NaN
    in the plane split: Inf - Inf in the residual - documented), 4,096 the exact-product 16-row kernels."""
    from rl_games_amd import ops
    layers, g = _net(60, [256, 128], 9, 'elu', seed=5)
    chain = ops.MlpChain(layers, DEV)
    x = torch.randn(rows, 60, generator=g).to(DEV)
    bad_rows = {7: float('inf'), 100: float('-inf'), rows - 3: float('nan')}

    def fwd(xin):
        heads = torch.empty(rows, 9, device=DEV)
        acts = [torch.empty(rows, 256, device=DEV), torch.empty(rows, 128, device=DEV)]
        chain.forward(xin, heads, act_out=acts)
        return acts, heads
    acts0, heads0 = fwd(x)
    assert torch.isfinite(heads0).all()
    xb = x.clone()
    for r, v in bad_rows.items():
        xb[r, 11] = v
    acts1, heads1 = fwd(xb)
    clean = torch.ones(rows, dtype=torch.bool, device=DEV)
    for r in bad_rows:
        clean[r] = False
        assert not torch.isfinite(heads1[r]).any(), (r, heads1[r])
        # (the first hidden layer: elu(-Inf) = -1 is finite and correct with exact products; at least the units with a
        #  positive weight on the bad observation are not)
        assert not torch.isfinite(acts1[0][r]).all()
    assert torch.equal(heads1[clean], heads0[clean])
    for a1, a0 in zip(acts1, acts0):
        assert torch.equal(a1[clean], a0[clean])

    # backward: non-finite d heads of a row -> non-finite dZ of that row only
    d_heads = torch.randn(rows, 9, generator=g).to(DEV)
    nblk = chain.num_blocks(rows, 1)

    def bwd(dh):
        dzs = [torch.empty(rows, 256, device=DEV), torch.empty(rows, 128, device=DEV)]
        parts = [torch.empty(nblk * u, dtype=torch.float64, device=DEV) for u in (256, 128)]
        chain.backward(dh, acts0, dzs, parts)
        return dzs
    dz0 = bwd(d_heads)
    dhb = d_heads.clone()
    for r, v in bad_rows.items():
        dhb[r, 3] = v
    dz1 = bwd(dhb)
    for r in bad_rows:
        for d in dz1:
            assert not torch.isfinite(d[r]).any(), r
    for d1, d0 in zip(dz1, dz0):
        assert torch.equal(d1[clean], d0[clean])

    # weight gradients: dW[o, :] of the column o that a non-finite dZ element feeds, dW[:, i] for a non-finite X element
    dz = torch.randn(rows, 128, generator=g).to(DEV)
    xx = torch.randn(rows, 256, generator=g).to(DEV)
    plan = ops.MlpDwPlan([(128, 256)], rows, DEV)
    grad0 = torch.empty(128, 256, device=DEV)
    plan.launch([(dz, xx, grad0)])
    dzb, xxb = dz.clone(), xx.clone()
    dzb[5, 17] = float('inf')
    xxb[rows - 1, 200] = float('nan')
    grad1 = torch.empty(128, 256, device=DEV)
    plan.launch([(dzb, xxb, grad1)])
    assert not torch.isfinite(grad1[17]).any() and not torch.isfinite(grad1[:, 200]).any()
    keep = torch.ones(128, 256, dtype=torch.bool, device=DEV)
    keep[17] = False
    keep[:, 200] = False
    assert torch.equal(grad1[keep], grad0[keep])


def test_nan_observation_survives_the_normaliser_clamp():
    """torch.clamp propagates NaN (running_mean_std.py:112-113: clamp((x - mean) / sqrt(var + eps), -5, 5)); v_max_f32 /
    v_min_f32 return the non-NaN operand, so a clamp written as fminf(fmaxf(.)) would turn a NaN observation into -5.0 -
    a finite wrong value.  All clamps of this library go through clamp_nan (csrc/rlg_device.hpp).  +-Inf clamps to
    +-5 like the reference."""
    from rl_games_amd import ops
    for rows in (4096, 16384):
        layers, g = _net(60, [256, 128], 9, 'elu', seed=6)
        chain = ops.MlpChain(layers, DEV)
        x = torch.randn(rows, 60, generator=g).to(DEV)
        x[3, 5], x[9, 7], x[11, 0] = float('nan'), float('inf'), float('-inf')
        mean = torch.zeros(60, dtype=torch.float64, device=DEV)
        var = torch.ones(60, dtype=torch.float64, device=DEV)
        heads = torch.empty(rows, 9, device=DEV)
        xn = torch.empty(rows, 60, device=DEV)
        acts = [torch.empty(rows, 256, device=DEV), torch.empty(rows, 128, device=DEV)]
        chain.forward(x, heads, act_out=acts, rms=(mean, var), eps=1e-5, xn_out=xn)
        want = torch.clamp((x - mean.float()) / torch.sqrt(var.float() + 1e-5), -5.0, 5.0)
        assert torch.isnan(xn[3, 5]) and torch.isnan(want[3, 5])
        assert xn[9, 7].item() == 5.0 and xn[11, 0].item() == -5.0
        assert not torch.isfinite(heads[3]).any()
        assert torch.isfinite(heads[9]).all() and torch.isfinite(heads[11]).all()
        # the stand-alone normaliser kernel
        out = ops.rms_apply(x, mean, var, 1e-5)
        assert torch.isnan(out[3, 5]) and out[9, 7].item() == 5.0 and out[11, 0].item() == -5.0


# ----------------------------------------------------------------------------- optimiser-written weight planes (round 4)

@pytest.mark.parametrize('in_dim,units,out_dim', [(108, [400, 200, 100], 22), (60, [256, 128, 64], 9), (12, [100, 52], 22)])
def test_adam_step_pack_equals_adam_then_pack(in_dim, units, out_dim):
    """rlg_adam_step_pack (csrc/mlp_chain_bx.hip, adam_pack_kernel): the optimiser step that writes the chain's bf16
    weight planes itself - every thread updates a 4 x 4 block of a weight matrix and stores its rows as forward and its
    columns as backward fragments - against rlg_adam_step followed by rlg_mlp_chain_pack_planes: the same parameters,
    moments and clipped gradients bit for bit, and the same plane bytes (zero padding included), also for a head whose
    22 outputs end inside a group of four and with the skip flag set (nothing changes)."""
    from rl_games_amd import ops
    layers, g = _net(in_dim, units, out_dim, 'elu', seed=7)
    flat = layers[0][0].untyped_storage()
    n = sum(w.numel() + b.numel() for w, b, _ in layers)
    params = torch.empty(0, device=DEV, dtype=torch.float32).set_(flat, 0, (n,))
    assert params.data_ptr() == layers[0][0].data_ptr()
    init = params.clone()
    grads = (0.1 * torch.randn(n, generator=g)).to(DEV)
    m0 = (0.01 * torch.randn(n, generator=g)).to(DEV)
    v0 = (0.001 * torch.rand(n, generator=g)).to(DEV)
    version = [0]
    chain = ops.MlpChain(layers, DEV, weights_version=lambda: version[0])
    res = {}
    for mode in ('pair', 'fused', 'fused_skipped'):
        params.copy_(init)
        g_, m_, v_ = grads.clone(), m0.clone(), v0.clone()
        lr_slots = torch.tensor([3e-4, 3e-4], dtype=torch.float64, device=DEV)
        counter = torch.tensor([3], dtype=torch.int64, device=DEV)
        norm = torch.zeros(ops.grad_norm_blocks(n), dtype=torch.float64, device=DEV)
        ops.grad_sumsq(g_, 1.0, norm, None)
        kl = torch.tensor([0.001], device=DEV)
        stats = torch.zeros(4, device=DEV)
        skip = torch.tensor([1 if mode == 'fused_skipped' else 0], dtype=torch.int32, device=DEV)
        kw = dict(betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, schedule_kind=1, kl=kl, kl_scale=1.0, kl_threshold=0.008,
                  min_lr=1e-6, max_lr=1e-2, lr_multiplier=1.5, stats_out=stats, skip_flag=skip.data_ptr())
        target = chain.adam_pack_target()                 # (first call: packs the buffer in full once)
        if mode == 'pair':
            ops.adam_step(params, g_, m_, v_, norm, 1.0, 0.5, lr_slots, counter, **kw)
            chain._plane_buffer().fill_(0x5a)
            chain.pack_planes(2, params)
        else:
            # poison the in-matrix part only through a full pack of the OLD weights (what the agent's state is)
            chain.pack_planes(2, params)
            ops.adam_step(params, g_, m_, v_, norm, 1.0, 0.5, lr_slots, counter, pack=target, **kw)
        torch.cuda.synchronize()
        res[mode] = (params.clone(), g_, m_, v_, lr_slots.clone(), stats.clone(), chain._plane_buffer().clone())
    for a, b in zip(res['pair'], res['fused']):
        assert torch.equal(a, b)
    assert not torch.equal(res['pair'][0], init)
    # skipped step: parameters, moments untouched, planes = those of the old weights
    assert torch.equal(res['fused_skipped'][0], init) and torch.equal(res['fused_skipped'][2], m0)
    params.copy_(init)
    chain.pack_planes(2, params)
    assert torch.equal(res['fused_skipped'][6], chain._plane_buffer())


@pytest.mark.gpu
@pytest.mark.parametrize('rows', [4096, 1000, 37])
def test_lean_kernels_against_the_pipelined_ones(rows, monkeypatch):
    """The lean 16-row forward / backward / one-launch step (csrc/mlp_chain_lean.hip: the weights as fragments in each wave's
    consumption order, the default of MlpChain below the 64-row kernels' thresholds) against the pipelined exact-product kernels
    they replace (RLG_CHAIN_LEAN=0; their own one-launch step included) - heads, activations, normalised observations, d heads,
    loss partials, dZ and bias partial sums, ragged last tile included.  Round 6: the lean kernels run three fp16 plane
    products per fp32 product - within the split kernels' tolerance of the exact-product kernels (4e-6 of a tensor's scale),
    the lean forward + backward pair and the lean one-launch step bit-identical to each other; a -DRLG_BX_F16=0 build runs
    exact fp32 products in the pipelined kernels' order: everything bit for bit.  Both: within 1e-6 of fp64."""
    from rl_games_amd import ops
    layers, g = _net(108, [400, 200, 100], 22, 'elu', seed=3)
    x = (3 * torch.randn(rows, 108, generator=g) + 1).to(DEV)
    mean = torch.ones(108, dtype=torch.float64, device=DEV)
    var = 9 * torch.ones(108, dtype=torch.float64, device=DEV)
    A = 21
    gg = torch.Generator().manual_seed(5)
    data = dict(actions=torch.randn(rows, A, generator=gg).to(DEV), old_neglogp=torch.randn(rows, generator=gg).to(DEV),
                adv=torch.randn(rows, generator=gg).to(DEV), old_values=torch.randn(rows, generator=gg).to(DEV),
                returns=torch.randn(rows, generator=gg).to(DEV), old_mu=torch.randn(rows, A, generator=gg).to(DEV),
                old_sigma=(0.5 + torch.rand(rows, A, generator=gg)).to(DEV))
    logstd = torch.zeros(A, device=DEV)
    out = {}
    for mode in ('pipe', 'pipe_step', 'lean', 'lean_step'):
        chain = ops.MlpChain(layers, DEV)
        chain._lean = not mode.startswith('pipe')
        heads = torch.full((rows, 22), float('nan'), device=DEV)
        acts = [torch.full((rows, u), float('nan'), device=DEV) for u in (400, 200, 100)]
        xn = torch.full((rows, 108), float('nan'), device=DEV)
        dh = torch.full((rows, 22), float('nan'), device=DEV)
        dzs = [torch.full((rows, u), float('nan'), device=DEV) for u in (400, 200, 100)]
        nb = chain.num_blocks(rows, 1)
        parts = [torch.full((nb * u,), float('nan'), dtype=torch.float64, device=DEV) for u in (400, 200, 100)]
        partials = torch.full((nb, ops.ppo_loss_partials_per_block(A)), float('nan'), dtype=torch.float64, device=DEV)
        om, osg = data['old_mu'].clone(), data['old_sigma'].clone()
        desc = ops.ppo_loss_desc(heads[:, 1:], logstd, heads[:, 0], data['actions'], data['old_neglogp'], data['adv'],
                                 data['old_values'], data['returns'], om, osg, dh[:, 1:], dh[:, 0], partials, 0.2, 2.0, 1e-4,
                                 clip_value=True, smooth=False, bound_kind=1)
        if mode.endswith('_step'):
            assert chain.step(x, heads, acts, dh, dzs, parts, desc, rms=(mean, var), eps=1e-5, xn_out=xn) == (rows <= 16 * 256)
        if not mode.endswith('_step') or rows > 16 * 256:
            chain.forward(x, heads, act_out=acts, rms=(mean, var), eps=1e-5, xn_out=xn)
            chain.backward(dh, acts, dzs, parts, ppo_loss=desc)
        torch.cuda.synchronize()
        out[mode] = [heads, xn, dh, partials, om, osg] + acts + dzs + parts
    exact_lean = ops.chain_split_form()[1] != 'fp16'
    for k, (a, b, c, d) in enumerate(zip(out['pipe'], out['lean'], out['lean_step'], out['pipe_step'])):
        assert torch.isfinite(a.double()).all(), k
        assert torch.equal(a, d), k
        assert torch.equal(b, c), k
        if exact_lean or k == 1:                                   # (k = 1: the normalised observations - no product in them)
            assert torch.equal(a, b), k
        else:
            assert torch.isfinite(b.double()).all(), k
            scale = a.double().abs().max().item()
            assert (a.double() - b.double()).abs().max().item() <= 4e-6 * scale + 1e-30, (k, scale)
    # inference form = training form
    chain = ops.MlpChain(layers, DEV)
    heads_i = torch.empty(rows, 22, device=DEV)
    chain.forward(x, heads_i, rms=(mean, var), eps=1e-5)
    assert torch.equal(heads_i, out['lean'][0])
    a = ((x.double() - mean) / torch.sqrt(var.float() + 1e-5).double()).clamp(-5, 5)
    for (w, b, act) in layers:
        a = torch.addmm(b.double(), a, w.double().t())
        if act == 'elu':
            a = torch.nn.functional.elu(a)
    assert float((out['lean'][0].double() - a).abs().max() / a.abs().max()) < 1e-6


# ----------------------------------------------------------------------------- split-fp16 form (round 6)

def _fp16_form():
    from rl_games_amd import ops
    return ops.chain_split_form()[1] == 'fp16'


@pytest.mark.parametrize('rows,per', [(32768, 64), (16384 + 640, 64), (4096, 64), (4096, 16), (1000, 16)])
def test_dw_fp16_form_with_per_wave_gradient_scales_matches_fp64(rows, per):
    """The weight-gradient launch on three fp16 plane products (MlpDwPlan.launch(maxima=...)): dZ whose magnitude varies
    over 2^-20 .. 1 from one block of `per` rows to the next (64: what the 64-row kernels leave, 16: the lean ones; every wave
    scales by the largest entry over ITS rows), blocks of
    all-zero rows (entry 0), activations under the forward's fixed scales - against fp64, judged like the bf16 form, and
    beside it."""
    from rl_games_amd import ops
    if not _fp16_form():
        pytest.skip('bf16 build')
    g = torch.Generator().manual_seed(rows)
    shapes = [(200, 400), (400, 108), (100, 200), (22, 100)]
    nblk = -(-rows // per)
    jobs, entries = [], torch.zeros(8, max(1024, nblk), device=DEV)
    for k, (No, Mi) in enumerate(shapes):
        block_scale = torch.exp2(-20.0 * torch.rand(nblk, generator=g))
        block_scale[::7] = 0.0                                               # whole blocks of zero gradient (masked rows)
        dz = torch.randn(rows, No, generator=g) * block_scale.repeat_interleave(per)[:rows, None] * 3e-4
        x = torch.nn.functional.elu(torch.randn(rows, Mi, generator=g))
        if k == 1:
            x = x.clamp(-5.0, 5.0)                                           # "normalised observations"
        dz, x = dz.to(DEV), x.to(DEV)
        pad = torch.zeros(nblk * per, No, device=DEV)
        pad[:rows] = dz.abs()
        entries[k, :nblk] = pad.view(nblk, per * No).max(dim=1).values
        jobs.append((dz, x, torch.full((No, Mi), float('nan'), device=DEV)))
    plan = ops.MlpDwPlan(shapes, rows, DEV)
    xscale = [ops.SPLIT_SCALE_HIDDEN, ops.SPLIT_SCALE_OBS_NORM, ops.SPLIT_SCALE_HIDDEN, ops.SPLIT_SCALE_HIDDEN]
    plan.launch(jobs, maxima=(entries, [0, 1, 2, 3], xscale, per))
    f16 = [j[2].clone() for j in jobs]
    plan.launch(jobs)                                                        # the bf16 form on the same operands
    for (dz, x, bf), got in zip(jobs, f16):
        assert torch.isfinite(got).all()
        t64 = dz.double().t() @ x.double()
        scale = dz.double().abs().t() @ x.double().abs()
        e16 = ((got.double() - t64).abs() / scale.clamp_min(1e-300))
        eb = ((bf.double() - t64).abs() / scale.clamp_min(1e-300))
        lib = (((dz.t() @ x).double() - t64).abs() / scale.clamp_min(1e-300))
        assert e16.max() <= max(2.0 * eb.max().item(), lib.max().item()), (e16.max(), eb.max(), lib.max())
        assert e16.pow(2).mean().sqrt() <= 1.5 * eb.pow(2).mean().sqrt() + 1e-10


@pytest.mark.parametrize('rows', [16384, 4096])
def test_split_fp16_backward_scales_gradient_rows_by_their_own_maxima(rows):
    """The fp16 backward splits the d heads tile ROW BY ROW: a row of gradients a million times smaller than its tile
    neighbours gives the same dZ bits as in a tile of rows like itself, and the gradient maxima it leaves for the
    weight-gradient launch are the per-workgroup maxima (64 rows; 16 for the lean kernels of a 4,096-row launch) of what it wrote."""
    from rl_games_amd import ops
    if not _fp16_form():
        pytest.skip('bf16 build')
    layers, g = _net(60, [256, 128], 9, 'elu', seed=31)
    chain = ops.MlpChain(layers, DEV)
    assert chain.split_products(rows, 1) == (rows >= 8192) and chain.lean_used(rows, 1) == (rows < 8192)
    x = torch.randn(rows, 60, generator=g).to(DEV)
    heads = torch.empty(rows, 9, device=DEV)
    acts = [torch.empty(rows, 256, device=DEV), torch.empty(rows, 128, device=DEV)]
    chain.forward(x, heads, act_out=acts)
    d = (1e-4 * torch.randn(rows, 9, generator=g)).to(DEV)

    def run(d_heads):
        dzs = [torch.full((rows, 256), float('nan'), device=DEV), torch.full((rows, 128), float('nan'), device=DEV)]
        nb = chain.num_blocks(rows, 1)
        parts = [torch.empty(nb * 256, dtype=torch.float64, device=DEV), torch.empty(nb * 128, dtype=torch.float64, device=DEV)]
        chain.backward(d_heads, acts, dzs, parts)
        return dzs
    small = d.clone()
    small[::3] *= 2.0 ** -20
    mixed = run(small)
    maxima = chain.gradient_maxima(rows)
    assert maxima is not None
    per = chain.maxima_rows_per_entry
    assert per == (64 if rows >= 8192 else 16)
    nblk = rows // per
    for l, dz in enumerate(mixed):
        assert torch.equal(maxima[l, :nblk], dz.abs().view(nblk, -1).max(dim=1).values), l
    assert torch.equal(maxima[2, :nblk], small.abs().view(nblk, -1).max(dim=1).values)
    alone = run(d * 2.0 ** -20)
    for m, a_ in zip(mixed, alone):
        assert torch.equal(m[::3], a_[::3])


def test_split_fp16_range_limits_end_in_non_finite_values_never_in_wrong_ones():
    """The fixed scales of the fp16 form hold |weight| < 1023 and |hidden activation| < 4094: beyond them the planes are
    Inf and what is computed from them is Inf / NaN - in the rows concerned (an activation) or everywhere (a weight) -
    while everything inside the range is what the exact-product kernels give."""
    from rl_games_amd import ops
    if not _fp16_form():
        pytest.skip('bf16 build')
    rows = 16384
    layers, g = _net(60, [256, 128], 9, 'elu', seed=32)
    x = torch.randn(rows, 60, generator=g).to(DEV)

    def run(layers, x, split):
        chain = ops.MlpChain(layers, DEV)
        heads = torch.empty(rows, 9, device=DEV)
        acts = [torch.empty(rows, 256, device=DEV), torch.empty(rows, 128, device=DEV)]
        chain.forward(x, heads, act_out=acts, split_products=None if split else False)
        return acts, heads
    # a row whose first-layer activations reach 1e4 (raw observations have no bound; the activations they produce do)
    big = x.clone()
    big[5] *= 3e4
    acts, heads = run(layers, big, True)
    ex_acts, ex_heads = run(layers, big, False)
    assert ex_acts[0][5].abs().max() > 4094 and torch.isfinite(ex_heads).all()
    assert not torch.isfinite(heads[5]).all()
    keep = torch.ones(rows, dtype=torch.bool, device=DEV)
    keep[5] = False
    assert torch.isfinite(heads[keep]).all()
    assert (heads[keep] - ex_heads[keep]).abs().max() <= 4e-6 * ex_heads[keep].abs().max()
    # a weight of 2,000
    heavy = [(w.clone(), b.clone(), a) for w, b, a in layers]
    heavy[1][0][3, 7] = 2000.0
    acts, heads = run(heavy, x, True)
    assert not torch.isfinite(acts[1][:, 3]).any()
