"""GPU parity of the GAE kernels (through the C ABI) against the oracle, the golden fixtures
generated from the reference, and size-independent properties at BASELINE.json's sizes.
fp32 results are required to be BIT-EXACT: the kernels evaluate the same op chain."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import c_oracle, ppo_oracle
from oracle.seeded_inputs import gae_inputs

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _digest(t):
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()


def _gpu(ts):
    return [t.to(DEV) for t in ts]


def _envmajor_views(r, v, d, lv, ld):
    """Rollout-buffer layout: physical [N,H] storage exposed as [H,N,(1)] views, u8 dones."""
    rp = r[..., 0].t().contiguous().to(DEV)
    vp = v[..., 0].t().contiguous().to(DEV)
    dp = d.t().contiguous().to(torch.uint8).to(DEV)
    return (rp.t().unsqueeze(2), vp.t().unsqueeze(2), dp.t(), lv.to(DEV),
            ld.to(torch.uint8).to(DEV)), (rp, vp, dp)


def test_strided_matches_reference_goldens(golden):
    from rl_games_amd.gae import compute_gae
    for case in golden('gae.pt')['cases']:
        inp = case['inputs'] if 'inputs' in case else gae_inputs(*case['shape'], seed=case['seed'])
        out = compute_gae(*_gpu(inp), case['gamma'], case['tau'])
        assert out.shape == case['advs'].shape
        assert torch.equal(out.cpu(), case['advs']), case['shape']


def test_u8_and_bool_dones_match_float_dones():
    from rl_games_amd.gae import compute_gae
    r, v, d, lv, ld = gae_inputs(24, 300, 2, seed=5)
    ref = ppo_oracle.gae_scan(r, v, d, lv, ld, 0.99, 0.95)
    for dt in (torch.uint8, torch.bool):
        out = compute_gae(r.to(DEV), v.to(DEV), d.to(dt).to(DEV), lv.to(DEV), ld.to(dt).to(DEV),
                          0.99, 0.95)
        assert torch.equal(out.cpu(), ref)
    # mixed: float mb_dones, u8 last_dones
    out = compute_gae(r.to(DEV), v.to(DEV), d.to(DEV), lv.to(DEV), ld.to(torch.uint8).to(DEV),
                      0.99, 0.95)
    assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize('shape', [(32, 257, 1), (16, 130, 1), (64, 70, 1), (8, 64, 1),
                                   (4, 1, 1), (20, 63, 1), (48, 200, 1)])
@pytest.mark.parametrize('gamma,tau', [(0.99, 0.95), (1.0, 1.0)])
def test_envmajor_raw_path_bit_exact(shape, gamma, tau):
    """compute_gae on rollout-buffer views takes the LDS-tiled kernel; ragged last tile."""
    from rl_games_amd.gae import compute_gae
    r, v, d, lv, ld = gae_inputs(*shape, seed=11)
    ref = c_oracle.gae_f32_scan(r.numpy(), v.numpy(), d.numpy(), lv.numpy(), ld.numpy(), gamma, tau)
    views, _ = _envmajor_views(r, v, d, lv, ld)
    out = compute_gae(*views, gamma, tau)
    assert out.shape == r.shape
    assert np.array_equal(out.cpu().numpy(), ref)


def test_view_inputs_like_reference_test():
    """tests/test_triton_gae.py:82-98: last-dim slices and a transposed dones view."""
    from rl_games_amd.gae import compute_gae
    H, N, V = 12, 8, 2
    r, v, d, lv, ld = gae_inputs(H, N, V + 1, seed=0)
    ref = ppo_oracle.gae_scan(r[:, :, :V].contiguous(), v[:, :, :V].contiguous(), d,
                              lv[:, :V].contiguous(), ld, 0.99, 0.95)
    rg, vg, dg, lvg, ldg = _gpu((r, v, d, lv, ld))
    rg, vg, lvg = rg[:, :, :V], vg[:, :, :V], lvg[:, :V]
    dg = dg.t().contiguous().t()
    assert not rg.is_contiguous()
    out = compute_gae(rg, vg, dg, lvg, ldg, 0.99, 0.95)
    assert torch.equal(out.cpu(), ref)


def test_all_done_and_no_done_edges():
    from rl_games_amd.gae import compute_gae
    r, v, d, lv, ld = gae_inputs(10, 4, 1, seed=0)
    for fill in (0.0, 1.0):
        dd, ldd = torch.full_like(d, fill), torch.full_like(ld, fill)
        ref = ppo_oracle.gae_scan(r, v, dd, lv, ldd, 0.99, 0.95)
        out = compute_gae(*_gpu((r, v, dd, lv, ldd)), 0.99, 0.95)
        assert torch.equal(out.cpu(), ref)
        views, _ = _envmajor_views(r.repeat(1, 20, 1), v.repeat(1, 20, 1), dd.repeat(1, 20),
                                   lv.repeat(20, 1), ldd.repeat(20))
        out2 = compute_gae(*views, 0.99, 0.95)
        assert torch.equal(out2.cpu()[:, :4], ref)
    # every step terminal: A_t = r_t - v_t exactly
    dd, ldd = torch.ones_like(d), torch.ones_like(ld)
    out = compute_gae(*_gpu((r, v, dd, lv, ldd)), 0.99, 0.95)
    assert torch.equal(out.cpu(), (r + 0.99 * v * 0.0) - v)


@pytest.mark.parametrize('shape', [(32, 1000, 1), (16, 4096, 1), (8, 130, 1), (64, 129, 1)])
def test_fused_returns_advantages_and_moments(shape):
    from rl_games_amd.gae import gae_returns_advantages
    r, v, d, lv, ld = gae_inputs(*shape, seed=2, p_done=0.05)
    v = v * 3 + 1
    advs = c_oracle.gae_f32_scan(r.numpy(), v.numpy(), d.numpy(), lv.numpy(), ld.numpy(), 0.99, 0.95)
    ret_ref, adv_ref = c_oracle.returns_and_advantages(advs, v.numpy())
    _, (rp, vp, dp) = _envmajor_views(r, v, d, lv, ld)
    ret, adv, part = gae_returns_advantages(rp, vp, dp, lv[:, 0].contiguous().to(DEV),
                                            ld.to(torch.uint8).to(DEV), 0.99, 0.95)
    # physical layout is [N,H]: compare against the time-major oracle transposed
    assert np.array_equal(ret.cpu().numpy(), ret_ref[..., 0].T)
    assert np.array_equal(adv.cpu().numpy(), adv_ref[..., 0].T)
    m = part.sum(0).cpu().numpy()
    for k, x in enumerate((adv_ref, v.numpy(), ret_ref)):
        x = x.astype(np.float64)
        assert np.isclose(m[2 * k], x.sum(), rtol=1e-12, atol=1e-9)
        assert np.isclose(m[2 * k + 1], (x * x).sum(), rtol=1e-12)


def test_baseline_size_digests(golden):
    """65,536 envs x 32 (BASELINE.json config #3): bit-exact against digests of the reference's
    _pytorch_gae output, through both kernels."""
    from rl_games_amd.gae import compute_gae, gae_returns_advantages
    for case in golden('gae.pt')['big']:
        inp = gae_inputs(*case['shape'], seed=case['seed'], p_done=case['p_done'])
        assert [_digest(t) for t in inp] == case['inputs_sha256']
        out = compute_gae(*_gpu(inp), case['gamma'], case['tau'])
        assert _digest(out) == case['advs_sha256']
        views, (rp, vp, dp) = _envmajor_views(*inp)
        out2 = compute_gae(*views, case['gamma'], case['tau'])
        assert _digest(out2) == case['advs_sha256']
        ret, adv, _ = gae_returns_advantages(rp, vp, dp, inp[3][:, 0].contiguous().to(DEV),
                                             inp[4].to(torch.uint8).to(DEV), case['gamma'], case['tau'])
        assert _digest(ret.t().unsqueeze(2)) == case['returns_sha256']
        assert _digest(adv.t().unsqueeze(2)) == case['advantages_sha256']


def test_properties_at_full_size():
    """Size-independent properties at 65,536 x 32: (i) with no terminals and gamma=tau=1 the
    scan telescopes: A_t = sum_{s>=t} r_s + v_last - v_t (checked in fp64 tolerance);
    (ii) scaling all of r, v, v_last by 2 scales A by exactly 2 (power-of-two linearity is
    exact in fp32); (iii) a done at t+1 cuts the dependence on everything after t."""
    from rl_games_amd.gae import compute_gae
    H, N = 32, 65536
    g = torch.Generator().manual_seed(9)
    r = torch.randn(H, N, 1, generator=g).to(DEV)
    v = torch.randn(H, N, 1, generator=g).to(DEV)
    lv = torch.randn(N, 1, generator=g).to(DEV)
    zeros = torch.zeros(H, N, dtype=torch.uint8, device=DEV)
    lz = torch.zeros(N, dtype=torch.uint8, device=DEV)
    a = compute_gae(r, v, zeros, lv, lz, 1.0, 1.0)
    tele = torch.flip(torch.cumsum(torch.flip(r.double(), [0]), 0), [0]) + lv.double() - v.double()
    assert torch.allclose(a.double(), tele, atol=2e-5)
    d = (torch.rand(H, N, generator=g) < 0.05).to(torch.uint8).to(DEV)
    ld = (torch.rand(N, generator=g) < 0.05).to(torch.uint8).to(DEV)
    a1 = compute_gae(r, v, d, lv, ld, 0.99, 0.95)
    a2 = compute_gae(r * 2, v * 2, d, lv * 2, ld, 0.99, 0.95)
    assert torch.equal(a2, a1 * 2)
    # (iii) perturb everything at t >= 20; rows t < 19 of envs with done[20] are unchanged
    r2, v2 = r.clone(), v.clone()
    r2[20:] += 1.0
    v2[20:] -= 1.0
    a3 = compute_gae(r2, v2, d, lv + 5, ld, 0.99, 0.95)
    cut = d[20].bool()
    assert cut.any()
    assert torch.equal(a3[:19, cut], a1[:19, cut])
