"""CPU: pins the GAE oracle (torch restatement + plain C) to the reference's outputs and to
the reference's own fp64 ground truth (tests/test_triton_gae.py:20-61)."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import c_oracle, ppo_oracle
from oracle.seeded_inputs import gae_inputs


def _digest(t):
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()


def _case_inputs(case):
    if 'inputs' in case:
        return case['inputs']
    inp = gae_inputs(*case['shape'], seed=case['seed'], p_done=case.get('p_done', 0.15))
    assert [_digest(t) for t in inp] == case['inputs_sha256'], 'seeded inputs drifted'
    return inp


def test_oracle_matches_reference_outputs(golden):
    g = golden('gae.pt')
    for case in g['cases']:
        inp = _case_inputs(case)
        ref = case['advs']
        out = ppo_oracle.gae_scan(*inp, case['gamma'], case['tau'])
        assert torch.equal(out, ref), case['shape']
        c = c_oracle.gae_f32_scan(*[t.numpy() for t in inp], case['gamma'], case['tau'])
        assert np.array_equal(c, ref.numpy()), case['shape']


def test_oracle_matches_reference_fp64_ground_truth(golden):
    """Same tolerance as the reference's own test (atol 1e-5, test_triton_gae.py:61)."""
    g = golden('gae.pt')
    seen = 0
    for case in g['cases']:
        if 'advs_f64' not in case:
            continue
        inp = _case_inputs(case)
        out = ppo_oracle.gae_scan(*inp, case['gamma'], case['tau'])
        assert torch.allclose(out, case['advs_f64'], atol=1e-5)
        mine = ppo_oracle.gae_scalar_f64(*inp, case['gamma'], case['tau'])
        assert torch.equal(mine, case['advs_f64'])
        f64 = c_oracle.gae_f64_reference(*[t.numpy() for t in inp], case['gamma'], case['tau'])
        assert np.allclose(f64, case['advs_f64'].double().numpy(), atol=1e-6)
        seen += 1
    assert seen >= 4


@pytest.mark.parametrize('idx', [1, 2])
def test_oracle_full_size_digests(golden, idx):
    """BASELINE.json shapes (digest fixtures): (16,4096,1) and (32,8192,1) here; the 65,536-env
    case runs in the GPU suite next to the kernel."""
    case = golden('gae.pt')['big'][idx]
    inp = _case_inputs(case)
    out = torch.from_numpy(c_oracle.gae_f32_scan(*[t.numpy() for t in inp], case['gamma'], case['tau']))
    assert _digest(out) == case['advs_sha256']
    ret, adv = c_oracle.returns_and_advantages(out.numpy(), inp[1].numpy())
    assert _digest(torch.from_numpy(ret)) == case['returns_sha256']
    assert _digest(torch.from_numpy(adv)) == case['advantages_sha256']
    assert torch.equal(out[::7, ::4099, 0], case['advs_probe'])


def test_returns_minus_values_is_not_gae_output():
    """SURVEY 8a' pitfall 1: (A+v)-v != A in fp32 - the fused kernel must round twice."""
    r, v, d, lv, ld = gae_inputs(32, 512, 1, seed=3)
    v = v * 10
    a = ppo_oracle.gae_scan(r, v, d, lv, ld, 0.99, 0.95)
    ret = ppo_oracle.returns_from_advantages(a, v)
    assert not torch.equal(ret - v, a)


def test_flatten_env_major_index_map():
    x = torch.arange(3 * 5 * 2).reshape(3, 5, 2)
    flat = ppo_oracle.flatten_env_major(x)
    for env in range(5):
        for t in range(3):
            assert torch.equal(flat[env * 3 + t], x[t, env])
