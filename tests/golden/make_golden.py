"""Generates the golden fixtures under tests/golden/ from the REAL reference.

Run in the build container only (needs /root/reference; the GPU box does not have it):

    python tests/golden/make_golden.py

Every fixture stores seeded inputs and the outputs of the reference's own functions, so the
oracle (oracle/ppo_oracle.py, oracle/gae_oracle.c) and the HIP kernels can be checked
against the reference without the reference being present.  Large cases store SHA-256
digests of the raw little-endian bytes instead of the tensors.
"""
import hashlib
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = os.environ.get('RLG_REFERENCE', '/root/reference')
sys.path.insert(0, ROOT)
sys.path.append(REFERENCE)
os.environ['RLG_NO_TRITON'] = '1'

from oracle.seeded_inputs import gae_inputs  # noqa: E402


def digest(t):
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()


def make_gae():
    from rl_games.triton_kernels.gae_kernel import _pytorch_gae, compute_gae
    sys.path.append(os.path.join(REFERENCE, 'tests'))
    from test_triton_gae import reference_gae  # the reference's own fp64 ground truth

    cases = []
    # shapes of tests/test_triton_gae.py:55-61 (CPU) and :72-79 (GPU), plus V>1 / odd sizes
    small = [(8, 4, 1), (16, 6, 3), (8, 16, 1), (36, 64, 3), (200, 128, 1), (32, 257, 1),
             (16, 130, 1), (5, 7, 2), (64, 70, 1), (12, 8, 2)]
    for shape in small:
        for gamma, tau in ((0.99, 0.95), (1.0, 1.0)):
            inp = gae_inputs(*shape, seed=0)
            out = _pytorch_gae(*inp, gamma, tau)
            assert torch.equal(out, compute_gae(*inp, gamma, tau))
            # inputs regenerate from the seed (oracle/seeded_inputs.py); keep their digests
            case = {'shape': shape, 'gamma': gamma, 'tau': tau, 'seed': 0,
                    'inputs_sha256': [digest(t) for t in inp], 'advs': out.clone()}
            if shape[0] * shape[1] * shape[2] <= 2000:
                case['advs_f64'] = reference_gae(*inp, gamma, tau)
            cases.append(case)
    # all-done / no-done edges (tests/test_triton_gae.py:101-110)
    for fill in (0.0, 1.0):
        r, v, d, lv, ld = gae_inputs(10, 4, 1, seed=0)
        d = torch.full_like(d, fill)
        ld = torch.full_like(ld, fill)
        cases.append({'shape': (10, 4, 1), 'gamma': 0.99, 'tau': 0.95, 'seed': 0, 'fill': fill,
                      'inputs': [r, v, d, lv, ld], 'advs': _pytorch_gae(r, v, d, lv, ld, 0.99, 0.95)})
    # BASELINE.json shapes: digests only
    big = []
    for shape in ((32, 65536, 1), (16, 4096, 1), (32, 8192, 1)):
        inp = gae_inputs(*shape, seed=0, p_done=0.05)
        out = _pytorch_gae(*inp, 0.99, 0.95)
        ret = out + inp[1]                       # a2c_common.py:1060
        adv = ret - inp[1]                       # a2c_common.py:1598
        big.append({'shape': shape, 'gamma': 0.99, 'tau': 0.95, 'seed': 0, 'p_done': 0.05,
                    'inputs_sha256': [digest(t) for t in inp], 'advs_sha256': digest(out),
                    'returns_sha256': digest(ret), 'advantages_sha256': digest(adv),
                    'advs_probe': out[::7, ::4099, 0].clone()})
    torch.save({'cases': cases, 'big': big, 'torch': torch.__version__},
               os.path.join(HERE, 'gae.pt'))
    print('gae.pt:', len(cases), 'cases +', len(big), 'digest cases')


SECTIONS = {'gae': make_gae}

if __name__ == '__main__':
    only = sys.argv[1:] or list(SECTIONS)
    for name in only:
        SECTIONS[name]()
