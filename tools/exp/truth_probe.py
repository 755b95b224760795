"""The 320-step job against the oracle AND against the oracle's fp64 evaluation ("truth"), per mini-epoch (round 6):

    python tools/exp/truth_probe.py [envs minibatch]          (RLG_HIP_LIB selects a variant build of the library)

Prints, per scalar and mini-epoch, max |agent - oracle|, max |agent - truth|, max |oracle - truth|: is the agent as close to
the exact-arithmetic trajectory of the algorithm as the reference's own fp32 arithmetic is?  (-> profiles/r6_truth_probe.txt;
the criterion of tests/test_headline_gpu.py::test_whole_epoch_of_320_steps_matches_oracle since round 6.)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_headline_gpu as T  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    MB = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    NMB, ME = N * 32 // MB, 5
    params, agent, caps, res = T._epoch_deviation_rows(N, MB)
    rows = agent._mb_scalars[:ME * NMB].cpu()
    torch.set_num_threads(T._oracle_threads())
    oracle = T._oracle_for(params, caps[0], N, 108, 21)
    ref = oracle.update(caps[0]['batch'])
    truth = T._truth_for(params, caps[0], N, 108, 21)
    tru = truth.update(T._batch64(caps[0]['batch']))
    cols = {'a_loss': 0, 'c_loss': 1, 'entropy': 2, 'b_loss': 3, 'kl': 4}
    ref_rows = torch.stack([torch.stack([r[k].reshape(()).double() for k in cols]) for r in ref])
    tru_rows = torch.stack([torch.stack([r[k].reshape(()).double() for k in cols]) for r in tru])
    a = rows[:, :5].double()
    print(f'envs {N} minibatch {MB}: lean {agent._engine.chain.lean_used(MB, 0)} split {bool(agent._engine.chain.split_products(MB, 0))} '
          f'lib {os.environ.get("RLG_HIP_LIB", "product")}')
    lr_agent = None
    mism = next((k for k in range(len(ref)) if ref[k]['lr'] != tru[k]['lr']), None)
    print('first lr mismatch oracle vs truth:', mism)
    for key, c in cols.items():
        scale = ref_rows[:, c].abs().max().item()
        f = lambda x, y: ' '.join(f'{v:.1e}' for v in (x[:, c] - y[:, c]).abs().reshape(ME, NMB).max(1).values.tolist())
        print(f'{key:8s} scale {scale:.2e}  agent-oracle {f(a, ref_rows)} | agent-truth {f(a, tru_rows)} | oracle-truth {f(ref_rows, tru_rows)}')


if __name__ == '__main__':
    main()
