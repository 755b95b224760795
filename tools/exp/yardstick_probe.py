"""GPU: what should the parity yardstick of tests/test_headline_gpu.py::test_whole_epoch_of_320_steps_matches_oracle be at a
rank's shape (8,192 envs x 32, 4,096-row minibatches: exact products, lean 16-row kernels)?  Prints, per mini-epoch and
scalar, the deviation envelope from the oracle of
  * the agent (fused kernels),
  * the SAME agent on its per-layer engine (`fused_mlp: False`: library GEMMs, stand-alone loss kernel) - a second
    fp32 implementation on the device,
  * twins of the oracle under first-layer GEMM-order noise (1e-6) and under one-ulp-class noise on the rows' neglogp
    (the sum over the actions in another order),
plus the mini-epochs' mean KL and the learning-rate range (how hard the run drives the clip)."""
import copy
import importlib.util
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location('headline', os.path.join(ROOT, 'tests', 'test_headline_gpu.py'))
T = importlib.util.module_from_spec(spec)
spec.loader.exec_module(T)
from rl_games_amd import configs  # noqa: E402
from rl_games_amd.agent import A2CAgent  # noqa: E402

N, MB = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8192, 4096)
NMB, ME = 64, 5
COLS = ('a_loss', 'c_loss', 'entropy', 'b_loss', 'kl')


def run_agent(**over):
    over.setdefault('hip_graphs', True)
    params = configs.humanoid_65536(num_actors=N, minibatch_size=MB, **over)
    torch.manual_seed(5)
    agent = A2CAgent('epoch', copy.deepcopy(params))
    agent.init_tensors()
    agent.obs = agent.env_reset()
    caps = T._capture_rollout(agent)
    agent._eager_epochs = 1
    agent.update_epoch()
    agent.train_epoch()
    return params, agent, caps, agent._mb_scalars[:ME * NMB].cpu()


def fmt(d):
    return '  '.join(f'{k} ' + ' '.join(f'{v:.1e}' for v in d[k]) for k in COLS)


def nlp_noise(oracle, seed, level):
    gen = torch.Generator().manual_seed(seed)

    def hook(nlp):
        sign = torch.randint(0, 2, nlp.shape, generator=gen).to(nlp.dtype).mul_(2.0).sub_(1.0)
        return nlp * (1.0 + level * sign)
    oracle.nlp_hook = hook
    return oracle


torch.set_num_threads(T._oracle_threads())
params, agent, caps, rows = run_agent()
t0 = time.time()
oracle = T._oracle_for(params, caps[0], N, 108, 21)
ref = oracle.update(caps[0]['batch'])
print(f'oracle update {time.time() - t0:.1f} s; lean {agent._lean_chain() is not None}; split {bool(agent._engine.chain.split_products(MB, 0))}')
kls = torch.stack([r['kl'].reshape(()) for r in ref]).reshape(ME, NMB).mean(1)
lrs = [r['lr'] for r in ref]
print('oracle mean KL per mini-epoch', [f'{float(k):.4f}' for k in kls], 'lr min/max', min(lrs), max(lrs))
from oracle import ppo_oracle as O  # noqa: E402
cfg = params['config']
lr, traj = float(cfg['learning_rate']), []
for k in range(ME * NMB):
    traj.append(lr)
    lr = O.adaptive_lr(lr, float(rows[k, 4]), cfg['kl_threshold'], cfg.get('min_lr', 1e-6), cfg.get('max_lr', 1e-2),
                       cfg.get('lr_multiplier', 1.5))
print('agent lr trajectory: first mismatch with the oracle at step', next((k for k in range(ME * NMB) if traj[k] != lrs[k]), None))
print('AGENT (fused)      ', fmt(T._deviation_per_mini_epoch(rows, ref, NMB, ME)))
def twin_rows(make):
    tw = make(T._oracle_for(params, caps[0], N, 108, 21))
    r = tw.update(caps[0]['batch'])
    return torch.stack([torch.stack([x[k].reshape(()).float() for k in COLS]) for x in r]), [x['lr'] for x in r]


for name, make in (('gemm 1e-6 s11', lambda o: T._gemm_order_noise(o, 11)), ('gemm 1e-6 s12', lambda o: T._gemm_order_noise(o, 12)),
                   ('nlp 6e-8 s1', lambda o: nlp_noise(o, 1, 6e-8)), ('nlp 6e-8 s2', lambda o: nlp_noise(o, 2, 6e-8)),
                   ('nlp 2e-7 s1', lambda o: nlp_noise(o, 1, 2e-7)), ('nlp 2e-7 s2', lambda o: nlp_noise(o, 2, 2e-7)),
                   ('gemm+nlp 6e-8', lambda o: nlp_noise(T._gemm_order_noise(o, 13), 3, 6e-8))):
    tr, tl = twin_rows(make)
    mism = next((k for k in range(len(lrs)) if tl[k] != lrs[k]), None)
    print(f'TWIN {name:14s}', fmt(T._deviation_per_mini_epoch(tr, ref, NMB, ME)), ' first lr mismatch', mism)


# the same job on the per-layer engine, against the oracle on ITS rollout (same seeds: the rollouts agree unless the
# rollout kernels differ)
p2, a2, c2, rows2 = run_agent(fused_mlp=False, hip_graphs=False)      # (library GEMMs are not capturable: eager)
o2 = T._oracle_for(p2, c2[0], N, 108, 21)
ref2 = o2.update(c2[0]['batch'])
print('AGENT (per-layer)  ', fmt(T._deviation_per_mini_epoch(rows2, ref2, NMB, ME)), ' chain', a2._engine.chain is not None)
same_rollout = all(torch.equal(caps[0]['batch'][k], c2[0]['batch'][k]) for k in caps[0]['batch'])
print('per-layer run played the same rollout:', same_rollout)


