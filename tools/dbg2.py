import sys, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import ppo_oracle as O
from rl_games_amd import ops
import test_ops_gpu as T
DEV='cuda:0'
mb, A = 4096, 21
mu, logstd, values, batch = T._loss_inputs(mb, A, seed=mb + A)
hp = {'e_clip': 0.2, 'critic_coef': 2.0, 'entropy_coef': 0.01, 'bounds_loss_coef': 0.01,
      'clip_value': True, 'use_smooth_clamp': True, 'bound_loss_type': 'regularisation'}
ref = O.distribution_loss_and_grads(mu, logstd, values, batch, hp, None)
to64 = lambda t: t.double() if torch.is_floating_point(t) else t
truth = O.distribution_loss_and_grads(to64(mu), to64(logstd), to64(values), {k: to64(v) for k, v in batch.items()}, hp, None)
d = lambda t: t.contiguous().to(DEV)
old_mu, old_sigma = d(batch['mu']), d(batch['sigma'])
d_mu = torch.empty(mb, A, device=DEV); d_val = torch.empty(mb, device=DEV)
nb = ops.ppo_loss_blocks(mb)
partials = torch.empty(nb, 6 + A, dtype=torch.float64, device=DEV)
ops.ppo_loss_fused(d(mu), d(logstd), d(values.reshape(-1)), d(batch['actions']), d(batch['old_logp_actions']), d(batch['advantages']),
                   d(batch['old_values'].reshape(-1)), d(batch['returns'].reshape(-1)), old_mu, old_sigma, d_mu, d_val, partials,
                   0.2, 2.0, 0.01, True, True, 2, True)
for name, got, r32, r64 in (('d_mu', d_mu.cpu(), ref['d_mu'], truth['d_mu']), ('d_val', d_val.cpu(), ref['d_values'].reshape(-1), truth['d_values'].reshape(-1))):
    diff = (got - r32).abs()
    rel = diff / r32.abs().clamp_min(1e-30)
    bad = diff > (2e-3 * r32.abs() + 1e-6 / mb)
    print(name, 'max abs', diff.max().item(), 'max |ref|', r32.abs().max().item(), 'bad', bad.sum().item(), 'of', got.numel())
    idx = bad.nonzero()[:8]
    for ix in idx.tolist():
        ix = tuple(ix)
        print('   ', ix, got[ix].item(), r32[ix].item(), r64[ix].item())
    rows_bad = bad.reshape(mb, -1).any(1).nonzero().reshape(-1); globals()["RB_"+name] = rows_bad
    print('   bad rows', rows_bad[:10].tolist(), len(rows_bad))
    # error vs fp64 truth
    e_k = (got.double() - r64).abs().max().item(); e_o = (r32.double() - r64).abs().max().item()
    print('   max err vs f64 truth: kernel', e_k, 'oracle32', e_o)
with torch.no_grad():
    sigma = torch.exp(logstd)
    nlp = O.neglogp(batch['actions'], mu, mu * 0 + sigma, mu * 0 + logstd)
    ratio = torch.exp(batch['old_logp_actions'] - nlp)
    rb = bad.reshape(mb, -1).any(1).nonzero().reshape(-1) if False else None
print('ratio stats', ratio.min().item(), ratio.max().item())
rbm = globals()['RB_d_mu']
print('bad rows ratio', ratio[rbm][:10].tolist(), 'adv', batch['advantages'][rbm][:10].tolist(), 'nlp', nlp[rbm][:10].tolist())
