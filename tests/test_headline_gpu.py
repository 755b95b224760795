"""GPU parity at the BENCHMARKED shapes (BASELINE.json configs[1] and configs[2]), through the same
code path bench.py times: 32,768-row minibatches on the fused MLP kernels, one HIP graph per
mini-epoch, dataset preparation at 65,536 x 32, and the rollout head fed with the reference's own
recorded rollout.

Oracle = oracle/ppo_epoch_oracle.OracleAgent (CPU restatement of a2c_common.py:1586-1660 /
a2c_continuous.py:136-234, pinned to the reference by tests/test_oracle_epoch.py and
tests/test_vs_reference_cpu.py) on the SAME rollout tensors.  Tolerances: rtol 1e-5 (north_star)
plus an absolute floor for scalars that are means of +-O(1) terms - a_loss of the first pass is
-mean(normalised advantages) ~ 5e-3 over 32,768 rows, where one fp32 ulp of the O(1) summands is
6e-8 and the summation order (GPU block tree vs ATen's) alone moves the mean by ~1e-7.
"""
import copy

import numpy as np
import pytest
import torch

from oracle import ppo_oracle as O
from oracle.ppo_epoch_oracle import OracleAgent
from rl_games_amd.synthetic_env import SyntheticTensorEnv

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'

RTOL = 1e-5
ATOL = {'a_loss': 2e-6, 'c_loss': 2e-6, 'entropy': 2e-6, 'b_loss': 1e-7, 'kl': 2e-7}


def _make_agent(cap, **over):
    from rl_games_amd.agent import A2CAgent
    params = copy.deepcopy(cap['params'])
    params['config'].update(device=DEV, **over)
    env = SyntheticTensorEnv(cap['env']['num_envs'], cap['env']['obs_dim'], cap['env']['act_dim'],
                             device=DEV, seed=cap['env']['seed'])
    params['config']['vec_env'] = env
    params['config']['env_info'] = env.get_env_info()
    agent = A2CAgent('test', params)
    agent.init_tensors()
    return agent


def _capture_rollout(agent):
    """Wraps play_steps: keeps a CPU copy of every rollout batch and the model state it was played with."""
    caps = []
    orig = agent.play_steps

    def play():
        b = orig()
        caps.append({'batch': {k: v.detach().cpu().clone() for k, v in b.items() if isinstance(v, torch.Tensor)},
                     'state': {k: v.detach().cpu().clone() for k, v in agent.model.state_dict().items()}})
        return b
    agent.play_steps = play
    return caps


def _check_epoch(res, ref, nmb, mini_epochs, bounds=True):
    a_losses, c_losses, b_losses, entropies, kls = res[4], res[5], res[6], res[7], res[8]
    assert len(a_losses) == len(ref) == nmb * mini_epochs
    got = {'a_loss': torch.stack(a_losses).cpu(), 'c_loss': torch.stack(c_losses).cpu(),
           'entropy': torch.stack(entropies).cpu()}
    if bounds:
        got['b_loss'] = torch.stack(b_losses).cpu()
    for key, g in got.items():
        want = torch.stack([r[key].reshape(()) for r in ref])
        assert torch.allclose(g, want, rtol=RTOL, atol=ATOL[key]), (key, (g - want).abs().max().item(), g[:4], want[:4])
    want_kl = torch.stack([r['kl'].reshape(()) for r in ref]).reshape(mini_epochs, nmb).mean(1)
    got_kl = torch.stack(kls).cpu()
    assert torch.allclose(got_kl, want_kl, rtol=1e-4, atol=ATOL['kl']), (got_kl, want_kl)


def _check_final_params(agent, oracle, steps, lr_max):
    """Parameters after `steps` Adam steps.  Adam's first updates are +-lr * g/|g|-like, so an element
    whose gradient is at rounding-noise level (|g| ~ 1e-9: dead units, bound-loss-only paths) can move
    by lr in the opposite direction when its fp32 gradient differs in the last bits between the GPU's
    and the CPU's summation order.  Hence: (i) the bulk agrees to rtol 1e-4 / atol 2e-6, at most 0.2 % of
    a tensor's elements may deviate, and (ii) no element deviates by more than the total step budget."""
    final, want = agent.model.state_dict(), oracle.model.full_state_dict()
    for name, v in want.items():
        got = final[name].cpu().to(v.dtype)
        if not v.is_floating_point():
            assert torch.equal(got, v), name
            continue
        bad = ~torch.isclose(got, v, rtol=1e-4, atol=2e-6)
        assert bad.float().mean().item() <= 2e-3, (name, bad.float().mean().item())
        assert (got - v).abs().max().item() <= 2.1 * steps * lr_max, (name, (got - v).abs().max().item())


def _oracle_for(params, cap, N, obs_dim, act_dim):
    cpu_params = copy.deepcopy(params)
    cpu_params['config']['device'] = 'cpu'
    oracle = OracleAgent(cpu_params, SyntheticTensorEnv(N, obs_dim, act_dim, device='cpu', seed=1))
    oracle.model.load_full_state_dict(cap['state'])
    return oracle


@pytest.mark.parametrize('graphs', [True, False])
def test_full_size_minibatches_on_the_benchmarked_path_match_oracle(graphs):
    """BASELINE config #3 network and minibatch: 32,768 x 108 -> [400,200,100] -> (1 | 21), two
    minibatches x two mini-epochs.  graphs=True runs the update as the replayed mini-epoch HIP
    graph bench.py times (forced from the first epoch, so the fresh optimiser state matches a
    fresh oracle); graphs=False the eager launches."""
    from rl_games_amd import configs
    from rl_games_amd.agent import A2CAgent
    N, H = 2048, 32
    params = configs.humanoid_65536(num_actors=N, minibatch_size=32768, hip_graphs=graphs)
    params['config']['mini_epochs'] = 2
    torch.manual_seed(5)
    agent = A2CAgent('headline', copy.deepcopy(params))
    assert agent._engine is not None and agent._engine.chain is not None
    agent.init_tensors()
    agent.obs = agent.env_reset()
    caps = _capture_rollout(agent)
    if graphs:
        agent._eager_epochs = 1           # capture + replay already in the first epoch
    agent.update_epoch()
    res = agent.train_epoch()
    if graphs:
        assert agent._graph_epoch is not None and not agent._graph_failed
    assert agent._engine.last_dw_path == 'mfma' and agent._engine.last_dw_library_jobs == 0
    oracle = _oracle_for(params, caps[0], N, 108, 21)
    ref = oracle.update(caps[0]['batch'])
    # dataset preparation at this batch (65,536 rows): normalised values / returns / advantages
    vd = agent.dataset.values_dict
    for key in ('old_values', 'returns', 'advantages'):
        assert torch.allclose(vd[key].cpu().reshape(-1), oracle.dataset[key].reshape(-1), rtol=RTOL, atol=2e-6), key
    _check_epoch(res, ref, nmb=2, mini_epochs=2)
    # device-side adaptive learning rate == the oracle's python-float schedule, bit for bit
    assert agent.optimizer.last_and_next_lr()[1] == oracle.lr
    _check_final_params(agent, oracle, steps=4, lr_max=max(oracle.lr, 3e-4))


@pytest.mark.parametrize('graphs', [True, False])
def test_config2_ant_epoch_matches_oracle(graphs):
    """BASELINE config #2 end to end: 4,096 envs x horizon 16, obs 60, act 8, MLP [256,128,64],
    minibatch 32,768 x 4 mini-epochs (8 optimiser steps) - rollout on the device, update vs the oracle."""
    from rl_games_amd import configs
    from rl_games_amd.agent import A2CAgent
    N, H = 4096, 16
    params = configs.ant_4096(hip_graphs=graphs)
    torch.manual_seed(9)
    agent = A2CAgent('ant', copy.deepcopy(params))
    assert agent._engine is not None and agent._engine.chain is not None
    assert (agent.horizon_length, agent.num_actors, agent.minibatch_size, agent.mini_epochs_num) == (16, 4096, 32768, 4)
    agent.init_tensors()
    agent.obs = agent.env_reset()
    caps = _capture_rollout(agent)
    if graphs:
        agent._eager_epochs = 1
    agent.update_epoch()
    res = agent.train_epoch()
    oracle = _oracle_for(params, caps[0], N, 60, 8)
    batch = caps[0]['batch']
    # the rollout itself: the GAE of the recorded buffers reproduces the recorded returns bit for bit
    tb = agent.experience_buffer.tensor_dict
    ref = oracle.update(batch)
    vd = agent.dataset.values_dict
    for key in ('old_values', 'returns', 'advantages'):
        assert torch.allclose(vd[key].cpu().reshape(-1), oracle.dataset[key].reshape(-1), rtol=RTOL, atol=2e-6), key
    assert tb['rewards'].shape[:2] == (H, N)
    _check_epoch(res, ref, nmb=2, mini_epochs=4)
    assert agent.optimizer.last_and_next_lr()[1] == oracle.lr
    _check_final_params(agent, oracle, steps=8, lr_max=max(oracle.lr, 3e-4))


def test_dataset_preparation_and_obs_statistics_at_65536x32():
    """prepare_dataset (value RunningMeanStd update + normalisation, advantage normalisation) on the
    full 2,097,152-row batch and the observation RunningMeanStd update on a 32,768 x 108 minibatch,
    against the oracle's op-for-op restatement (running_mean_std.py:19-114, a2c_common.py:1586-1660)."""
    from rl_games_amd import ops
    from rl_games_amd.gae import gae_returns_advantages
    N, H = 65536, 32
    g = torch.Generator().manual_seed(3)
    r = torch.randn(N, H, generator=g)
    v = torch.randn(N, H, generator=g) * 2 + 0.5
    d = (torch.rand(N, H, generator=g) < 0.05).to(torch.uint8)
    lv = torch.randn(N, generator=g)
    ld = (torch.rand(N, generator=g) < 0.05).to(torch.uint8)
    ret, adv, partials = gae_returns_advantages(r.to(DEV), v.to(DEV), d.to(DEV), lv.to(DEV), ld.to(DEV), 0.99, 0.95)
    # oracle: time-major scan, then flatten env-major
    advs = O.gae_scan(r.t().unsqueeze(-1).contiguous(), v.t().unsqueeze(-1).contiguous(), d.t().float().contiguous(),
                      lv.unsqueeze(-1), ld.float(), 0.99, 0.95)
    ret_ref = O.flatten_env_major(advs + v.t().unsqueeze(-1))
    assert torch.equal(ret.reshape(-1, 1).cpu(), ret_ref)
    values_flat = v.reshape(-1, 1)
    state = O.new_running_stats(1)
    state['running_mean'].fill_(0.3)
    state['running_var'].fill_(1.7)
    state['count'] = torch.tensor(12345, dtype=torch.int64)
    dmean = state['running_mean'].clone().to(DEV)
    dvar = state['running_var'].clone().to(DEV)
    dcount = state['count'].clone().reshape(1).to(DEV)
    stats = ops.prepare_stats_buffer(DEV)
    B = N * H
    flags = ops.PREP_NORM_VALUE | ops.PREP_NORM_ADV
    ops.prepare_finalize(partials, B, flags, (dmean, dvar, dcount), 1e-5, None, stats)
    nv, nr, na = (torch.empty(B, device=DEV) for _ in range(3))
    ops.prepare_apply(v.reshape(-1).to(DEV), ret.reshape(-1), adv.reshape(-1), flags, stats, out=(nv, nr, na))
    # oracle op chain (a2c_common.py:1598-1634): values first, then returns through the UPDATED statistics
    want = O.prepare_dataset(ret_ref, values_flat, {k: t.clone() for k, t in state.items()})
    st = want['value_stats']
    assert dcount.item() == st['count'].item()
    assert torch.allclose(dmean.cpu(), st['running_mean'], rtol=1e-6, atol=1e-9)
    assert torch.allclose(dvar.cpu(), st['running_var'], rtol=2e-6, atol=1e-9)
    assert torch.allclose(nv.cpu(), want['old_values'].reshape(-1), rtol=RTOL, atol=2e-6)
    assert torch.allclose(nr.cpu(), want['returns'].reshape(-1), rtol=RTOL, atol=2e-6)
    assert torch.allclose(na.cpu(), want['advantages'].reshape(-1), rtol=RTOL, atol=2e-6)
    # observation statistics at the minibatch shape of the benchmark
    x = (torch.randn(32768, 108, generator=g) * 3 + 1)
    ost = O.new_running_stats(108)
    y_ref, ost2 = O.running_stats_forward(ost, x, training=True)
    om, ov, oc = ost['running_mean'].clone().to(DEV), ost['running_var'].clone().to(DEV), ost['count'].clone().reshape(1).to(DEV)
    part, nb = ops.column_moments(x.to(DEV))
    ops.rms_update(part, nb, 108, 32768, 0, om, ov, oc)
    assert torch.allclose(om.cpu(), ost2['running_mean'], rtol=1e-6, atol=1e-8)
    assert torch.allclose(ov.cpu(), ost2['running_var'], rtol=2e-6, atol=1e-9)
    y = ops.rms_apply(x.to(DEV), om, ov, 1e-5, 0)
    assert torch.allclose(y.cpu(), y_ref, rtol=RTOL, atol=2e-6)


def test_reference_rollout_through_the_fused_policy_head(golden):
    """The REAL reference's recorded rollout (tests/golden/epoch.pt: observations, sampled actions,
    mus, sigmas, neglogpacs, values of ModelA2CContinuousLogStd in eval mode, models.py:346-364) fed
    through the fused chain forward + rollout_policy_head kernel: same mus / values, and - with the
    reference's own noise (a - mu)/sigma - the same actions and neglogpacs."""
    from rl_games_amd import ops
    cap = golden('epoch.pt')['default']
    agent = _make_agent(cap)
    agent.model.load_state_dict(cap['init_state'])            # the state the reference played with
    agent.set_eval()
    N, H = cap['env']['num_envs'], cap['params']['config']['horizon_length']
    b = cap['batch']
    A = b['actions'].shape[1]
    obs = b['obses'].reshape(N, H, -1)
    eng = agent._engine
    assert eng is not None and eng.chain is not None
    vm = agent.model.value_mean_std
    buf = agent.experience_buffer
    for t in range(H):
        o = obs[:, t].contiguous().to(DEV)
        heads = eng.forward_obs(o, agent._obs_rms(), agent._obs_eps(), keep=False)
        mu_ref = b['mus'].reshape(N, H, A)[:, t]
        sg_ref = b['sigmas'].reshape(N, H, A)[:, t]
        act_ref = b['actions'].reshape(N, H, A)[:, t]
        noise = ((act_ref.double() - mu_ref.double()) / sg_ref.double()).float().to(DEV)
        actions, values = torch.empty(N, A, device=DEV), torch.empty(N, device=DEV)
        ops.rollout_policy_head(heads, agent.model.a2c_network.sigma.data, noise, (vm.running_mean, vm.running_var),
                                vm.epsilon, actions, values, buf.storage, H, t)
    tb = buf.tensor_dict
    flat = lambda x: x.transpose(0, 1).reshape((N * H,) + tuple(x.shape[2:]))
    assert torch.allclose(flat(tb['mus']).cpu(), b['mus'], rtol=1e-5, atol=1e-6)
    assert torch.allclose(flat(tb['sigmas']).cpu(), b['sigmas'], rtol=1e-6, atol=0)
    assert torch.allclose(flat(tb['actions']).cpu(), b['actions'], rtol=1e-5, atol=2e-6)
    assert torch.allclose(flat(tb['neglogpacs']).cpu().reshape(-1), b['neglogpacs'].reshape(-1), rtol=1e-5, atol=1e-5)
    assert torch.allclose(flat(tb['values']).cpu().reshape(-1, 1), b['values'].reshape(-1, 1), rtol=1e-5, atol=2e-6)
