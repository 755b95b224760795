"""CPU, build container only: the oracle and the host-side mirrors checked DIRECTLY against the
real reference imported from /root/reference (skipped where it is absent, e.g. the GPU box -
there the committed golden fixtures carry the same information).

Pins the rows SURVEY 8c lists as 'parity unpinned' (clipped surrogate, value loss, bound loss,
policy_kl, advantage normalisation) to outputs of the reference functions themselves on seeded
inputs, plus RunningMeanStd / GeneralizedMovingStats / apply_masks / schedulers / PPODataset /
ExperienceBuffer semantics."""
import copy
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import ref_import  # noqa: E402

try:
    ref_import.enable()
    HAVE_REF = ref_import.source() == 'checkout'      # (the staged archive alone - a GPU box - is not this suite's subject)
except ref_import.ReferenceUnavailable:
    HAVE_REF = False

pytestmark = pytest.mark.skipif(not HAVE_REF, reason='/root/reference not present')

from oracle import ppo_oracle as O  # noqa: E402


def gen(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_losses_and_kl_equal_reference_functions(seed):
    from rl_games.common import common_losses
    from rl_games.algos_torch import torch_ext
    g = gen(seed)
    mb, A = 4096, 21
    old_nlp = torch.randn(mb, generator=g)
    nlp = old_nlp + 0.2 * torch.randn(mb, generator=g)
    adv = torch.randn(mb, generator=g)
    v_old = torch.randn(mb, 1, generator=g)
    v = v_old + 0.3 * torch.randn(mb, 1, generator=g)
    R = torch.randn(mb, 1, generator=g)
    mu = 1.2 * torch.randn(mb, A, generator=g)
    assert torch.equal(O.actor_loss(old_nlp, nlp, adv, 0.2), common_losses.actor_loss(old_nlp, nlp, adv, True, 0.2))
    assert torch.equal(O.actor_loss(old_nlp, nlp, adv, 0.2, smooth=True),
                       common_losses.smoothed_actor_loss(old_nlp, nlp, adv, True, 0.2))
    assert torch.equal(O.actor_loss(old_nlp, nlp, adv, 0.2, ppo=False), common_losses.actor_loss(old_nlp, nlp, adv, False, 0.2))
    for clip in (True, False):
        assert torch.equal(O.critic_loss(v_old, v, 0.2, R, clip),
                           common_losses.default_critic_loss(v_old, v, 0.2, R, clip))
    s0 = torch.exp(0.1 * torch.randn(mb, A, generator=g))
    s1 = torch.exp(0.1 * torch.randn(mb, A, generator=g))
    mu1 = mu + 0.1 * torch.randn(mb, A, generator=g)
    assert torch.equal(O.policy_kl(mu, s0, mu1, s1), torch_ext.policy_kl(mu, s0, mu1, s1, True))
    mask = (torch.rand(mb, generator=g) < 0.7).float()
    ref = torch_ext.policy_kl(mu, s0, mu1, s1, False)
    assert torch.equal(O.policy_kl(mu, s0, mu1, s1, mask), (ref * mask).sum() / mask.sum().clamp(min=1.0))
    # apply_masks
    losses = [adv.unsqueeze(1), v, R]
    for m in (None, mask):
        mine, _ = O.masked_means(losses, m)
        theirs, _ = torch_ext.apply_masks(losses, m)
        assert all(torch.equal(a, b) for a, b in zip(mine, theirs))
    # advantage normalisation (a2c_common.py:1634) and its masked variant
    assert torch.equal(O.normalize_advantages(adv), (adv - adv.mean()) / (adv.std() + 1e-8))
    assert torch.equal(O.normalize_advantages(adv, mask), torch_ext.normalization_with_masks(adv, mask))


def test_bound_losses_equal_reference_methods():
    from rl_games.algos_torch.a2c_continuous import A2CAgent as RefAgent

    class Stub:
        bounds_loss_coef = 1e-4
    mu = 1.5 * torch.randn(1000, 8, generator=gen(3))
    assert torch.equal(O.bound_loss(mu, 'bound'), RefAgent.bound_loss(Stub(), mu))
    assert torch.equal(O.bound_loss(mu, 'regularisation'), RefAgent.reg_loss(Stub(), mu))


def test_neglogp_and_entropy_equal_reference_model():
    from rl_games.algos_torch.models import ModelA2CContinuousLogStd
    g = gen(4)
    x, mu = torch.randn(512, 21, generator=g), torch.randn(512, 21, generator=g)
    logstd = (0.2 * torch.randn(21, generator=g)).expand(512, 21)
    sigma = torch.exp(logstd)
    ref = ModelA2CContinuousLogStd.Network.neglogp(None, x, mu, sigma, logstd)
    assert torch.equal(O.neglogp(x, mu, sigma, logstd), ref)
    assert torch.equal(O.normal_entropy(mu, sigma), torch.distributions.Normal(mu, sigma).entropy().sum(-1))


@pytest.mark.parametrize('shape', [(1,), (12,)])
def test_running_mean_std_equals_reference_module(shape):
    from rl_games.algos_torch.running_mean_std import RunningMeanStd
    g = gen(5)
    ref = RunningMeanStd(shape)
    state = O.new_running_stats(shape[0])
    for it in range(4):
        x = torch.randn(300, shape[0], generator=g) * (it + 1) + it
        mask = (torch.rand(300, generator=g) < 0.8).float().unsqueeze(1) if (it == 2 and shape[0] == 1) else None
        ref.train()
        y_ref = ref(x, mask=mask)
        y, state = O.running_stats_forward(state, x, True, mask=mask)
        assert torch.equal(y, y_ref)
        assert torch.equal(state['running_mean'], ref.running_mean)
        assert torch.equal(state['running_var'], ref.running_var)
        assert state['count'].item() == ref.count.item()
    ref.eval()
    x = torch.randn(50, shape[0], generator=g) * 9
    assert torch.equal(O.running_stats_forward(state, x, False)[0], ref(x))
    assert torch.equal(O.running_stats_forward(state, x, False, denorm=True)[0], ref(x, denorm=True))


def test_moving_stats_equal_reference_module():
    from rl_games.algos_torch.moving_mean_std import GeneralizedMovingStats
    g = gen(6)
    ref = GeneralizedMovingStats((1,), decay=0.5)
    ref.train()
    state = O.new_moving_stats(1)
    for it in range(3):
        x = torch.randn(2000, generator=g) * (it + 1)
        mask = (torch.rand(2000, generator=g) < 0.6).float() if it == 1 else None
        y_ref = ref(x, mask=mask)
        y, state = O.moving_stats_forward(state, x, True, 0.5, mask=mask)
        assert torch.equal(y, y_ref)
        for k in ('mean', 'sqrs', 'step'):
            assert torch.equal(state[k], getattr(ref, k))


def test_schedulers_equal_reference():
    from rl_games.common import schedulers as ref
    from rl_games_amd import lr_control as mine
    a, b = ref.AdaptiveScheduler(0.008), mine.AdaptiveScheduler(0.008)
    lr = 3e-4
    for kl in (0.02, 0.001, 0.008, 0.5, 1e-5, 0.016, 0.004):
        assert a.update(lr, 0.0, 0, 0, kl) == b.update(lr, 0.0, 0, 0, kl)
        assert b.update(lr, 0.0, 0, 0, kl)[0] == O.adaptive_lr(lr, kl)
        lr = a.update(lr, 0.0, 0, 0, kl)[0]
    la = ref.LinearScheduler(3e-4, max_steps=100, apply_to_entropy=True, start_entropy_coef=0.01)
    lb = mine.LinearScheduler(3e-4, max_steps=100, apply_to_entropy=True, start_entropy_coef=0.01)
    for ep in (0, 1, 50, 99, 100, 150):
        assert la.update(0, 0.01, ep, 0, 0) == lb.update(0, 0.01, ep, 0, 0)
        assert lb.update(0, 0.01, ep, 0, 0)[0] == O.linear_lr(3e-4, ep, 100)


def test_ppo_dataset_equals_reference():
    from rl_games.common.datasets import PPODataset as Ref
    from rl_games_amd.minibatch import PPODataset as Mine
    B, mb = 64, 16
    vals = {'obs': torch.arange(B * 3).reshape(B, 3).float(), 'advantages': torch.arange(B).float(),
            'mu': torch.zeros(B, 2), 'sigma': torch.zeros(B, 2), 'rnn_states': None, 'rnn_masks': None}
    r, m = Ref(B, mb, False, False, 'cpu', 4), Mine(B, mb, False, False, 'cpu', 4)
    r.update_values_dict({k: (v.clone() if v is not None else None) for k, v in vals.items()})
    m.update_values_dict({k: (v.clone() if v is not None else None) for k, v in vals.items()})
    assert len(r) == len(m)
    for i in range(len(r)):
        a, b = r[i], m[i]
        assert a.keys() == b.keys()
        for k in a:
            assert torch.equal(a[k], b[k])
        r.update_mu_sigma(torch.full((mb, 2), float(i)), torch.full((mb, 2), 2.0 * i))
        m.update_mu_sigma(torch.full((mb, 2), float(i)), torch.full((mb, 2), 2.0 * i))
    assert torch.equal(r.values_dict['mu'], m.values_dict['mu'])
    # rnn slicing
    rs = [torch.arange(2 * 16 * 5).reshape(2, 16, 5).float()]
    r2, m2 = Ref(B, mb, False, True, 'cpu', 4), Mine(B, mb, False, True, 'cpu', 4)
    for d in (r2, m2):
        d.update_values_dict({'obs': vals['obs'].clone(), 'rnn_states': rs})
    for i in range(len(r2)):
        assert torch.equal(r2[i]['obs'], m2[i]['obs'])
        assert torch.equal(r2[i]['rnn_states'][0], m2[i]['rnn_states'][0])
    with pytest.raises(ValueError):
        Mine(10, 3, False, False, 'cpu', 1)


def test_experience_buffer_equals_reference_layout_and_flatten():
    from rl_games.common.experience import ExperienceBuffer as Ref
    from rl_games.common.a2c_common import swap_and_flatten01 as ref_flatten
    from rl_games_amd.rollout_buffer import ExperienceBuffer as Mine
    from rl_games_amd.agent import swap_and_flatten01
    from rl_games_amd.spaces import Box
    env_info = {'observation_space': Box(-np.inf, np.inf, (7,), np.float32),
                'action_space': Box(-1, 1, (3,), np.float32), 'agents': 1, 'value_size': 1}
    algo = {'num_actors': 5, 'horizon_length': 4, 'has_central_value': False, 'use_action_masks': False}
    r, m = Ref(env_info, algo, 'cpu'), Mine(env_info, algo, 'cpu')
    assert r.tensor_dict.keys() == m.tensor_dict.keys()
    g = gen(8)
    for k in r.tensor_dict:
        assert r.tensor_dict[k].shape == m.tensor_dict[k].shape and r.tensor_dict[k].dtype == m.tensor_dict[k].dtype
    for n in range(4):
        for k, t in r.tensor_dict.items():
            val = (torch.rand(t.shape[1:], generator=g) * 5).to(t.dtype)
            r.update_data(k, n, val)
            m.update_data(k, n, val)
    names = ['actions', 'neglogpacs', 'values', 'mus', 'sigmas', 'obses', 'states', 'dones']
    a, b = r.get_transformed_list(ref_flatten, names), m.get_transformed_list(swap_and_flatten01, names)
    assert a.keys() == b.keys()
    for k in a:
        assert torch.equal(a[k], b[k]), k
        assert b[k].data_ptr() == m.storage[k].data_ptr()       # zero-copy in the env-major layout


def test_reference_restores_amd_checkpoint(tmp_path):
    """Interop, reverse direction: tests/golden/amd_checkpoint.pth was written by
    rl_games_amd.A2CAgent.save() on an MI355X (tests/golden/make_amd_checkpoint.py).  The REAL
    reference agent restores it (`A2CBase.restore`, a2c_common.py:927-930) - weights, normaliser
    statistics, Adam state, counters - and trains on."""
    import copy
    from rl_games.torch_runner import Runner
    from rl_games_amd.synthetic_env import SyntheticTensorEnv
    src = os.path.join(HERE, 'golden', 'amd_checkpoint.pth')
    ck = torch.load(src, map_location='cpu', weights_only=False)
    meta = ck.pop('_meta')
    path = str(tmp_path / 'amd_ckpt.pth')
    torch.save(ck, path)
    params = copy.deepcopy(meta['params'])
    params['config'].update(device='cpu', train_dir=str(tmp_path / 'runs'))
    env = SyntheticTensorEnv(meta['env']['num_envs'], meta['env']['obs_dim'], meta['env']['act_dim'],
                             device='cpu', seed=meta['env']['seed'])
    runner = Runner()
    runner.load({'params': copy.deepcopy(params)})
    runner.params['config']['vec_env'] = env
    runner.params['config']['env_info'] = env.get_env_info()
    agent = runner.algo_factory.create(runner.algo_name, base_name='interop', params=runner.params)
    agent.restore(path)
    sd = agent.model.state_dict()
    assert list(sd.keys()) == list(ck['model'].keys())
    for k, v in ck['model'].items():
        assert sd[k].dtype == v.dtype and torch.equal(sd[k], v), k
    assert agent.epoch_num == ck['epoch'] == 2 and agent.frame == ck['frame']
    assert agent.last_mean_rewards == ck['last_mean_rewards'] == -3.5
    # (the reference restores the optimiser's lr but keeps `last_lr` at the config value)
    assert agent.optimizer.param_groups[0]['lr'] == ck['optimizer']['param_groups'][0]['lr']
    ref_opt = agent.optimizer.state_dict()
    for i, st in ck['optimizer']['state'].items():
        assert torch.equal(ref_opt['state'][i]['exp_avg'], st['exp_avg']), i
        assert float(ref_opt['state'][i]['step']) == float(st['step'])
    agent.init_tensors()
    agent.obs = agent.env_reset()
    agent.epoch_num += 1
    res = agent.train_epoch()
    assert all(torch.isfinite(x).all() for x in res[4] + res[5])


def test_schedulers_equal_reference_trajectories():
    """lr_control (pure-function form) vs rl_games/common/schedulers.py: identical python-float
    trajectories for the KL-band rule and the linear ramp (incl. the entropy ramp)."""
    import random
    from rl_games.common import schedulers as R
    from rl_games_amd import lr_control as M
    rnd = random.Random(0)
    a, b = R.AdaptiveScheduler(0.008, 1e-6, 1e-2, 1.5), M.AdaptiveScheduler(0.008, 1e-6, 1e-2, 1.5)
    lr1 = lr2 = 3e-4
    for _ in range(5000):
        kl = rnd.choice([0.001, 0.003, 0.004, 0.005, 0.01, 0.016, 0.02, 0.05, rnd.random() * 0.03])
        lr1, e1 = a.update(lr1, 0.01, 0, 0, kl)
        lr2, e2 = b.update(lr2, 0.01, 0, 0, kl)
        assert lr1 == lr2 and e1 == e2
    assert b.device_rule() == dict(kl_threshold=0.008, min_lr=1e-6, max_lr=1e-2, lr_multiplier=1.5)
    for use_epochs in (True, False):
        l1 = R.LinearScheduler(3e-4, 1e-6, 1000, use_epochs, True, start_entropy_coef=0.01)
        l2 = M.LinearScheduler(3e-4, 1e-6, 1000, use_epochs, True, start_entropy_coef=0.01)
        for e in range(0, 1300, 7):
            assert l1.update(0, 0.01, e, e * 3, 0) == l2.update(0, 0.01, e, e * 3, 0)
    assert M.IdentityScheduler().update(1e-3, 0.5, 3, 4, 0.1) == R.IdentityScheduler().update(1e-3, 0.5, 3, 4, 0.1)


@pytest.mark.parametrize('heads,with_masks', [([5], False), ([5], True), ([3, 4, 2], False), ([3, 4, 2], True)])
def test_discrete_models_equal_reference_models(heads, with_masks):
    """policy.DiscreteA2CModel vs the reference's ModelA2C / ModelA2CMultiDiscrete (models.py:66-206)
    built by the reference's own ModelBuilder from the same params, same weights: training-mode
    outputs (neglogp, entropy, values, normalised logits) are bit-identical, with and without
    CategoricalMasked action masks."""
    import copy
    from rl_games.algos_torch import model_builder
    from rl_games_amd import configs
    from rl_games_amd.policy import PolicyBuilder
    multi = len(heads) > 1
    params = configs.cartpole_discrete()
    if multi:
        params['network']['space'] = {'multi_discrete': None}
        params['model']['name'] = 'multi_discrete_a2c'
    build_cfg = {'actions_num': heads if multi else heads[0], 'input_shape': (6,), 'num_seqs': 4, 'value_size': 1,
                 'normalize_value': False, 'normalize_input': False}
    ref_net = model_builder.ModelBuilder().load(copy.deepcopy(params))
    ref = ref_net.build(copy.deepcopy(build_cfg))
    mine = PolicyBuilder(copy.deepcopy(params)).build(copy.deepcopy(build_cfg))
    assert list(mine.state_dict().keys()) == list(ref.state_dict().keys())
    mine.load_state_dict(ref.state_dict())
    g = gen(sum(heads))
    B, n = 64, sum(heads)
    obs = torch.randn(B, 6, generator=g)
    masks = None
    if with_masks:
        masks = torch.rand(B, n, generator=g) > 0.4
        o = 0
        for s in heads:
            masks[torch.arange(B), o + torch.randint(0, s, (B,), generator=g)] = True
            o += s
    acts = []
    o = 0
    for s in heads:
        w = torch.ones(B, s) if masks is None else masks[:, o:o + s].float()
        acts.append(torch.multinomial(w, 1, generator=g))
        o += s
    prev = torch.cat(acts, 1) if multi else acts[0].squeeze(1)
    a = ref({'is_train': True, 'obs': obs.clone(), 'prev_actions': prev, 'action_masks': masks})
    b = mine({'is_train': True, 'obs': obs.clone(), 'prev_actions': prev, 'action_masks': masks})
    assert torch.equal(a['prev_neglogp'], b['prev_neglogp'])
    assert torch.equal(a['entropy'], b['entropy'])
    assert torch.equal(a['values'], b['values'])
    la = a['logits'] if multi else [a['logits']]
    lb = b['logits'] if multi else [b['logits']]
    assert all(torch.equal(x, y) for x, y in zip(la, lb))
    # and the oracle's categorical loss consumes the same quantities
    lg, vals = mine.forward_heads({'obs': obs})
    batch = {'actions': prev, 'old_logp_actions': a['prev_neglogp'].detach() + 0.1, 'advantages': torch.randn(B, generator=g),
             'old_values': torch.randn(B, 1, generator=g), 'returns': torch.randn(B, 1, generator=g)}
    out = O.categorical_loss_and_grads(lg, vals, batch, dict(e_clip=0.2, clip_value=True, critic_coef=1.0, entropy_coef=0.01),
                                       None, heads, masks)
    assert torch.equal(out['neglogp'], a['prev_neglogp'].detach())
    assert torch.allclose(out['entropy'], a['entropy'].mean().detach(), rtol=1e-6)


@pytest.mark.parametrize('kind', ['mlp', 'lstm', 'central_value'])
def test_continuous_and_recurrent_and_central_models_equal_reference(kind):
    """policy.ContinuousA2CLogStdModel (MLP and LSTM with done resets) and CentralValueModel vs the
    reference's ModelBuilder-built networks from the same params and weights (training-mode forward;
    normalisers off - they are HIP kernels on our side and are tested on the GPU)."""
    import copy
    from rl_games.algos_torch import model_builder
    from rl_games_amd import configs
    from rl_games_amd.policy import PolicyBuilder
    g = gen(11)
    if kind == 'central_value':
        params = {'model': {'name': 'central_value'},
                  'network': {'name': 'actor_critic', 'central_value': True,
                              'mlp': {'units': [24, 16], 'activation': 'elu', 'initializer': {'name': 'default'}}}}
        cfg = {'actions_num': 3, 'input_shape': (9,), 'num_seqs': 4, 'value_size': 1, 'normalize_value': False,
               'normalize_input': False, 'num_agents': 1}
    else:
        params = configs.pendulum_lstm_4096() if kind == 'lstm' else configs.tiny()
        params = {k: params[k] for k in ('model', 'network')}
        if kind == 'lstm':
            params['network']['rnn'] = {'name': 'lstm', 'units': 16, 'layers': 1}
        cfg = {'actions_num': 3, 'input_shape': (9,), 'num_seqs': 8, 'value_size': 1, 'normalize_value': False,
               'normalize_input': False}
    ref = model_builder.ModelBuilder().load(copy.deepcopy(params)).build(copy.deepcopy(cfg))
    mine = PolicyBuilder(copy.deepcopy(params)).build(copy.deepcopy(cfg))
    assert sorted(mine.state_dict().keys()) == sorted(ref.state_dict().keys())
    mine.load_state_dict(ref.state_dict())
    T, S = 4, 8
    B = T * S
    obs = torch.randn(B, 9, generator=g)
    if kind == 'central_value':
        a = ref({'is_train': True, 'obs': obs.clone()})
        b = mine({'is_train': True, 'obs': obs.clone()})
        assert torch.equal(a['values'], b['values'])
        return
    batch = {'is_train': True, 'obs': obs, 'prev_actions': torch.randn(B, 3, generator=g)}
    if kind == 'lstm':
        batch['rnn_states'] = [0.3 * torch.randn(1, S, 16, generator=g) for _ in range(2)]
        batch['seq_length'] = T
        batch['dones'] = (torch.rand(B, generator=g) < 0.3).to(torch.uint8)
    a = ref({k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()})
    b = mine({k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()})
    tol = dict(rtol=1e-5, atol=1e-6) if kind == 'lstm' else dict(rtol=0, atol=0)   # per-step vs segment LSTM calls
    for k in ('prev_neglogp', 'values', 'entropy', 'mus', 'sigmas'):
        assert torch.allclose(a[k], b[k], **tol), k


def test_reference_runner_plugin_seam_constructs_our_agents():
    """the REAL reference Runner with our agents registered in its
    algo_factory (torch_runner.py:117-120) loads an unmodified params dict and calls our
    constructors with (base_name, params) (:258).  Without a GPU the constructors stop at the
    device check - after parsing the reference-format params - which is what this container can
    verify; the GPU tests cover everything behind it."""
    import copy
    from rl_games.torch_runner import Runner
    from rl_games_amd import configs
    from rl_games_amd.agent import A2CAgent
    from rl_games_amd.discrete_agent import DiscreteA2CAgent
    from rl_games_amd.synthetic_env import SyntheticTensorEnv
    for params, cls, env_kw in ((configs.tiny(device='cpu'), A2CAgent, dict(obs_dim=12, act_dim=3)),
                                (configs.cartpole_discrete(device='cpu'), DiscreteA2CAgent,
                                 dict(obs_dim=4, discrete_actions=2))):
        runner = Runner()
        runner.algo_factory.register_builder('a2c_continuous', lambda **kw: A2CAgent(**kw))
        runner.algo_factory.register_builder('a2c_discrete', lambda **kw: DiscreteA2CAgent(**kw))
        runner.load({'params': copy.deepcopy(params)})
        env = SyntheticTensorEnv(params['config']['num_actors'], device='cpu', **env_kw)
        runner.params['config']['vec_env'] = env
        runner.params['config']['env_info'] = env.get_env_info()
        with pytest.raises(RuntimeError, match='MI355X HIP device only'):
            runner.algo_factory.create(runner.algo_name, base_name='run', params=runner.params)


# ----------------------------------------------------------------------------- value_size > 1 torch forms (round 5)

@pytest.mark.parametrize('V,masked,smooth,bound', [(1, False, False, 'bound'), (2, False, False, 'bound'), (3, True, False, 'regularisation'),
                                                   (2, True, True, 'bound'), (2, False, True, None),
                                                   (1, True, 2, 'bound'), (2, False, 2, None),
                                                   (1, False, False, 'bound-rowsigma'), (2, True, True, 'regularisation-rowsigma')])
def test_torch_fallback_losses_equal_the_reference_agent_functions(V, masked, smooth, bound):
    """rl_games_amd/torch_fallback.py (the value_size > 1 path of the agent) against the reference's OWN calc_losses
    (a2c_continuous.py:97-134, called unbound on a stand-in with exactly the attributes it reads), its model epilogue
    (models.py:329-364) and policy_kl - loss, the four scalars, KL, and the gradients w.r.t. mu / logstd / values."""
    import types
    from rl_games.algos_torch import a2c_continuous, torch_ext
    from rl_games.algos_torch.models import ModelA2CContinuousLogStd
    from rl_games.common import common_losses
    from rl_games_amd import torch_fallback as tf
    g = gen(3 + V)
    mb, A = 512, 5
    # ('-rowsigma': a state-dependent sigma head, fixed_sigma False - log sigma per row [mb, A]; round 6)
    row_sigma = bound is not None and bound.endswith('-rowsigma')
    if row_sigma:
        bound = bound[:-len('-rowsigma')]
    mu0 = 1.3 * torch.randn(mb, A, generator=g)
    logstd0 = 0.2 * torch.randn(*((mb, A) if row_sigma else (A,)), generator=g)
    values0 = torch.randn(mb, V, generator=g)
    actions = mu0 + torch.randn(mb, A, generator=g)
    old_nlp = 0.5 * torch.randn(mb, generator=g) + 6.0
    adv = torch.randn(mb, generator=g)
    old_values = values0 + 0.3 * torch.randn(mb, V, generator=g)
    returns = torch.randn(mb, V, generator=g)
    old_mu = mu0 + 0.1 * torch.randn(mb, A, generator=g)
    old_sigma = torch.exp(0.2 * torch.randn(mb, A, generator=g))
    mask = (torch.rand(mb, generator=g) < 0.7).float() if masked else None
    coef_b = None if bound is None else 1e-3
    kind = {None: 0, 'bound': 1, 'regularisation': 2}[bound]

    def leaves():
        return [t.clone().requires_grad_(True) for t in (mu0, logstd0, values0)]
    # ---- ours
    mu, logstd, values = leaves()
    loss, sc, sigma = tf.ppo_loss(mu, logstd, values, actions, old_nlp, adv, old_values, returns, e_clip=0.2, critic_coef=2.0,
                                  entropy_coef=0.01, bounds_coef=coef_b if coef_b is not None else 0.0, bound_kind=kind,
                                  clip_value=True, smooth=smooth, mask=mask)
    loss.backward()
    ours = (loss.detach(), sc, [t.grad.clone() for t in (mu, logstd, values)])
    kl = tf.policy_kl(mu.detach(), sigma.detach().expand_as(mu), old_mu, old_sigma, mask)
    # ---- the reference: epilogue as ModelA2CContinuousLogStd.Network.forward computes it, then calc_losses
    mu, logstd, values = leaves()
    logstd_full = mu * 0.0 + logstd                                                    # network_builder.py:506-512
    sigma_full = torch.exp(logstd_full)
    distr = torch.distributions.Normal(mu, sigma_full, validate_args=False)
    entropy = distr.entropy().sum(dim=-1)
    nlp = torch.squeeze(ModelA2CContinuousLogStd.Network.neglogp(None, actions, mu, sigma_full, logstd_full))
    # (smooth == 2: `ppo: False`, the plain A2C actor loss - round 6)
    stub = types.SimpleNamespace(ppo=(smooth != 2), has_value_loss=True, model=None, clip_value=True,
                                 bound_loss_type=bound, bounds_loss_coef=coef_b, critic_coef=2.0, entropy_coef=0.01,
                                 ppo_device='cpu')
    stub.bound_loss = types.MethodType(a2c_continuous.A2CAgent.bound_loss, stub)
    stub.reg_loss = types.MethodType(a2c_continuous.A2CAgent.reg_loss, stub)
    stub.bounds_loss_coef = coef_b
    fn = common_losses.smoothed_actor_loss if smooth == 1 else common_losses.actor_loss
    ref_loss, a, c, e, b, _ = a2c_continuous.A2CAgent.calc_losses(stub, fn, old_nlp, nlp, adv, 0.2, old_values, values, returns,
                                                               mu, entropy, mask)
    ref_loss.backward()
    for got, want in ((ours[0], ref_loss.detach()), (ours[1]['a_loss'], a), (ours[1]['c_loss'], c), (ours[1]['entropy'], e),
                      (ours[1]['b_loss'], b.reshape(()) if bound is None else b)):
        assert torch.allclose(got, want.detach(), rtol=1e-6, atol=1e-7), (got, want)
    for got, leaf in zip(ours[2], (mu, logstd, values)):
        assert torch.allclose(got, leaf.grad, rtol=1e-5, atol=1e-8)
    ref_kl = torch_ext.policy_kl(mu.detach(), sigma_full.detach(), old_mu, old_sigma, mask is None)
    if mask is not None:
        ref_kl = (ref_kl * mask).sum() / mask.sum()                                     # a2c_continuous.py:218-221
    assert torch.allclose(kl, ref_kl, rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize('masked', [False, True])
def test_torch_fallback_advantage_normalisation_and_ema_equal_the_reference(masked):
    from rl_games.algos_torch import torch_ext
    from rl_games.algos_torch.moving_mean_std import GeneralizedMovingStats as RefStats
    from rl_games_amd import torch_fallback as tf
    from rl_games_amd.normalizers import GeneralizedMovingStats
    g = gen(11)
    adv = 3 * torch.randn(4096, generator=g) + 0.5
    mask = (torch.rand(4096, generator=g) < 0.6).float() if masked else None
    assert torch.allclose(tf.normalize_advantages(adv, mask), torch_ext.normalization_with_masks(adv, mask), rtol=1e-6, atol=1e-7)
    ours, ref = GeneralizedMovingStats((1,), decay=0.5), RefStats((1,), decay=0.5)
    for k in range(3):
        x = (k + 1) * torch.randn(4096, generator=g) - k
        assert torch.allclose(ours(x, mask=mask), ref(x, mask=mask), rtol=1e-6, atol=1e-7)
    for name in ('mean', 'sqrs', 'step'):
        assert torch.equal(getattr(ours, name), getattr(ref, name))
    # nothing valid: no update
    before = (ours.mean.clone(), ours.step.clone())
    ours(adv, mask=torch.zeros(4096))
    assert torch.equal(ours.mean, before[0]) and torch.equal(ours.step, before[1])


def test_state_dependent_sigma_network_equals_the_reference_builder():
    """fixed_sigma False (network_builder.py:341-344, :508-511, init_state_dependent_sigma_head :14-25): the policy module
    builds the same parameters as the reference's A2CBuilder network - names, shapes, the constant-initialiser rule (bias =
    val, weights zero) - and, on the reference's weights, returns the same (mu, log sigma, value) and the same model outputs
    (neglogp, entropy) bit for bit."""
    from rl_games.algos_torch import network_builder
    from rl_games.algos_torch.models import ModelA2CContinuousLogStd
    from rl_games_amd.policy import ActorCriticNetwork, ContinuousA2CLogStdModel
    net_params = {'name': 'actor_critic', 'separate': False,
                  'space': {'continuous': {'mu_activation': 'None', 'sigma_activation': 'None',
                                           'mu_init': {'name': 'default'}, 'sigma_init': {'name': 'const_initializer', 'val': -0.7},
                                           'fixed_sigma': False}},
                  'mlp': {'units': [32, 16], 'activation': 'elu', 'initializer': {'name': 'default'}}}
    builder = network_builder.A2CBuilder()
    builder.load(copy.deepcopy(net_params))
    torch.manual_seed(4)
    ref_net = builder.build('a2c', actions_num=3, input_shape=(7,), value_size=1, num_seqs=8)
    ours = ActorCriticNetwork(copy.deepcopy(net_params), actions_num=3, input_shape=(7,), value_size=1, num_seqs=8)
    ref_sd, our_sd = ref_net.state_dict(), ours.state_dict()
    assert {k: tuple(v.shape) for k, v in ref_sd.items()} == {k: tuple(v.shape) for k, v in our_sd.items()}
    assert torch.equal(our_sd['sigma.bias'], torch.full((3,), -0.7)) and not our_sd['sigma.weight'].any()
    assert torch.equal(ref_sd['sigma.bias'], our_sd['sigma.bias']) and torch.equal(ref_sd['sigma.weight'], our_sd['sigma.weight'])
    with torch.no_grad():
        for k, v in ref_sd.items():
            if k.startswith('sigma.'):
                ref_sd[k] = v + 0.1 * torch.randn(v.shape, generator=gen(9))      # a sigma head that does depend on the state
    ref_net.load_state_dict(ref_sd)
    ours.load_state_dict(ref_sd)
    obs = torch.randn(8, 7, generator=gen(10))
    want = ref_net({'obs': obs, 'rnn_states': None})
    got = ours({'obs': obs, 'rnn_states': None})
    for w, g_ in zip(want[:3], got[:3]):
        assert torch.equal(w, g_)
    assert got[1].shape == (8, 3) and (got[1][0] != got[1][1]).any()
    # the model epilogue around it (models.py:329-364)
    ref_model = ModelA2CContinuousLogStd.Network(ref_net, obs_shape=(7,), normalize_value=False, normalize_input=False, value_size=1)
    our_model = ContinuousA2CLogStdModel(ours, (7,), False, False, 1)
    acts = torch.randn(8, 3, generator=gen(11))
    a = ref_model({'is_train': True, 'prev_actions': acts, 'obs': obs, 'rnn_states': None})
    b = our_model({'is_train': True, 'prev_actions': acts, 'obs': obs, 'rnn_states': None})
    for key in ('prev_neglogp', 'values', 'entropy', 'mus', 'sigmas'):
        assert torch.equal(a[key], b[key]), key
    mu, logstd, value, _ = our_model.forward_heads({'obs': obs})
    assert torch.equal(logstd, got[1]) and torch.equal(mu, got[0])


@pytest.mark.parametrize('space', [{'sigma_parametrization': 'softplus', 'min_sigma': 0.05},
                                   {'sigma_parametrization': 'softplus'},
                                   {'sigma_parametrization': 'scalar'}, {'sigma_parametrization': 'linear', 'min_sigma': 0.2},
                                   {'logstd_bounds': [-1.0, 0.5]}, {'min_sigma': 0.1}, {'logstd_bounds': [-2.0, 0.0], 'min_sigma': 0.3},
                                   {}])
@pytest.mark.parametrize('fixed', [True, False])
def test_sigma_parametrisations_equal_the_reference_model(space, fixed):
    """network_builder.py:311-322 + models.py:272-301 (`apply_sigma_parametrization`): the reference's model and this
    one on the same weights - sigmas, neglogp and entropy of a training call bit for bit; only the unbounded exp form is
    `plain_sigma` (what the fused kernels compute)."""
    from rl_games.algos_torch import network_builder, models
    from rl_games_amd.policy import ActorCriticNetwork, ContinuousA2CLogStdModel
    sp = copy.deepcopy(_SPACE)
    sp['continuous'].update(space, fixed_sigma=fixed, sigma_init={'name': 'const_initializer', 'val': 0.4})
    net_params = {'name': 'actor_critic', 'separate': False, 'space': sp,
                  'mlp': {'units': [32, 16], 'activation': 'elu', 'initializer': {'name': 'default'}}}
    builder = network_builder.A2CBuilder()
    builder.load(copy.deepcopy(net_params))
    torch.manual_seed(5)
    kw = dict(actions_num=3, input_shape=(7,), value_size=1, num_seqs=4)
    ref = models.ModelA2CContinuousLogStd(builder).build(dict(kw, normalize_value=False, normalize_input=False))
    ours_net = ActorCriticNetwork(copy.deepcopy(net_params), **kw)
    assert ours_net.plain_sigma == (space == {})
    ours = ContinuousA2CLogStdModel(ours_net, (7,), False, False, 1)
    sd = ref.state_dict()
    with torch.no_grad():                                       # raw values on both sides of every bound / floor
        for k, v in sd.items():
            if 'sigma' in k:
                sd[k] = v + 1.5 * torch.randn(v.shape, generator=gen(3))
    ref.load_state_dict(sd)
    ours.load_state_dict(sd)
    obs = torch.randn(16, 7, generator=gen(4))
    acts = torch.randn(16, 3, generator=gen(5))
    want = ref({'is_train': True, 'obs': obs.clone(), 'prev_actions': acts})
    got = ours({'is_train': True, 'obs': obs.clone(), 'prev_actions': acts})
    for k in ('sigmas', 'mus', 'prev_neglogp', 'entropy', 'values'):
        assert torch.equal(want[k].expand_as(got[k]), got[k]), k


_SPACE = {'continuous': {'mu_activation': 'None', 'sigma_activation': 'None', 'mu_init': {'name': 'default'},
                         'sigma_init': {'name': 'const_initializer', 'val': 0}, 'fixed_sigma': True}}


@pytest.mark.parametrize('name,over', [
    ('layer_norm', {'normalization': 'layer_norm'}),
    ('layer_norm_first_only', {'normalization': 'layer_norm', 'mlp': {'units': [24, 24], 'norm_only_first_layer': True}}),
    ('batch_norm', {'normalization': 'batch_norm'}),
    ('d2rl_batch_norm_first_only', {'mlp': {'d2rl': True, 'norm_only_first_layer': True}, 'normalization': 'batch_norm'}),
    ('d2rl', {'mlp': {'d2rl': True}}),
    ('d2rl_layer_norm', {'mlp': {'d2rl': True}, 'normalization': 'layer_norm'}),
    ('gru_two_layers', {'rnn': {'name': 'gru', 'units': 12, 'layers': 2}}),
    ('lstm_layer_norm_concat', {'rnn': {'name': 'lstm', 'units': 12, 'layers': 1, 'layer_norm': True, 'concat_input': True,
                                        'concat_output': True}}),
    ('lstm_before_mlp', {'rnn': {'name': 'lstm', 'units': 12, 'layers': 1, 'before_mlp': True, 'concat_output': True}}),
    ('discrete_d2rl_layer_norm', {'mlp': {'d2rl': True}, 'normalization': 'layer_norm', 'space': {'discrete': {}}}),
    ('separate_lstm_layer_norm', {'separate': True, 'rnn': {'name': 'lstm', 'units': 12, 'layers': 1, 'layer_norm': True}}),
    ('separate_gru_before_mlp', {'separate': True, 'rnn': {'name': 'gru', 'units': 12, 'layers': 2, 'before_mlp': True}}),
    ('discrete_separate_lstm_concat', {'separate': True, 'space': {'discrete': {}},
                                       'rnn': {'name': 'lstm', 'units': 12, 'layers': 1, 'concat_input': True}}),
    ('discrete_lstm', {'rnn': {'name': 'lstm', 'units': 12, 'layers': 1}, 'space': {'discrete': {}}}),
    ('discrete_gru_before_mlp_concat', {'rnn': {'name': 'gru', 'units': 12, 'layers': 1, 'before_mlp': True,
                                                'concat_output': True}, 'space': {'discrete': {}}}),
])
def test_network_zoo_layouts_equal_the_reference_builder(name, over):
    """The network layouts that left the NotImplementedError list in round 6 - layer normalisation (network_builder.py:105-132),
    D2RL trunks (d2rl.py), GRU / multi-layer RNNs, an RNN in front of the MLP, concatenated RNN inputs / outputs, layer norm
    behind the RNN (:250-276, :447-487) - against the reference's own A2CBuilder network: the same parameter names and shapes,
    and on the reference's weights the same outputs (and next RNN states) bit for bit, with done resets inside a sequence."""
    from rl_games.algos_torch import network_builder
    from rl_games_amd.policy import ActorCriticNetwork
    net_params = {'name': 'actor_critic', 'separate': False, 'space': copy.deepcopy(_SPACE),
                  'mlp': {'units': [32, 16], 'activation': 'elu', 'initializer': {'name': 'default'}}}
    for k, v in over.items():
        if k == 'mlp':
            net_params['mlp'].update(v)
        else:
            net_params[k] = v
    discrete = 'discrete' in net_params['space']
    builder = network_builder.A2CBuilder()
    builder.load(copy.deepcopy(net_params))
    torch.manual_seed(5)
    kw = dict(actions_num=3, input_shape=(7,), value_size=1, num_seqs=4)
    ref_net = builder.build('a2c', **kw)
    ours = ActorCriticNetwork(copy.deepcopy(net_params), **kw)
    ref_sd = ref_net.state_dict()
    assert {k: tuple(v.shape) for k, v in ref_sd.items()} == {k: tuple(v.shape) for k, v in ours.state_dict().items()}
    assert not getattr(ours, 'plain_trunk', True) or name in ('gru_two_layers', 'discrete_lstm')
    with torch.no_grad():                                       # LayerNorm weights away from their (1, 0) initialisation
        for k, v in ref_sd.items():
            if 'norm' in k and v.is_floating_point():
                ref_sd[k] = v + 0.2 * torch.randn(v.shape, generator=gen(3))
    ref_net.load_state_dict(ref_sd)
    ours.load_state_dict(ref_sd)
    T, S = 3, 4                                                 # seq_length 3, 4 sequences
    obs = torch.randn(S * T, 7, generator=gen(4))
    d = {'obs': obs, 'rnn_states': None}
    if 'rnn' in net_params:
        layers, units = net_params['rnn']['layers'], net_params['rnn']['units']
        n_states = (2 if net_params['rnn']['name'] == 'lstm' else 1) * (2 if net_params['separate'] else 1)
        assert len(ours.get_default_rnn_state()) == len(ref_net.get_default_rnn_state()) == n_states
        states = tuple(torch.randn(layers, S, units, generator=gen(6 + i)) for i in range(n_states))
        dones = (torch.rand(S * T, generator=gen(8)) < 0.3).float()
        d = {'obs': obs, 'rnn_states': states, 'seq_length': T, 'dones': dones}
    want = ref_net(dict(d))
    got = ours(dict(d))
    n_out = 2 if discrete else 3
    for w, g_ in zip(want[:n_out], got[:n_out]):
        assert torch.allclose(w, g_, rtol=0, atol=0) or torch.allclose(w, g_, rtol=1e-6, atol=1e-7), name
    if 'rnn' in net_params:
        assert len(want[n_out]) == len(got[n_out]) == n_states
        for w, g_ in zip(want[n_out], got[n_out]):
            assert torch.allclose(w, g_, rtol=1e-6, atol=1e-7)
    else:
        for w, g_ in zip(want[:n_out], got[:n_out]):
            assert torch.equal(w, g_), name
    if 'batch_norm' in name:                                    # that call was in training mode: batch statistics, running
        ref_net.eval()                                          # statistics updated; now the rollout's mode
        ours.eval()
        for (k, w), g_ in zip(ref_net.state_dict().items(), ours.state_dict().values()):
            assert torch.equal(w, g_), k
        for w, g_ in zip(ref_net(dict(d))[:n_out], ours(dict(d))[:n_out]):
            assert torch.equal(w, g_), name
