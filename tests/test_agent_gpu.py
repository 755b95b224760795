"""GPU: the A2CAgent (rl_games_amd/agent.py) against golden vectors recorded from the REAL
reference agent's train_epoch, and against the CPU oracle epoch, on identical rollout tensors.

Comparison granularity follows SURVEY 8a' pitfall 12: minibatches are contiguous env blocks and
never shuffled, so per-minibatch scalars are compared in order."""
import copy

import numpy as np
import pytest
import torch

from oracle import ppo_oracle as O
from oracle.ppo_epoch_oracle import OracleAgent
from rl_games_amd.synthetic_env import SyntheticTensorEnv

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


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


def _to_dev(batch):
    return {k: v.to(DEV) for k, v in batch.items()}


_SIGMA_FORMS = ('softplus_state_sigma', 'bounded_exp_floor', 'linear_sigma')
_BATCH_NORM = ('batch_norm', 'd2rl_batch_norm')


@pytest.mark.parametrize('variant', ('default', 'smooth_reg_ema', 'state_sigma', 'ppo_false', 'd2rl_layer_norm') + _SIGMA_FORMS + _BATCH_NORM)
def test_update_matches_reference_epoch(golden, variant):
    """An epoch's update against the recording of the REAL reference agent's train_epoch on the same rollout
    (tests/golden/make_golden.py).  Round 6 (epoch_extra.pt): 'state_sigma' - a state-dependent sigma head
    (fixed_sigma False: the policy runs the reference's operation sequence as torch ops with autograd around this library's
    dataset / optimiser kernels, agent._forward_loss_backward_general), 'ppo_false' - the plain A2C actor loss inside the
    fused loss tile - and 'd2rl_layer_norm' - a D2RL trunk with layer normalisation: torch modules + autograd between this
    library's loss, statistics and optimiser kernels."""
    extra = variant in ('state_sigma', 'ppo_false', 'd2rl_layer_norm')
    # (epoch_sigma_forms.pt, round 6: the sigma parametrisations of models.py:272-301 - softplus + floor on a state-dependent
    #  head, exp with active log-sigma bounds + floor, the linear form - as torch ops between the kernels)
    # (epoch_batch_norm.pt: BatchNorm1d trunks - batch statistics in the update, whose running averages and batch counters
    #  end where the reference's do: `final_state` holds the buffers too)
    fixture = ('epoch_sigma_forms.pt' if variant in _SIGMA_FORMS else 'epoch_batch_norm.pt' if variant in _BATCH_NORM else
               'epoch_extra.pt' if extra else 'epoch.pt')
    cap = golden(fixture)[variant]
    agent = _make_agent(cap)
    assert (agent._engine is None) == (variant in ('state_sigma', 'd2rl_layer_norm') + _SIGMA_FORMS + _BATCH_NORM)
    agent.model.load_state_dict(cap['state_after_rollout'])
    batch = _to_dev(cap['batch'])
    agent.set_train()
    agent.prepare_dataset(batch)
    ds = cap['dataset']
    vd = agent.dataset.values_dict
    assert torch.allclose(vd['old_values'].cpu(), ds['old_values'], rtol=1e-5, atol=1e-6)
    assert torch.allclose(vd['returns'].cpu(), ds['returns'], rtol=1e-5, atol=1e-6)
    assert torch.allclose(vd['advantages'].cpu(), ds['advantages'], rtol=1e-5, atol=1e-6)
    rows = []
    for mini_ep in range(agent.mini_epochs_num):
        for i in range(len(agent.dataset)):
            a, c, e, kl, lr, lr_mul, mu, sigma, b = agent.train_actor_critic(agent.dataset[i])
            rows.append(torch.stack([a, c, e, kl, b]).clone())
    rows = torch.stack(rows).cpu()
    assert torch.allclose(rows[:, 0], cap['a_losses'], rtol=1e-5, atol=2e-6)
    assert torch.allclose(rows[:, 1], cap['c_losses'], rtol=1e-5, atol=2e-6)
    assert torch.allclose(rows[:, 2], cap['entropies'], rtol=1e-5, atol=2e-6)
    if cap['b_losses'] is not None:
        assert torch.allclose(rows[:, 4], cap['b_losses'], rtol=1e-5, atol=1e-7)
    nmb = len(agent.dataset)
    kls = rows[:, 3].reshape(agent.mini_epochs_num, nmb).mean(1)
    assert torch.allclose(kls, cap['mini_epoch_kls'], rtol=1e-4, atol=1e-7)
    # device-side adaptive learning rate == the reference's python-float trajectory, bit for bit
    used, nxt = agent.optimizer.last_and_next_lr()
    assert nxt == cap['lrs'][-1]
    assert used == cap['last_lr']
    final = agent.model.state_dict()
    for k, v in cap['final_state'].items():
        if variant == 'd2rl_batch_norm' and (k.endswith('linears.1.bias') or k.endswith('norm_layers.0.bias')):
            # a constant shift in front of the second layer's BatchNorm (d2rl.py:29-31: linear, norm, activation) - that
            # layer's bias, and the first BatchNorm's bias through the linear map - has the gradient 0: what arrives is
            # rounding noise of either implementation, which Adam turns into lr-sized steps of random sign
            assert v.abs().max() < 1e-3 and final[k].abs().max() < 1e-3, k
            continue
        if variant == 'd2rl_batch_norm' and k.endswith('norm_layers.1.running_mean'):
            # (... and the running mean of that BatchNorm's input follows those two random walks)
            assert torch.allclose(final[k].cpu(), v, rtol=0, atol=3e-4), k
            continue
        tol = dict(rtol=1e-4, atol=2e-6) if v.is_floating_point() else dict(rtol=0, atol=0)
        assert torch.allclose(final[k].cpu().to(v.dtype), v, **tol), k
    assert torch.allclose(vd['mu'].cpu(), ds['mu'], rtol=1e-4, atol=1e-5)
    assert torch.allclose(vd['sigma'].cpu(), ds['sigma'], rtol=1e-5, atol=1e-6)
    # optimiser moments in torch.optim.Adam's state_dict format
    opt = agent.optimizer.state_dict()['state']
    for i, st in cap['opt_state'].items():
        assert torch.allclose(opt[i]['exp_avg'].cpu(), st['exp_avg'], rtol=1e-3, atol=1e-6), i
        assert int(float(opt[i]['step'])) == int(float(st['step']))
    if 'adv_ema' in cap:
        ema = agent.advantage_mean_std
        assert ema.step.item() == cap['adv_ema']['step'].item()
        assert torch.allclose(ema.mean.cpu(), cap['adv_ema']['mean'], rtol=1e-5, atol=1e-7)
        assert torch.allclose(ema.sqrs.cpu(), cap['adv_ema']['sqrs'], rtol=1e-5)


def test_rollout_epilogue_matches_reference_buffers(golden):
    """Reference rollout buffers -> env-major storage -> fused GAE == the reference's returns,
    bit for bit; values view unchanged."""
    from rl_games_amd.gae import gae_returns_advantages
    cap = golden('epoch.pt')['default']
    buf = cap['buffers']
    H, N = buf['dones'].shape
    rp = buf['rewards'][..., 0].t().contiguous().to(DEV)
    vp = buf['values'][..., 0].t().contiguous().to(DEV)
    dp = buf['dones'].t().contiguous().to(DEV)
    cfg = cap['params']['config']
    ret, adv, _ = gae_returns_advantages(rp, vp, dp, cap['last_values'].reshape(-1).to(DEV),
                                         cap['last_dones'].to(DEV), cfg['gamma'], cfg['tau'])
    assert torch.equal(ret.reshape(-1, 1).cpu(), cap['batch']['returns'])
    assert torch.equal(adv.reshape(-1).cpu(), (cap['batch']['returns'] - cap['batch']['values']).reshape(-1))


def test_full_train_epoch_runs_and_matches_oracle_on_same_rollout():
    """End to end on the device: play_steps -> (copy the rollout to the CPU oracle) -> both update;
    per-minibatch losses agree.  Also checks counts, meters and buffer/index layout."""
    from rl_games_amd import configs
    from rl_games_amd.agent import A2CAgent
    params = configs.tiny(num_actors=128, horizon=8, obs_dim=10, act_dim=4)
    agent = A2CAgent('test', copy.deepcopy(params))
    agent.init_tensors()
    agent.obs = agent.env_reset()
    agent.set_eval()
    with torch.no_grad():
        batch = agent.play_steps()
    N, H = 128, 8
    # env-major flat index: row env*H + t of the flat batch == buffer[t, env]
    tb = agent.experience_buffer.tensor_dict
    for key, flat in (('obses', batch['obses']), ('actions', batch['actions']), ('dones', batch['dones'])):
        assert flat.data_ptr() == agent.experience_buffer.storage[key].data_ptr()      # zero copy
        assert torch.equal(flat.reshape((N, H) + flat.shape[1:]).transpose(0, 1), tb[key])
    assert torch.equal(tb['dones'][0], torch.ones(N, dtype=torch.uint8, device=DEV))      # :666
    # oracle on the same rollout
    cpu_params = copy.deepcopy(params)
    env = SyntheticTensorEnv(N, 10, 4, device='cpu', seed=1)
    oracle = OracleAgent(cpu_params, env)
    sd = {k: v.detach().cpu().clone() for k, v in agent.model.state_dict().items()}
    oracle.model.load_full_state_dict(sd)
    cpu_batch = {k: v.detach().cpu().clone() for k, v in batch.items() if isinstance(v, torch.Tensor)}
    # GAE consistency of the device rollout with the oracle scan: bit-exact given the bootstrap
    # values the rollout itself used (engine forward; the torch-module forward differs by an ulp)
    last_values = (agent._fast_values(agent.obs).reshape(-1, 1) if agent._fast_rollout_ok()
                   else agent.get_values(agent.obs))
    assert torch.allclose(last_values, agent.get_values(agent.obs), rtol=1e-5, atol=1e-6)
    advs = O.gae_scan(tb['rewards'].cpu(), tb['values'].cpu(), tb['dones'].cpu().float(),
                      last_values.cpu(), agent.dones.cpu().float(), 0.99, 0.95)
    assert torch.equal(cpu_batch['returns'], O.flatten_env_major(advs + tb['values'].cpu()))
    ref = oracle.update(cpu_batch)
    agent.set_train()
    agent.prepare_dataset(batch)
    k = 0
    for mini_ep in range(agent.mini_epochs_num):
        for i in range(len(agent.dataset)):
            a, c, e, kl, lr, lr_mul, mu, sigma, b = agent.train_actor_critic(agent.dataset[i])
            r = ref[k]
            for got, key in ((a, 'a_loss'), (c, 'c_loss'), (e, 'entropy'), (kl, 'kl'), (b, 'b_loss')):
                assert np.isclose(got.item(), r[key].item(), rtol=1e-5, atol=2e-6), (k, key, got.item(), r[key].item())
            k += 1
    assert agent.optimizer.last_and_next_lr()[1] == oracle.lr
    # a complete train_epoch through the public entry point
    agent.update_epoch()
    out = agent.train_epoch()
    assert len(out[4]) == agent.mini_epochs_num * agent.num_minibatches
    assert agent.model.running_mean_std.count.item() == 1 + 2 * agent.mini_epochs_num * N * H
    assert agent.model.value_mean_std.count.item() == 1 + 2 * 2 * N * H
    assert agent.game_lengths.current_size > 0


def test_plain_a2c_loss_when_ppo_is_false_matches_oracle():
    """`ppo: False` (a2c_common.py:280; the else branch of common_losses.actor_loss :80: a_loss = neglogp * advantage)
    through the fused backward launch's loss tile (surrogate kind 2): every minibatch step of an epoch against the
    oracle - losses, KL, learning rate, and after the epoch the parameters."""
    from rl_games_amd import configs
    from rl_games_amd.agent import A2CAgent
    params = configs.tiny(num_actors=128, horizon=8, obs_dim=10, act_dim=4)
    params['config']['ppo'] = False
    agent = A2CAgent('test', copy.deepcopy(params))
    assert agent.surrogate == 2 and agent._engine is not None
    agent.init_tensors()
    agent.obs = agent.env_reset()
    agent.set_eval()
    with torch.no_grad():
        batch = agent.play_steps()
    oracle = OracleAgent(copy.deepcopy(params), SyntheticTensorEnv(128, 10, 4, device='cpu', seed=1))
    assert oracle.hp['ppo'] is False
    oracle.model.load_full_state_dict({k: v.detach().cpu().clone() for k, v in agent.model.state_dict().items()})
    ref = oracle.update({k: v.detach().cpu().clone() for k, v in batch.items() if isinstance(v, torch.Tensor)})
    agent.set_train()
    agent.prepare_dataset(batch)
    k = 0
    for _ in range(agent.mini_epochs_num):
        for i in range(len(agent.dataset)):
            a, c, e, kl, lr, lr_mul, mu, sigma, b = agent.train_actor_critic(agent.dataset[i])
            r = ref[k]
            for got, key in ((a, 'a_loss'), (c, 'c_loss'), (e, 'entropy'), (kl, 'kl'), (b, 'b_loss')):
                assert np.isclose(got.item(), r[key].item(), rtol=1e-5, atol=2e-6), (k, key, got.item(), r[key].item())
            k += 1
    # (that the oracle's `ppo: False` branch is the reference's plain A2C loss is pinned on the CPU: the 'ppo_false' goldens of
    #  tests/test_oracle_epoch.py and tests/test_vs_reference_cpu.py; a magnitude check here depended on the random weights)
    assert agent.optimizer.last_and_next_lr()[1] == oracle.lr
    want = oracle.model.full_state_dict()
    for name, v in agent.model.state_dict().items():
        if v.is_floating_point() and v.numel() >= 16 and name in want:
            w = want[name]
            assert ((v.cpu().to(w.dtype) - w).abs().mean() / w.abs().mean().clamp_min(1e-12)).item() <= 1e-4, name


@pytest.mark.parametrize('rnn', [{'name': 'gru', 'units': 16, 'layers': 2},
                                 {'name': 'lstm', 'units': 16, 'layers': 1, 'layer_norm': True, 'concat_input': True, 'concat_output': True},
                                 {'name': 'lstm', 'units': 16, 'layers': 1, 'before_mlp': True},
                                 {'name': 'lstm', 'units': 16, 'layers': 1, 'separate': True},
                                 {'name': 'gru', 'units': 16, 'layers': 1, 'separate': True, 'concat_output': True}],
                         ids=['gru_two_layers', 'lstm_layer_norm_concat', 'lstm_before_mlp', 'separate_lstm',
                              'separate_gru_concat'])
def test_recurrent_layouts_outside_the_engine_train_on_the_device(rnn):
    """GRU / multi-layer RNNs and the RNN layout options of network_builder.py:250-276 (construction and outputs pinned to the
    reference builder on the CPU: tests/test_vs_reference_cpu.py::test_network_zoo_layouts_...): two epochs through
    play_steps_rnn and the update on the device - torch modules with autograd around this library's rollout, GAE, dataset,
    loss and optimiser kernels - finite losses, a small KL in an epoch's first minibatch (the policy that played is the one
    evaluated), parameters that move."""
    from rl_games_amd import configs
    from rl_games_amd.agent import A2CAgent
    params = configs.tiny(num_actors=64, horizon=8, obs_dim=10, act_dim=3, seq_length=4)
    rnn = dict(rnn)
    params['network']['separate'] = rnn.pop('separate', False)   # (each trunk with its own RNN: network_builder.py:272-277)
    params['network']['rnn'] = rnn
    torch.manual_seed(11)
    agent = A2CAgent('zoo', copy.deepcopy(params))
    assert agent.is_rnn and agent._engine is None
    agent.init_tensors()
    agent.obs = agent.env_reset()
    before = {k: v.detach().clone() for k, v in agent.model.state_dict().items() if v.is_floating_point()}
    for epoch in range(3):
        first = agent._mb_index % agent._mb_scalars.shape[0]
        agent.update_epoch()
        out = agent.train_epoch()
        for x in out[4] + out[5] + out[7]:
            assert torch.isfinite(x).all()
    # (the third epoch: by then the observation statistics - which the training forward updates BEFORE it normalises, and
    #  which feed the heads directly under concat_output - have settled; in the very first minibatch they jump from (0, 1)
    #  to the batch's moments, in the reference as here)
    first_kl = float(agent._mb_scalars[first, 4])
    assert 0.0 <= first_kl < 0.05, first_kl
    after = agent.model.state_dict()
    assert any(not torch.equal(after[k], v) for k, v in before.items() if 'running' not in k)


def test_checkpoint_round_trip(tmp_path):
    from rl_games_amd import configs
    from rl_games_amd.agent import A2CAgent
    params = configs.tiny(num_actors=64, horizon=8)
    a1 = A2CAgent('t', copy.deepcopy(params))
    a1.init_tensors()
    a1.obs = a1.env_reset()
    a1.update_epoch()
    a1.train_epoch()
    path = a1.save(str(tmp_path / 'ckpt'))
    a2 = A2CAgent('t', copy.deepcopy(params))
    a2.restore(path)
    for (k1, v1), (k2, v2) in zip(a1.model.state_dict().items(), a2.model.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2), k1
    assert a2.epoch_num == a1.epoch_num
    assert torch.equal(a1.optimizer.exp_avg, a2.optimizer.exp_avg)
    assert a2.optimizer.step_count == a1.optimizer.step_count
    ck = torch.load(path, map_location='cpu', weights_only=False)
    assert set(ck) >= {'model', 'optimizer', 'epoch', 'frame', 'last_mean_rewards'}
    assert ck['model']['running_mean_std.running_mean'].dtype == torch.float64
    assert ck['model']['running_mean_std.count'].dtype == torch.int64


def test_manual_mlp_engine_matches_autograd_gradients():
    """Same parameters, same minibatch: the hand-written backward (mlp_engine.ManualMLP + fused
    act-backward/colsum kernel + head-bias sums from the loss kernel) produces the gradients
    torch autograd produces through nn.Linear / ELU."""
    from rl_games_amd import configs
    from rl_games_amd.agent import A2CAgent
    base = configs.humanoid_65536(num_actors=512, minibatch_size=4096, grad_norm=1e9, lr_schedule=None,
                                  learning_rate=0.0)
    torch.manual_seed(0)
    a1 = A2CAgent('eng', copy.deepcopy(base))
    p2 = copy.deepcopy(base)
    p2['config']['manual_mlp'] = False
    a2 = A2CAgent('auto', p2)
    assert a1._engine is not None and a2._engine is None
    a2.model.load_state_dict(a1.model.state_dict())
    a1.init_tensors()
    a1.obs = a1.env_reset()
    a1.set_eval()
    with torch.no_grad():
        batch = a1.play_steps()
    a2.init_tensors()
    snapshot = {k: v.detach().clone() for k, v in a1.model.state_dict().items()}
    grads = []
    for ag in (a1, a2):
        b = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in batch.items() if k != '_fused'}
        ag.model.load_state_dict(snapshot)
        ag.set_train()
        ag.prepare_dataset(b)
        ag.train_actor_critic(ag.dataset[1])
        grads.append({n: p.grad.detach().clone() for n, p in ag.model.named_parameters()})
        res = ag.train_result
        grads[-1]['_scalars'] = torch.stack([res[0], res[1], res[2], res[3], res[8]])
    g1, g2 = grads
    # the f32-MFMA weight-gradient launch for every layer, not a fallback
    assert a1._engine.last_dw_path == 'mfma' and a1._engine.last_dw_library_jobs == 0
    assert torch.allclose(g1.pop('_scalars'), g2.pop('_scalars'), rtol=1e-6, atol=1e-8)
    for n in g2:
        scale = g2[n].abs().max().item() + 1e-12
        assert torch.allclose(g1[n], g2[n], rtol=1e-4, atol=2e-6 * scale), (n, (g1[n] - g2[n]).abs().max().item(), scale)


def test_fused_rollout_step_matches_model_forward():
    """The fused rollout path (obs normalise -> engine GEMMs -> policy-head kernel writing into the
    buffer) stores exactly what model(eval) + update_data would store for the same noise."""
    from rl_games_amd import configs
    from rl_games_amd.agent import A2CAgent
    params = configs.tiny(num_actors=300, horizon=4, obs_dim=10, act_dim=5)
    agent = A2CAgent('t', copy.deepcopy(params))
    agent.init_tensors()
    agent.obs = agent.env_reset()
    # make the statistics non-trivial
    agent.model.running_mean_std.running_mean.normal_()
    agent.model.running_mean_std.running_var.uniform_(0.5, 2.0)
    agent.model.value_mean_std.running_mean.fill_(3.0)
    agent.model.value_mean_std.running_var.fill_(4.0)
    agent.set_eval()
    with torch.no_grad():
        res = agent._fast_policy_step(2)
        noise = agent._roll_noise.clone()
        ref = agent.model({'is_train': False, 'prev_actions': None, 'obs': agent.obs['obs'], 'rnn_states': None})
        mu, sigma = ref['mus'], ref['sigmas']
        action = mu + sigma * noise
        nlp = agent.model.neglogp(action, mu, sigma, torch.log(sigma))
    tb = agent.experience_buffer.tensor_dict
    assert torch.allclose(tb['mus'][2], mu, rtol=1e-5, atol=1e-6)
    assert torch.allclose(tb['sigmas'][2], sigma, rtol=1e-6)
    assert torch.allclose(tb['actions'][2], action, rtol=1e-5, atol=1e-6)
    assert torch.allclose(tb['neglogpacs'][2], nlp, rtol=1e-5, atol=1e-5)
    assert torch.allclose(tb['values'][2], ref['values'], rtol=1e-5, atol=1e-5)
    assert torch.equal(tb['obses'][2], agent.obs['obs'])
    assert torch.equal(res['actions'], tb['actions'][2]) and torch.equal(res['values'], tb['values'][2])
    # what goes to the env: preprocess_actions (a2c_common.py:725-733) = rescale(clamp(actions, -1, 1)), bit for bit
    from rl_games_amd.agent import rescale_actions
    assert agent.clip_actions
    want = rescale_actions(agent.actions_low, agent.actions_high, torch.clamp(res['actions'], -1.0, 1.0))
    assert torch.equal(res['env_actions'], want)
    assert torch.allclose(agent._fast_values(agent.obs).view(-1, 1), agent.get_values(agent.obs), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('variant,manual_lstm', [('lstm', True), ('lstm', False), ('separate_lstm', False),
                                                 ('separate_gru_layer_norm', False)])
def test_lstm_update_matches_reference_epoch(golden, variant, manual_lstm):
    """BASELINE.json config #5 path (play_steps_rnn / seq_length chunks / done resets) against the
    real reference's LSTM agent on identical rollout tensors and initial rnn states - through the
    sequence-persistent LSTM kernels (manual engine) and through torch autograd (RnnWithDones).  `separate_*`
    (tests/golden/epoch_separate_rnn.pt, round 6): separate actor / critic trunks, each with its own RNN - the actor's
    states in front of the critic's (network_builder.py:372-421)."""
    cap = golden('epoch.pt' if variant == 'lstm' else 'epoch_separate_rnn.pt')[variant]
    agent = _make_agent(cap, manual_lstm=manual_lstm)
    assert agent.is_rnn and (agent._engine is not None) == manual_lstm
    assert len(agent.model.get_default_rnn_state()) == len(cap['batch']['rnn_states'])
    if manual_lstm:
        assert agent._engine.lstm is not None
    agent.model.load_state_dict(cap['state_after_rollout'])
    batch = {k: (v.to(DEV) if isinstance(v, torch.Tensor) else [s.to(DEV) for s in v])
             for k, v in cap['batch'].items()}
    agent.set_train()
    agent.prepare_dataset(batch)
    rows = []
    for mini_ep in range(agent.mini_epochs_num):
        for i in range(len(agent.dataset)):
            a, c, e, kl, lr, lr_mul, mu, sigma, b = agent.train_actor_critic(agent.dataset[i])
            rows.append(torch.stack([a, c, e, kl, b]).clone())
    rows = torch.stack(rows).cpu()
    # 1e-5 like the MLP goldens; the absolute floor covers minibatch means of +-O(1) terms (see
    # tests/test_headline_gpu.py)
    assert torch.allclose(rows[:, 0], cap['a_losses'], rtol=1e-5, atol=2e-6)
    assert torch.allclose(rows[:, 1], cap['c_losses'], rtol=1e-5, atol=2e-6)
    assert torch.allclose(rows[:, 2], cap['entropies'], rtol=1e-5, atol=2e-6)
    kls = rows[:, 3].reshape(agent.mini_epochs_num, len(agent.dataset)).mean(1)
    # KL at 1e-4 like the MLP configurations (tests/test_headline_gpu.py::test_kl_conditioning_fp64_demonstration
    # shows why no pair of fp32 implementations can be held to 1e-5 on this quantity)
    assert torch.allclose(kls, cap['mini_epoch_kls'], rtol=1e-4, atol=2e-7), (kls, cap['mini_epoch_kls'])
    assert agent.optimizer.last_and_next_lr()[1] == cap['lrs'][-1]
    final = agent.model.state_dict()
    for k, v in cap['final_state'].items():
        tol = dict(rtol=1e-3, atol=5e-6) if v.is_floating_point() else dict(rtol=0, atol=0)
        assert torch.allclose(final[k].cpu().to(v.dtype), v, **tol), k


def test_lstm_config5_at_its_own_size_matches_reference_epoch():
    """BASELINE.json config #5 AT ITS OWN SIZE - 4,096 envs x seq_len 16, obs 3, act 1, MLP [64,64] + LSTM 64,
    minibatch 16,384 x 4 mini-epochs = 16 optimiser steps - against one train_epoch of the REAL reference agent
    (tests/golden/lstm_full.pt.gz, written by tests/golden/make_golden.py lstm_full from /root/reference:
    play_steps_rnn a2c_common.py:1071-1202, LSTMWithDones recurrent.py:26-83) on the recorded rollout and rnn
    states, through the sequence-persistent LSTM kernels of the manual engine."""
    import gzip
    import io
    import os
    from conftest import GOLDEN_DIR
    with gzip.open(os.path.join(GOLDEN_DIR, 'lstm_full.pt.gz'), 'rb') as f:
        cap = torch.load(io.BytesIO(f.read()), map_location='cpu', weights_only=False)
    agent = _make_agent(cap, manual_lstm=True)
    assert agent.is_rnn and agent._engine is not None and agent._engine.lstm is not None
    assert (agent.num_actors, agent.horizon_length, agent.seq_length, agent.minibatch_size) == (4096, 16, 16, 16384)
    agent.model.load_state_dict(cap['state_after_rollout'])
    batch = {k: (v.to(DEV) if isinstance(v, torch.Tensor) else [s.to(DEV) for s in v])
             for k, v in cap['batch'].items()}
    agent.set_train()
    agent.prepare_dataset(batch)
    rows = []
    for mini_ep in range(agent.mini_epochs_num):
        for i in range(len(agent.dataset)):
            a, c, e, kl, lr, lr_mul, mu, sigma, b = agent.train_actor_critic(agent.dataset[i])
            rows.append(torch.stack([a, c, e, kl, b]).clone())
    rows = torch.stack(rows).cpu()
    assert rows.shape[0] == 16
    assert torch.allclose(rows[:, 0], cap['a_losses'], rtol=1e-5, atol=2e-6), (rows[:, 0] - cap['a_losses']).abs().max()
    assert torch.allclose(rows[:, 1], cap['c_losses'], rtol=1e-5, atol=2e-6), (rows[:, 1] - cap['c_losses']).abs().max()
    assert torch.allclose(rows[:, 2], cap['entropies'], rtol=1e-5, atol=2e-6)
    assert torch.allclose(rows[:, 4], cap['b_losses'], rtol=1e-5, atol=1e-7)
    kls = rows[:, 3].reshape(agent.mini_epochs_num, len(agent.dataset)).mean(1)
    assert torch.allclose(kls, cap['mini_epoch_kls'], rtol=1e-4, atol=2e-7), (kls, cap['mini_epoch_kls'])
    # the learning-rate trajectory of the 16 steps (update_lr calls of the reference), bit for bit
    assert agent.optimizer.last_and_next_lr()[1] == cap['lrs'][-1]


def test_lstm_engine_matches_autograd_gradients_and_rollout():
    """LSTM policy: (i) the engine's rollout step (fused head + persistent LSTM kernel, T = 1) leaves
    the same buffer contents as the torch model path given the same noise-free quantities, and
    (ii) for one minibatch the hand-written BPTT produces autograd's gradients."""
    from rl_games_amd import configs
    from rl_games_amd.agent import A2CAgent
    base = configs.pendulum_lstm_4096(num_actors=128, minibatch_size=1024, grad_norm=1e9, lr_schedule=None,
                                      learning_rate=0.0)
    base['config']['env_config']['p_done'] = 0.2          # plenty of mid-sequence resets
    torch.manual_seed(0)
    a1 = A2CAgent('eng', copy.deepcopy(base))
    p2 = copy.deepcopy(base)
    p2['config']['manual_lstm'] = False
    a2 = A2CAgent('auto', p2)
    assert a1._engine is not None and a1._engine.lstm is not None and a2._engine is None
    a2.model.load_state_dict(a1.model.state_dict())
    a1.init_tensors()
    a1.obs = a1.env_reset()
    a1.set_eval()
    with torch.no_grad():
        batch = a1.play_steps_rnn()
    # (i) values / mus stored by the fused rollout == the torch model evaluated on the stored
    # observations with the stored initial states, sequence by sequence
    a2.init_tensors()
    a2.set_eval()
    H, N = a1.horizon_length, a1.num_actors
    obs = batch['obses'].reshape(N, H, -1)
    with torch.no_grad():
        st = [s[:, :N].contiguous() for s in batch['rnn_states']]     # states at t = 0 (one seq per env)
        for t in range(H):
            keep = (1.0 - a1.experience_buffer.tensor_dict['dones'][t].float()).reshape(1, -1, 1)
            st = [s * keep for s in st] if t > 0 else st
            res = a2.model({'is_train': False, 'obs': obs[:, t], 'rnn_states': st})
            st = res['rnn_states']
            assert torch.allclose(res['mus'], batch['mus'].reshape(N, H, -1)[:, t], rtol=1e-4, atol=2e-6), t
            assert torch.allclose(res['values'], batch['values'].reshape(N, H, 1)[:, t], rtol=1e-4, atol=2e-5), t
    # (ii) gradients
    snapshot = {k: v.detach().clone() for k, v in a1.model.state_dict().items()}
    grads = []
    for ag in (a1, a2):
        b = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in batch.items() if k != '_fused'}
        ag.model.load_state_dict(snapshot)
        ag.set_train()
        ag.prepare_dataset(b)
        ag.train_actor_critic(ag.dataset[1])
        grads.append({n: p.grad.detach().clone() for n, p in ag.model.named_parameters()})
        res = ag.train_result
        grads[-1]['_scalars'] = torch.stack([res[0], res[1], res[2], res[3], res[8]])
    g1, g2 = grads
    # heads, W_ih, W_hh and the second trunk layer through the MFMA launch; the [64 x 3] first layer (3
    # observations: not a multiple of 4) through the narrow weight-gradient kernel - no library GEMM left
    assert a1._engine.last_dw_path == 'mfma' and a1._engine.last_dw_library_jobs == 0
    assert torch.allclose(g1.pop('_scalars'), g2.pop('_scalars'), rtol=1e-5, atol=1e-7)
    for n in g2:
        scale = g2[n].abs().max().item() + 1e-12
        assert torch.allclose(g1[n], g2[n], rtol=1e-4, atol=5e-6 * scale), (n, (g1[n] - g2[n]).abs().max().item(), scale)


@pytest.mark.parametrize('manual_lstm', [True, False])
def test_lstm_config_train_epoch_runs(manual_lstm):
    from rl_games_amd import configs
    from rl_games_amd.agent import A2CAgent
    params = configs.pendulum_lstm_4096(num_actors=256, manual_lstm=manual_lstm)
    agent = A2CAgent('lstm', params)
    agent.init_tensors()
    agent.obs = agent.env_reset()
    for _ in range(3):                      # 1 eager epoch, then HIP-graph replays on the engine path
        agent.update_epoch()
        out = agent.train_epoch()
    assert len(out[4]) == agent.mini_epochs_num * agent.num_minibatches
    assert all(torch.isfinite(x).item() for x in out[4])
    st = agent.dataset.values_dict
    assert st is not None and st['rnn_states'][0].shape == (1, 256 * (16 // 16), 64)


def test_masked_rows_path_matches_oracle():
    """next_step-autoreset masking (SURVEY 8a' pitfall 9; a2c_common.py:1605-1615, torch_ext.py:157-191):
    rnn_masks in the batch -> valid-row value statistics, masked advantage normalisation, masked
    loss / KL means.  Same rollout tensors + mask through the device agent and the CPU oracle."""
    from rl_games_amd import configs
    from rl_games_amd.agent import A2CAgent
    params = configs.tiny(num_actors=96, horizon=8, obs_dim=9, act_dim=3, hip_graphs=False)
    agent = A2CAgent('m', copy.deepcopy(params))
    agent.init_tensors()
    agent.obs = agent.env_reset()
    agent.set_eval()
    with torch.no_grad():
        batch = agent.play_steps()
    B = 96 * 8
    mask = (torch.rand(B, generator=torch.Generator().manual_seed(3)) < 0.75).float()
    batch = {k: v for k, v in batch.items() if k != '_fused'}
    batch['rnn_masks'] = mask.to(DEV)
    env = SyntheticTensorEnv(96, 9, 3, device='cpu', seed=1)
    oracle = OracleAgent(copy.deepcopy(params), env)
    oracle.model.load_full_state_dict({k: v.detach().cpu().clone() for k, v in agent.model.state_dict().items()})
    cpu_batch = {k: v.detach().cpu().clone() for k, v in batch.items() if isinstance(v, torch.Tensor)}
    ref = oracle.update(cpu_batch)
    agent.set_train()
    agent.prepare_dataset(batch)
    vd = agent.dataset.values_dict
    assert torch.allclose(vd['old_values'].cpu(), oracle_ds(oracle, 'old_values'), rtol=1e-5, atol=2e-6)
    assert torch.allclose(vd['returns'].cpu(), oracle_ds(oracle, 'returns'), rtol=1e-5, atol=2e-6)
    k = 0
    for mini_ep in range(agent.mini_epochs_num):
        for i in range(len(agent.dataset)):
            a, c, e, kl, lr, lr_mul, mu, sigma, b = agent.train_actor_critic(agent.dataset[i])
            r = ref[k]
            for got, key in ((a, 'a_loss'), (c, 'c_loss'), (e, 'entropy'), (kl, 'kl'), (b, 'b_loss')):
                assert np.isclose(got.item(), r[key].item(), rtol=1e-5, atol=2e-6), (k, key, got.item(), r[key].item())
            k += 1
    assert agent.model.value_mean_std.count.item() == oracle.model.value_stats['count'].item()
    assert agent.model.value_mean_std.count.item() == 1 + 2 * int(mask.sum().item())


def oracle_ds(oracle, key):
    # the oracle keeps the first-epoch dataset tensors; mu/sigma are updated in place, the rest is static
    return oracle.dataset[key]


def test_restores_checkpoint_written_by_the_reference(golden):
    """tests/golden/ref_checkpoint.pth was written by the real reference A2CAgent.save()
    (make_golden.py, section `checkpoint`): our agent restores weights, normaliser statistics,
    Adam moments/step, epoch/frame counters and keeps training from it."""
    import os
    from rl_games_amd.agent import A2CAgent
    meta = golden('ref_checkpoint_meta.pt')
    path = os.path.join(os.path.dirname(__file__), 'golden', 'ref_checkpoint.pth')
    ck = torch.load(path, map_location='cpu', weights_only=False)
    params = copy.deepcopy(meta['params'])
    params['config']['device'] = DEV
    env = SyntheticTensorEnv(meta['env']['num_envs'], meta['env']['obs_dim'], meta['env']['act_dim'],
                             device=DEV, seed=meta['env']['seed'])
    params['config']['vec_env'] = env
    params['config']['env_info'] = env.get_env_info()
    agent = A2CAgent('interop', params)
    agent.restore(path)
    sd = agent.model.state_dict()
    assert list(sd.keys()) == list(ck['model'].keys())
    for k, v in ck['model'].items():
        assert sd[k].dtype == v.dtype and torch.equal(sd[k].cpu(), v), k
    assert agent.epoch_num == ck['epoch'] and agent.frame == ck['frame']
    assert agent.last_mean_rewards == ck['last_mean_rewards']
    opt = agent.optimizer.state_dict()
    assert opt['param_groups'][0]['lr'] == ck['optimizer']['param_groups'][0]['lr']
    for i, st in ck['optimizer']['state'].items():
        assert torch.equal(opt['state'][i]['exp_avg'].cpu(), st['exp_avg']), i
        assert torch.equal(opt['state'][i]['exp_avg_sq'].cpu(), st['exp_avg_sq']), i
        assert int(float(opt['state'][i]['step'])) == int(float(st['step']))
    # and the restored agent trains on
    agent.init_tensors()
    agent.obs = agent.env_reset()
    agent.update_epoch()
    res = agent.train_epoch()
    assert all(torch.isfinite(x) for x in res[4] + res[5])
    assert agent.optimizer.step_count == int(float(ck['optimizer']['state'][0]['step'])) + len(res[4])


def test_two_rank_bench_on_one_gpu():
    """The multi-rank agent path end to end (device-side lr from the all-reduced KL, eager gradient
    all-reduce between the two HIP-graph replays, pooled running-statistics merge, parameter
    broadcast) with 2 ranks sharing this box's single GPU: RLG_TEST_SINGLE_GPU=1 switches the
    collectives to gloo (RCCL refuses two ranks per device); everything else is the production path
    of `bench.py --gpus 2`."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RLG_TEST_SINGLE_GPU='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(root, 'bench.py'),
           '--gpus', '2', '--steps', '1', '--warmup', '2']
    res = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    line = [l for l in res.stdout.splitlines() if l.startswith('{')][-1]
    out = json.loads(line)
    assert out['n_gpus'] == 2 and out['config']['envs_per_gpu'] == 32768
    assert out['config']['ranks_in_sync'] is True
    assert out['value'] > 0 and out['roofline']['launches'] == 1


def _run_two_ranks(script_args, timeout=600, nproc=2):
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RLG_TEST_SINGLE_GPU='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc),
           '--master-addr', '127.0.0.1', '--master-port', str(port)] + script_args
    return subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize('nproc,two_phase', [(2, False), (4, False), (2, True), (4, True), (3, True)])
def test_ipc_allreduce_between_processes_on_one_gpu(nproc, two_phase):
    """The in-graph all-reduce kernels (hipIpc-mapped peer staging buffers, device-side flags) between
    2 - 4 processes sharing this box's GPU: eager and HIP-graph-replayed launches, every rank
    bit-identical to the rank-ordered fp32 sum - the one-shot kernel and the reduce-scatter + all-gather one."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.environ['RLG_IPC_CHECK_TWO_PHASE'] = '1' if two_phase else '0'
    try:
        res = _run_two_ranks([os.path.join(root, 'tools', 'two_rank_ipc_check.py')], nproc=nproc)
    finally:
        os.environ.pop('RLG_IPC_CHECK_TWO_PHASE', None)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert 'IPC_ALLREDUCE_CHECK ok' in res.stdout and f'two_phase {two_phase}' in res.stdout


@pytest.mark.parametrize('two_phase', [False, True])
def test_ipc_allreduce_is_fail_safe_when_a_peer_never_arrives(two_phase):
    """A rank skips a collective (2 processes on one GPU, 2 s bound): the waiting rank's launch gives up, leaves ZEROS
    in the gradients (not a sum of stale staging data), sets the sticky error word, the Adam launch behind it
    skips its step - parameters, moments and learning rate untouched - and later launches fail fast."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.environ['RLG_IPC_CHECK_TWO_PHASE'] = '1' if two_phase else '0'
    try:
        res = _run_two_ranks([os.path.join(root, 'tools', 'two_rank_ipc_failsafe.py')], nproc=2, timeout=300)
    finally:
        os.environ.pop('RLG_IPC_CHECK_TWO_PHASE', None)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert 'IPC_FAILSAFE_CHECK ok' in res.stdout


@pytest.mark.parametrize('variant', ['experimental_cv', 'no_actor_value_loss', 'rnn_actor_mlp_critic',
                                     'rnn_actor_rnn_critic', 'rnn_actor_gru_critic_layer_norm',
                                     'three_agents', 'two_agents_recurrent'])
def test_central_value_update_matches_reference_epoch(golden, variant):
    """Central (asymmetric) value function: the update phase against golden vectors
    of the REAL reference agent with `central_value_config` - critic minibatches first
    (CentralValueTrain.train_net), then the actor's, on identical rollout tensors.  `rnn_*`
    (tests/golden/central_value_rnn.pt, round 6): a recurrent actor, and critics with an RNN of their own - sequence
    minibatches that start from the states the rollout kept (central_value.py:163-170).  `*_agents*`
    (central_value_multi_agent.pt): several agents per env - one state row per env, the critic's dataset is agent 0's
    values and returns in env-major order (update_multiagent_tensors, :225-234)."""
    from rl_games_amd.agent import A2CAgent
    recurrent = variant.startswith('rnn_') or variant.endswith('_recurrent')
    fixture = ('central_value_multi_agent.pt' if 'agents' in variant else
               'central_value_rnn.pt' if recurrent else 'central_value.pt')
    cap = golden(fixture)[variant]
    params = copy.deepcopy(cap['params'])
    params['config']['device'] = DEV
    env = SyntheticTensorEnv(cap['env']['num_envs'], cap['env']['obs_dim'], cap['env']['act_dim'], device=DEV,
                             seed=cap['env']['seed'], state_dim=cap['env']['state_dim'], agents=cap['env'].get('agents', 1))
    params['config']['vec_env'] = env
    params['config']['env_info'] = env.get_env_info()
    agent = A2CAgent('cv', params)
    assert agent.num_agents == cap['env'].get('agents', 1)
    assert agent.has_central_value
    assert agent.has_value_loss == cap['params']['config'].get('use_experimental_cv', True)
    agent.init_tensors()
    agent.model.load_state_dict(cap['state_after_rollout'])
    agent.central_value_net.load_state_dict(cap['cv_state_after_rollout'])
    assert set(agent.central_value_net.state_dict().keys()) == set(cap['cv_state_after_rollout'].keys())
    batch = {k: ([s.to(DEV) for s in v] if isinstance(v, (list, tuple)) else v.to(DEV)) for k, v in cap['batch'].items()}
    cv = agent.central_value_net
    assert agent.is_rnn == recurrent and cv.is_rnn == ('cv_mb_rnn_states' in cap)
    if cv.is_rnn:                               # the states the critic's RNN had at the start of every rollout sequence
        for dst, src in zip(cv.mb_rnn_states, cap['cv_mb_rnn_states']):
            assert dst.shape == src.shape
            dst.copy_(src)
    agent.set_train()
    agent.epoch_num = 1
    agent.prepare_dataset(batch)
    ds, vd = cap['dataset'], agent.dataset.values_dict
    for k in ('old_values', 'returns', 'advantages'):
        assert torch.allclose(vd[k].cpu().reshape(ds[k].shape), ds[k], rtol=1e-5, atol=1e-6), k
    if 'cv_dataset' in cap:                     # what the critic trains on
        cvd = cv.dataset.values_dict
        for k, want in cap['cv_dataset'].items():
            got = cvd[k].cpu().reshape(want.shape)
            assert torch.allclose(got.to(want.dtype), want, rtol=1e-5, atol=1e-6), k
    # (round 5: the critic's MLP on the fused chain kernels, not autograd; a recurrent critic is a torch module)
    assert (cv._engine is not None) == (not cv.is_rnn)
    cv.train_net()
    assert cv.is_rnn or cv._engine.last_dw_path is not None
    n_cv = cv.mini_epoch * cv.num_minibatches
    assert torch.allclose(cv._rows[:n_cv, 5].cpu(), cap['cv_losses'], rtol=1e-5, atol=1e-6)
    agent.set_train()
    rows = []
    for mini_ep in range(agent.mini_epochs_num):
        for i in range(len(agent.dataset)):
            a, c, e, kl, lr, lr_mul, mu, sigma, b = agent.train_actor_critic(agent.dataset[i])
            rows.append(torch.stack([a, c.reshape(()), e, kl]).clone())
    rows = torch.stack(rows).cpu()
    assert torch.allclose(rows[:, 0], cap['a_losses'], rtol=1e-5, atol=2e-6)
    assert torch.allclose(rows[:, 1], cap['c_losses'], rtol=1e-5, atol=2e-6)
    assert torch.allclose(rows[:, 2], cap['entropies'], rtol=1e-5, atol=2e-6)
    kls = rows[:, 3].reshape(agent.mini_epochs_num, -1).mean(1)
    assert torch.allclose(kls, cap['mini_epoch_kls'], rtol=1e-4, atol=1e-7)
    assert agent.optimizer.last_and_next_lr()[1] == cap['lrs'][-1]
    for final, want in ((agent.model.state_dict(), cap['final_state']),
                        (cv.state_dict(), cap['cv_final_state'])):
        for k, v in want.items():
            tol = dict(rtol=1e-4, atol=2e-6) if v.is_floating_point() else dict(rtol=0, atol=0)
            assert torch.allclose(final[k].cpu().to(v.dtype), v, **tol), k


@pytest.mark.parametrize('recurrent', [False, True])
def test_multi_agent_central_value_train_epochs_run(recurrent):
    """Rollout side of a multi-agent central value function (central_value.py:223-225): one privileged state row per
    env, every agent of the env gets the env's value; epochs run with finite losses and the critic's dataset holds one
    row per env and step."""
    from rl_games_amd import configs
    from rl_games_amd.agent import A2CAgent
    params = configs.tiny(num_actors=32, horizon=8, obs_dim=10, act_dim=3, seq_length=4)
    cv_net = {'name': 'actor_critic', 'central_value': True,
              'mlp': {'units': [24, 16], 'activation': 'elu', 'initializer': {'name': 'default'}}}
    if recurrent:
        params['network']['rnn'] = {'name': 'lstm', 'units': 16, 'layers': 1}
        cv_net['rnn'] = {'name': 'lstm', 'units': 12, 'layers': 1}
    params['config']['central_value_config'] = {
        'minibatch_size': 64, 'mini_epochs': 2, 'learning_rate': 5e-4, 'clip_value': True, 'normalize_input': True,
        'truncate_grads': True, 'grad_norm': 1.0, 'network': cv_net}
    params['config']['env_config'].update(state_dim=9, agents=3)
    torch.manual_seed(3)
    agent = A2CAgent('macv', copy.deepcopy(params))
    cv = agent.central_value_net
    assert agent.num_agents == 3 and cv.num_agents == 3 and cv.batch_size == 32 * 8 and agent.batch_size == 32 * 3 * 8
    agent.init_tensors()
    agent.obs = agent.env_reset()
    for _ in range(2):
        agent.update_epoch()
        out = agent.train_epoch()
        assert all(torch.isfinite(x).all() for x in out[4] + out[5] + out[7])
    values = agent.experience_buffer.tensor_dict['values'].reshape(8, 32, 3)
    assert torch.equal(values[:, :, 0], values[:, :, 1]) and torch.equal(values[:, :, 0], values[:, :, 2])
    assert agent.experience_buffer.tensor_dict['states'].shape[:2] == (8, 32)
    vd = cv.dataset.values_dict
    assert vd['old_values'].shape[0] == vd['returns'].shape[0] == vd['obs'].shape[0] == 32 * 8
    if recurrent:
        assert cv.rnn_states[0].shape == (1, 32, 12) and vd['rnn_states'][0].shape == (1, 32 * 8 // 4, 12)


def test_recurrent_actor_and_critic_train_epochs_run():
    """Rollout side of a recurrent central value network: its states advance with play_steps_rnn, are kept at every
    sequence start (pre_step_rnn), are zero where an episode just ended, and the privileged states reach the buffer;
    three epochs with finite losses and a critic whose RNN weights move."""
    from rl_games_amd import configs
    from rl_games_amd.agent import A2CAgent
    params = configs.tiny(num_actors=64, horizon=8, obs_dim=10, act_dim=3, seq_length=4)
    params['network']['rnn'] = {'name': 'lstm', 'units': 16, 'layers': 1}
    params['config']['central_value_config'] = {
        'minibatch_size': 128, 'mini_epochs': 2, 'learning_rate': 5e-4, 'clip_value': True, 'normalize_input': True,
        'truncate_grads': True, 'grad_norm': 1.0,
        'network': {'name': 'actor_critic', 'central_value': True, 'rnn': {'name': 'gru', 'units': 12, 'layers': 1},
                    'mlp': {'units': [24, 16], 'activation': 'elu', 'initializer': {'name': 'default'}}}}
    params['config']['env_config'].update(state_dim=9, p_done=0.2)
    torch.manual_seed(3)
    agent = A2CAgent('rcv', copy.deepcopy(params))
    cv = agent.central_value_net
    assert agent.is_rnn and cv.is_rnn and cv._engine is None and len(cv.rnn_states) == 1
    agent.init_tensors()
    agent.obs = agent.env_reset()
    before = {k: v.detach().clone() for k, v in cv.state_dict().items()}
    for _ in range(3):
        agent.update_epoch()
        out = agent.train_epoch()
        assert all(torch.isfinite(x).all() for x in out[4] + out[5] + out[7])
    # the state kept for the rollout's second sequence (step 4): zero where step 3 ended an episode (the buffer's dones of
    # step n are the flags the env returned at step n - 1), the critic's running state elsewhere
    done = agent.experience_buffer.tensor_dict['dones'][4].bool()
    assert done.any() and not done.all()
    kept = cv.mb_rnn_states[0][1]
    assert kept[:, done].abs().max() == 0 and (kept[:, ~done].abs().amax(dim=(0, 2)) > 0).all()
    assert agent.experience_buffer.tensor_dict['states'].abs().max() > 0
    vd = cv.dataset.values_dict
    assert vd['rnn_states'][0].shape == (1, 64 * 8 // 4, 12)
    after = cv.state_dict()
    assert any('rnn' in k and not torch.equal(after[k], v) for k, v in before.items())


@pytest.mark.parametrize('state_dim,units,envs,mb', [(9, [32, 16], 64, 256), (24, [64, 32, 16], 64, 256),
                                                     (48, [256, 128, 64], 2048, 16384)])
def test_central_value_chain_gradients_equal_autograd(state_dim, units, envs, mb):
    """The central value network on the fused chain kernels (chain_net.ChainNet: one forward launch, one backward
    launch, MFMA weight gradients where a layer's input width is a multiple of 4) against the autograd path it replaces
    (`fused_mlp: False`), same weights, same minibatch: values, loss, every gradient to 1e-5 of its scale, the state
    statistics bit for bit, and the parameters behind the optimiser step.  The 16,384-row case runs the split-bf16 chain
    kernels and the split-product weight-gradient launch with the one-column head as the chain's last layer."""
    from rl_games_amd import configs
    from rl_games_amd.agent import A2CAgent
    res = {}
    for fused in (True, False):
        params = configs.tiny(num_actors=envs, horizon=8)
        params['config']['central_value_config'] = {
            'minibatch_size': mb, 'mini_epochs': 1, 'learning_rate': 5e-4, 'clip_value': True, 'normalize_input': True,
            'truncate_grads': True, 'grad_norm': 1.0, 'fused_mlp': fused,
            'network': {'name': 'actor_critic', 'central_value': True,
                        'mlp': {'units': units, 'activation': 'elu', 'initializer': {'name': 'default'}}}}
        params['config']['env_config']['state_dim'] = state_dim
        torch.manual_seed(21)
        agent = A2CAgent('cvg', copy.deepcopy(params))
        agent.init_tensors()
        agent.obs = agent.env_reset()
        agent.set_eval()
        with torch.no_grad():
            batch = agent.play_steps()
        agent.set_train()
        agent.prepare_dataset(batch)
        cv = agent.central_value_net
        assert (cv._engine is not None) == fused
        if fused and mb >= 16384:
            assert cv._engine.chain.split_products(mb, 0) and cv._engine.chain.split_products(mb, 1)
        loss = cv.train_critic(cv.dataset[0]).clone()
        grads = {n: p.grad.clone() for n, p in cv.model.named_parameters()}       # (clipped, as the Adam launch leaves them)
        res[fused] = (loss, grads, {n: p.detach().clone() for n, p in cv.model.named_parameters()},
                      cv.model.running_mean_std.running_mean.clone(), cv.model.running_mean_std.count.clone())
    a, b = res[True], res[False]
    assert torch.allclose(a[0], b[0], rtol=1e-5, atol=1e-7)
    for n in a[1]:
        scale = b[1][n].abs().max().item()
        assert (a[1][n] - b[1][n]).abs().max().item() <= 1e-5 * scale + 1e-9, n
        # the first Adam step is lr * sign(g): compare it where the gradient is above rounding noise
        solid = b[1][n].abs() > 1e-3 * scale
        assert torch.allclose(a[2][n][solid], b[2][n][solid], rtol=1e-4, atol=1e-6), n
        assert (a[2][n] - b[2][n]).abs().max().item() <= 2.0 * 5e-4 * (1 + 1e-5), n
    assert torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])


def test_central_value_train_epoch_and_checkpoint(tmp_path):
    from rl_games_amd import configs
    from rl_games_amd.agent import A2CAgent
    params = configs.tiny(num_actors=64, horizon=8)
    params['config']['central_value_config'] = {
        'minibatch_size': 128, 'mini_epochs': 2, 'learning_rate': 5e-4, 'clip_value': True,
        'normalize_input': True, 'truncate_grads': True, 'grad_norm': 1.0,
        'network': {'name': 'actor_critic', 'central_value': True,
                    'mlp': {'units': [32, 16], 'activation': 'elu', 'initializer': {'name': 'default'}}}}
    params['config']['env_config']['state_dim'] = 9
    a1 = A2CAgent('cv', copy.deepcopy(params))
    a1.init_tensors()
    a1.obs = a1.env_reset()
    for _ in range(2):
        a1.update_epoch()
        out = a1.train_epoch()
    assert all(torch.isfinite(x) for x in out[4] + out[5])
    assert a1.experience_buffer.tensor_dict['states'].shape == (8, 64, 9)
    cvm = a1.central_value_net.model
    assert cvm.running_mean_std.count.item() == 1 + 2 * 2 * 64 * 8        # states: per CV minibatch pass
    assert cvm.value_mean_std.count.item() == 1 + 2 * 2 * 64 * 8          # values + returns per epoch
    path = a1.save(str(tmp_path / 'cv_ckpt'))
    ck = torch.load(path, map_location='cpu', weights_only=False)
    assert 'assymetric_vf_nets' in ck and 'assymetric_vf_optimizer' in ck
    a2 = A2CAgent('cv', copy.deepcopy(params))
    a2.restore(path)
    for (k1, v1), (k2, v2) in zip(a1.central_value_net.state_dict().items(),
                                  a2.central_value_net.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2), k1
    assert torch.equal(a1.central_value_net.optimizer.exp_avg, a2.central_value_net.optimizer.exp_avg)


def test_player_restores_checkpoint_and_plays(tmp_path, golden):
    """Train -> save -> PpoPlayerContinuous.restore -> deterministic actions are the policy means
    (rescaled), for our own checkpoint and for the reference-written one; run() plays episodes."""
    import os
    from rl_games_amd import configs
    from rl_games_amd.agent import A2CAgent, rescale_actions
    from rl_games_amd.player import PpoPlayerContinuous
    params = configs.tiny(num_actors=64, horizon=8)
    agent = A2CAgent('t', copy.deepcopy(params))
    agent.init_tensors()
    agent.obs = agent.env_reset()
    agent.update_epoch()
    agent.train_epoch()
    path = agent.save(str(tmp_path / 'play_ckpt'))
    pp = copy.deepcopy(params)
    pp['config']['player'] = {'games_num': 20, 'print_stats': False}
    player = PpoPlayerContinuous(pp)
    player.restore(path)
    obs = agent.obs['obs']
    player.has_batch_dimension = True
    act = player.get_action(obs, is_deterministic=True)
    agent.set_eval()
    mu = agent.get_action_values(agent.obs)['mus']
    want = rescale_actions(agent.actions_low, agent.actions_high, torch.clamp(mu, -1.0, 1.0))
    assert torch.allclose(act, want, rtol=1e-6, atol=1e-7)
    mean_r, mean_n = player.run()
    assert np.isfinite(mean_r) and mean_n > 0
    # a checkpoint written by the reference agent
    meta = golden('ref_checkpoint_meta.pt')
    rp = copy.deepcopy(meta['params'])
    rp['config']['device'] = DEV
    rp['config']['env_config'].update(seed=5)
    rplayer = PpoPlayerContinuous(rp)
    rplayer.restore(os.path.join(os.path.dirname(__file__), 'golden', 'ref_checkpoint.pth'))
    ck = torch.load(os.path.join(os.path.dirname(__file__), 'golden', 'ref_checkpoint.pth'), map_location='cpu',
                    weights_only=False)
    assert torch.equal(rplayer.model.state_dict()['a2c_network.mu.weight'].cpu(), ck['model']['a2c_network.mu.weight'])
    rplayer.has_batch_dimension = True
    a = rplayer.get_action(torch.randn(7, meta['env']['obs_dim'], device=DEV), True)
    assert a.shape == (7, meta['env']['act_dim']) and torch.isfinite(a).all() and a.abs().max() <= 1.0


def test_train_loop_runs_to_max_epochs_with_checkpoints_and_observer(tmp_path):
    """The Runner's entry point: agent.train() -> (last_mean_rewards, epoch_num), with periodic and
    best checkpoints, observer callbacks and a stop after max_epochs (a2c_common.py:1662-1782)."""
    import os
    from rl_games_amd import configs
    from rl_games_amd.agent import A2CAgent

    class Observer:
        def __init__(self):
            self.calls = []

        def before_init(self, base_name, config, experiment_name):
            self.calls.append('before_init')

        def after_init(self, algo):
            self.calls.append('after_init')

        def process_infos(self, infos, done_indices):
            self.calls.append('process_infos')

        def after_steps(self):
            self.calls.append('after_steps')

        def after_clear_stats(self):
            self.calls.append('after_clear_stats')

        def after_print_stats(self, frame, epoch_num, total_time):
            self.calls.append(('after_print_stats', frame, epoch_num))

    obs = Observer()
    params = configs.tiny(num_actors=64, horizon=8, max_epochs=6, save_frequency=2, save_best_after=0,
                          train_dir=str(tmp_path), full_experiment_name='loop')
    params['config']['features'] = {'observer': obs}
    params['config']['env_config']['p_done'] = 0.2
    agent = A2CAgent('loop', params)
    last_mean_rewards, epoch_num = agent.train()
    assert epoch_num == 6 and agent.frame == 6 * 64 * 8
    assert np.isfinite(last_mean_rewards)
    nn_dir = os.path.join(str(tmp_path), 'loop', 'nn')
    files = sorted(os.listdir(nn_dir))
    assert any(f.startswith('last_tiny_ep_6') for f in files), files
    assert 'tiny.pth' in files                                              # best-so-far checkpoint
    assert sum(f.startswith('last_tiny_ep_') and '_rew_' in f for f in files) == 3   # epochs 2, 4, 6
    assert obs.calls[:2] == ['before_init', 'after_init']
    assert obs.calls.count('after_steps') == 6
    stats = [c for c in obs.calls if isinstance(c, tuple)]
    assert [c[2] for c in stats] == [1, 2, 3, 4, 5, 6] and stats[-1][1] == 6 * 64 * 8
    # restart from the last checkpoint continues the epoch counter
    p2 = configs.tiny(num_actors=64, horizon=8, max_epochs=8, train_dir=str(tmp_path), full_experiment_name='loop2')
    a2 = A2CAgent('loop2', p2)
    a2.restore(os.path.join(nn_dir, [f for f in files if f.startswith('last_tiny_ep_6') and '_rew_' not in f][0]))
    _, e2 = a2.train()
    assert e2 == 8 and a2.frame == 8 * 64 * 8


def test_multi_agent_env_rows():
    """num_agents > 1 (a2c_common.py:184,1039-1040): every per-step tensor has num_actors * num_agents
    rows, batch_size counts them, the epoch runs and the buffer keeps the env-major row order."""
    from rl_games_amd import configs
    from rl_games_amd.agent import A2CAgent
    params = configs.tiny(num_actors=32, horizon=8, minibatch_size=128)
    params['config']['env_config']['agents'] = 2
    agent = A2CAgent('ma', params)
    assert agent.num_agents == 2 and agent.batch_size == 32 * 2 * 8
    agent.init_tensors()
    agent.obs = agent.env_reset()
    assert agent.obs['obs'].shape[0] == 64
    for _ in range(2):
        agent.update_epoch()
        out = agent.train_epoch()
    assert len(out[4]) == agent.mini_epochs_num * (512 // 128)
    assert all(torch.isfinite(x) for x in out[4] + out[5])
    assert agent.experience_buffer.tensor_dict['obses'].shape == (8, 64, 12)
    assert agent.dataset.values_dict['obs'].shape == (512, 12)


@pytest.mark.parametrize('N,H,obs_dim,act_dim,units,mbs', [
    (37, 5, 7, 3, [20, 12], 37),          # nothing a multiple of 4 or 64: strided GAE, library dW for layer 1
    (96, 12, 16, 2, [64, 32], 288),       # H multiple of 4 but not 16
    (130, 16, 33, 5, [48, 24, 16], 520),  # ragged last GAE tile, odd obs width
    (64, 64, 8, 1, [32], 1024),           # maximum fused horizon, single hidden layer, one action
    (256, 4, 12, 21, [100, 52], 512),     # minimum fused horizon, BASELINE action count
])
def test_odd_shapes_match_oracle_epoch(N, H, obs_dim, act_dim, units, mbs):
    """Shape robustness of the whole path (fused/strided GAE, ragged tiles, MFMA vs library weight
    gradients, head GEMM paths): per-minibatch losses of a full update against the CPU oracle on the
    same rollout, for shapes with no convenient alignment."""
    from rl_games_amd import configs
    from rl_games_amd.agent import A2CAgent
    params = configs.tiny(num_actors=N, horizon=H, obs_dim=obs_dim, act_dim=act_dim, minibatch_size=mbs,
                          seq_length=1)      # PPODataset wants batch % seq_length == 0, like the reference
    params['network']['mlp']['units'] = list(units)
    agent = A2CAgent('odd', copy.deepcopy(params))
    agent.init_tensors()
    agent.obs = agent.env_reset()
    agent.set_eval()
    with torch.no_grad():
        batch = agent.play_steps()
    oracle = OracleAgent(copy.deepcopy(params), SyntheticTensorEnv(N, obs_dim, act_dim, device='cpu', seed=1))
    oracle.model.load_full_state_dict({k: v.detach().cpu().clone() for k, v in agent.model.state_dict().items()})
    ref = oracle.update({k: v.detach().cpu().clone() for k, v in batch.items() if isinstance(v, torch.Tensor)})
    agent.set_train()
    agent.prepare_dataset(batch)
    vd = agent.dataset.values_dict
    assert torch.allclose(vd['advantages'].cpu(), oracle.dataset['advantages'], rtol=1e-5, atol=1e-6)
    k = 0
    for _ in range(agent.mini_epochs_num):
        for i in range(len(agent.dataset)):
            res = agent.train_actor_critic(agent.dataset[i])
            for got, key in ((res[0], 'a_loss'), (res[1], 'c_loss'), (res[2], 'entropy'), (res[3], 'kl'),
                             (res[8], 'b_loss')):
                assert np.isclose(got.item(), ref[k][key].item(), rtol=1e-5, atol=2e-6), (k, key, got.item(), ref[k][key].item())
            k += 1
    assert agent.optimizer.last_and_next_lr()[1] == oracle.lr
    # Parameters: Adam's first updates are +-lr * g/|g|-like, so an element whose gradient is at
    # rounding-noise level can move by a whole lr step in the other direction when the fp32 summation
    # order differs (tests/test_headline_gpu.py::_check_final_params states the same): the bulk agrees,
    # a few elements (zero-initialised biases are all of this kind) may deviate, none by more than the
    # step budget.  The strong check is above: every later minibatch's losses, which are functions of the
    # updated parameters, agree to 1e-5.
    final = agent.model.state_dict()
    want = oracle.model.full_state_dict()
    steps = agent.mini_epochs_num * len(agent.dataset)
    for name, v in want.items():
        if v.is_floating_point():
            got = final[name].cpu().to(v.dtype)
            bad = ~torch.isclose(got, v, rtol=2e-3, atol=1e-5)
            assert bad.sum().item() <= max(2, 0.05 * v.numel()), (name, bad.sum().item(), v.numel())
            assert (got - v).abs().max().item() <= 2.1 * steps * max(oracle.lr, 3e-4), name
    # and two more epochs through the public entry point (HIP graphs from the 2nd on)
    for _ in range(2):
        agent.update_epoch()
        out = agent.train_epoch()
    assert all(torch.isfinite(x) for x in out[4] + out[5])


@pytest.mark.parametrize('cfg', ['mlp', 'lstm'])
def test_rollout_step_graphs_match_eager_rollout(cfg):
    """The per-step rollout HIP graphs (policy forward + fused head + buffer writes + action
    rescale) leave the same buffer contents as the eager fast path: two identically seeded agents,
    one with `rollout_graphs` off, run three epochs; every rollout tensor of the last epoch is
    bit-identical (same kernels, same RNG stream)."""
    from rl_games_amd import configs
    from rl_games_amd.agent import A2CAgent
    outs = []
    for graphs in (True, False):
        if cfg == 'mlp':
            params = configs.tiny(num_actors=96, horizon=8, rollout_graphs=graphs, hip_graphs=True)
        else:
            params = configs.pendulum_lstm_4096(num_actors=64, rollout_graphs=graphs, hip_graphs=True)
        torch.manual_seed(123)
        agent = A2CAgent('rg', params)
        agent.init_tensors()
        agent.obs = agent.env_reset()
        snap = {}
        orig_prepare = agent.prepare_dataset

        def prepare(batch_dict, _orig=orig_prepare, _agent=agent, _snap=snap):
            if _agent.is_rnn:            # recurrent state as the rollout left it
                _snap['rnn'] = [s.clone() for s in _agent.rnn_states]
            return _orig(batch_dict)
        agent.prepare_dataset = prepare
        for _ in range(3):
            agent.update_epoch()
            agent.train_epoch()
            if agent.is_rnn:             # ... must survive the update phase untouched (it is carried on)
                for s, want in zip(agent.rnn_states, snap['rnn']):
                    assert torch.equal(s, want)
        assert bool(agent._rollout_graphs) == graphs
        st = agent.experience_buffer.storage
        outs.append({k: st[k].clone() for k in ('obses', 'actions', 'mus', 'sigmas', 'values', 'neglogpacs',
                                                 'rewards', 'dones')})
        outs[-1]['params'] = agent.optimizer.flat_params.clone()
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k


@pytest.mark.parametrize('cfg', ['mlp', 'lstm'])
def test_hip_graph_replays_are_bit_identical_to_eager_training(cfg):
    """Same seeds, `hip_graphs` on vs off, six epochs: identical kernels in identical order, so the
    parameters, Adam moments, normaliser statistics and learning rate must match bit for bit."""
    from rl_games_amd import configs
    from rl_games_amd.agent import A2CAgent
    res = []
    for graphs in (True, False):
        if cfg == 'mlp':
            params = configs.tiny(num_actors=128, horizon=8, hip_graphs=graphs)
        else:
            params = configs.pendulum_lstm_4096(num_actors=64, hip_graphs=graphs)
        torch.manual_seed(7)
        agent = A2CAgent('g', params)
        agent.init_tensors()
        agent.obs = agent.env_reset()
        for _ in range(6):
            agent.update_epoch()
            agent.train_epoch()
        assert (agent._graph_epoch is not None or bool(agent._graphs)) == graphs
        res.append((agent.optimizer.flat_params.clone(), agent.optimizer.exp_avg.clone(),
                    agent.optimizer.exp_avg_sq.clone(), agent.model.running_mean_std.running_mean.clone(),
                    agent.model.value_mean_std.running_var.clone(), agent.optimizer.last_and_next_lr()))
    for a, b in zip(res[0][:5], res[1][:5]):
        assert torch.equal(a, b)
    assert res[0][5] == res[1][5]


def test_weights_set_between_epochs_reach_the_replayed_rollout_and_update_graphs():
    """set_weights() behind captured graphs: the rollout's step graphs and the update graphs contain no launch that packs
    the weights' derived forms (the lean 16-row kernels' fp32 fragments here, bf16 planes at the BASELINE sizes) - the
    agent re-packs them in front of the first replay behind a change that was not an optimiser step.  Same seeds, graphs on
    vs off, weights of epoch 2 put back behind epoch 4: bit-identical parameters, moments and statistics behind epoch 6."""
    from rl_games_amd import configs
    from rl_games_amd.agent import A2CAgent
    res = []
    for graphs in (True, False):
        params = configs.tiny(num_actors=128, horizon=8, hip_graphs=graphs)
        torch.manual_seed(11)
        agent = A2CAgent('w', params)
        agent.init_tensors()
        agent.obs = agent.env_reset()
        saved = None
        for ep in range(6):
            if ep == 2:
                saved = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in agent.get_weights()['model'].items()}
            if ep == 4:
                w = agent.get_weights()
                w['model'] = saved
                agent.set_weights(w)
            agent.update_epoch()
            agent.train_epoch()
        assert (agent._graph_epoch is not None or bool(agent._graphs)) == graphs
        if graphs:
            assert agent._lean_chain() is not None and len(agent._rollout_graphs) > 0
        res.append((agent.optimizer.flat_params.clone(), agent.optimizer.exp_avg.clone(), agent.optimizer.exp_avg_sq.clone(),
                    agent.model.running_mean_std.running_mean.clone()))
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)


def test_weights_written_from_outside_the_package_invalidate_the_derived_copies():
    """A write to the parameters that never calls weights_changed() - `model.load_state_dict` / `p.copy_()` by a user, a
    test, another restore path: the chains' planes / fragments are stamped with FlatArena.weights_token (the package's own
    counter + the autograd version counters), so the next forward packs again instead of running on the old weights.
    load_state_dict straight into the model behind epoch 3 against set_weights (which announces the change): bit-identical
    rollout outputs and parameters behind epoch 5, graphs on."""
    from rl_games_amd import configs
    from rl_games_amd.agent import A2CAgent
    res = []
    for how in ('set_weights', 'load_state_dict', 'param_copy'):
        params = configs.tiny(num_actors=128, horizon=8, hip_graphs=True)
        torch.manual_seed(11)
        agent = A2CAgent('w', params)
        agent.init_tensors()
        agent.obs = agent.env_reset()
        saved = None
        for ep in range(5):
            if ep == 1:
                saved = {k: v.clone() for k, v in agent._plain_model().state_dict().items()}
            if ep == 3:
                if how == 'set_weights':
                    w = agent.get_weights()
                    w['model'] = saved
                    agent.set_weights(w)
                elif how == 'load_state_dict':
                    agent._plain_model().load_state_dict(saved)
                else:
                    with torch.no_grad():
                        for k, p in agent._plain_model().named_parameters():
                            p.copy_(saved[k])
                    for k, b in agent._plain_model().named_buffers():
                        b.copy_(saved[k])
                probe = agent.get_action_values({'obs': agent.obs['obs']})
                res_probe = (probe['mus'].clone(), probe['values'].clone())
            agent.update_epoch()
            agent.train_epoch()
        assert agent._lean_chain() is not None and len(agent._rollout_graphs) > 0
        res.append(res_probe + (agent.optimizer.flat_params.clone(), agent.optimizer.exp_avg_sq.clone()))
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert torch.equal(a, b)


def test_a_failed_capture_leaves_nothing_marked_as_packed(monkeypatch):
    """_capture puts the host mirrors back behind a capture that failed part-way: the optimiser's step count AND
    weights_version, and what the chains believe their planes / fragments hold (the body's pack launches were recorded,
    never run).  The second (= last) minibatch of the first captured mini-epoch raises - behind a whole recorded step whose
    optimiser launch had marked the planes / fragments as those of the next weights; the epoch continues eagerly and ends
    bit for bit where an agent without graphs ends."""
    from rl_games_amd import configs
    from rl_games_amd.agent import A2CAgent
    res = []
    for fail in (True, False):
        params = configs.tiny(num_actors=128, horizon=8, hip_graphs=fail)
        torch.manual_seed(11)
        agent = A2CAgent('w', params)
        agent.init_tensors()
        agent.obs = agent.env_reset()
        if fail:
            real = agent._optimizer_kernels
            calls = [0]

            def flaky():
                if torch.cuda.is_current_stream_capturing():
                    calls[0] += 1
                    if calls[0] == 2:
                        raise RuntimeError('injected failure inside the capture')
                return real()
            monkeypatch.setattr(agent, '_optimizer_kernels', flaky)
        for ep in range(3):
            agent.update_epoch()
            agent.train_epoch()
        if fail:
            assert agent._graph_failed and calls[0] == 2
            assert agent.num_minibatches == 2
            assert agent.optimizer.step_count == int(agent.optimizer.step_counter.item())
        res.append((agent.optimizer.flat_params.clone(), agent.optimizer.exp_avg.clone(), agent.optimizer.exp_avg_sq.clone(),
                    agent.model.running_mean_std.running_mean.clone()))
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize('graphs', [True, False])
def test_folded_launches_match_the_separate_ones(graphs):
    """The launches that round 2 merged away - observation statistics folded in the forward's prologue
    (with an ODD number of minibatches: the state ends a mini-epoch in the second buffer set), loss
    partials and gradient-norm partials folded by the weight-gradient finalise - against the same agent
    with every one of them as its own launch.  Only summation orders differ."""
    from rl_games_amd import configs
    from rl_games_amd.agent import A2CAgent
    res = []
    for folded in (True, False):
        params = configs.tiny(num_actors=96, horizon=8, hip_graphs=graphs, fold_obs_stats=folded,
                              fold_loss_finalize=folded, norm_in_finalize=folded)
        params['config']['minibatch_size'] = 256                      # 768 rows -> 3 minibatches
        torch.manual_seed(11)
        agent = A2CAgent('f', params)
        agent.init_tensors()
        agent.obs = agent.env_reset()
        for _ in range(4):
            agent.update_epoch()
            agent.train_epoch()
        assert len(agent.dataset) == 3
        assert bool(agent._fin_norm_ok) == folded
        m = agent.model.running_mean_std
        res.append((m.running_mean.clone(), m.running_var.clone(), m.count.clone(),
                    agent.optimizer.flat_params.clone(), agent.optimizer.last_and_next_lr()))
    a, b = res
    assert a[2].item() == b[2].item() == 1 + 4 * 2 * 768              # every minibatch of every mini-epoch counted
    assert torch.allclose(a[0], b[0], rtol=1e-9, atol=1e-12) and torch.allclose(a[1], b[1], rtol=1e-9, atol=1e-12)
    assert a[4] == b[4]                                               # same learning-rate trajectory
    bad = ~torch.isclose(a[3], b[3], rtol=1e-3, atol=1e-6)
    assert bad.float().mean().item() <= 0.01
    assert (a[3] - b[3]).abs().max().item() <= 2.1 * 24 * 1e-2       # 24 Adam steps at lr <= max_lr


@pytest.mark.parametrize('graphs', [True, False])
def test_one_launch_step_in_the_agent_equals_the_two_launches(graphs):
    """The agent's optimiser step with forward + loss + backward as ONE launch (fused_step16, the default for
    minibatches below 16,384 rows) against the same agent with the two launches: bit-identical parameters, moments,
    normaliser state and learning rate after three epochs - and the fused launch really ran."""
    from rl_games_amd import configs
    from rl_games_amd.agent import A2CAgent
    res = []
    for fused in (True, False):
        params = configs.tiny(num_actors=96, horizon=8, hip_graphs=graphs, fused_step16=fused)
        params['config']['minibatch_size'] = 256                      # 768 rows -> 3 minibatches
        torch.manual_seed(11)
        agent = A2CAgent('s', params)
        agent.init_tensors()
        agent.obs = agent.env_reset()
        for _ in range(3):
            agent.update_epoch()
            agent.train_epoch()
        assert agent._engine.last_step_fused == fused
        m = agent.model.running_mean_std
        opt = agent.optimizer
        res.append((opt.flat_params.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), m.running_mean.clone(),
                    m.running_var.clone(), opt.last_and_next_lr()))
    a, b = res
    assert a[5] == b[5]
    for x, y in zip(a[:5], b[:5]):
        assert torch.equal(x, y)


def test_rccl_wrapper_of_the_c_abi_runs_inside_a_graph():
    """the RCCL wrapper taking an ncclComm_t (csrc/rccl_wrap.hip, rl_games_amd/rccl_allreduce.py) - a
    communicator of ONE rank (this box has one GPU, and RCCL refuses two ranks per device): unique id, comm create, an
    fp32 and an fp64 all-reduce eagerly and as nodes of a captured HIP graph (what the agent's mini-epoch graph needs of
    it), destroy.  The multi-rank behaviour is RCCL's own; the 2-rank agent tests cover the code around it with the
    hipIpc kernel and with torch.distributed."""
    from rl_games_amd import _lib
    from rl_games_amd.rccl_allreduce import RcclAllReduce
    if not _lib.load().rlg_rccl_available():
        pytest.skip('no librccl in this process')
    comm = RcclAllReduce(1 << 16, DEV, rank=0, world=1)
    x = torch.randn(50000, device=DEV)
    t = x.clone()
    comm.all_reduce_sum(t)
    d = torch.randn(777, device=DEV, dtype=torch.float64)
    td = d.clone()
    comm.all_reduce_sum(td)
    torch.cuda.synchronize()
    assert torch.equal(t, x) and torch.equal(td, d)
    buf = torch.zeros_like(x)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        comm.all_reduce_sum(buf)                       # (warm-up outside the capture)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        buf.mul_(2.0)
        comm.all_reduce_sum(buf)
        buf.add_(1.0)
    buf.copy_(x)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(buf, x * 2.0 + 1.0)
    assert comm.status()[1] == 0
    with pytest.raises(ValueError):
        comm.all_reduce_sum(buf, norm=(None, 0, 1.0, None))
    comm.close()


@pytest.mark.parametrize('graphs', [True, False])
def test_adam_written_planes_match_the_pack_launch(graphs):
    """Round 4: the Adam launch writes the split-bf16 chain's weight planes itself (adam_pack_kernel) - against the same
    agent with a pack launch in front of every forward: BASELINE configs[1] (32,768-row minibatches: forward and
    backward on planes, rollout forward at 4,096 rows on exact products), 2 epochs, eager and as mini-epoch graphs;
    then a restore (set_weights) in between - the planes must follow the weights.  Bit for bit."""
    from rl_games_amd import configs
    from rl_games_amd.agent import A2CAgent
    res = []
    for fused in (True, False):
        params = configs.ant_4096(hip_graphs=graphs, adam_writes_planes=fused)
        torch.manual_seed(3)
        agent = A2CAgent('p', params)
        agent.init_tensors()
        agent.obs = agent.env_reset()
        for _ in range(2):
            agent.update_epoch()
            agent.train_epoch()
        assert (agent._adam_pack_chain() is not None) == fused
        saved = copy.deepcopy(agent.get_weights())
        for _ in range(1):
            agent.update_epoch()
            agent.train_epoch()
        agent.set_weights(saved)                       # the weights change behind the chain's back
        agent.update_epoch()
        agent.train_epoch()
        opt = agent.optimizer
        res.append((opt.flat_params.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt.last_and_next_lr()))
    a, b = res
    assert a[3] == b[3]
    for x, y in zip(a[:3], b[:3]):
        assert torch.equal(x, y)


@pytest.mark.parametrize('kind,collective', [('mlp', 'ipc'), ('lstm', 'ipc'), ('discrete', 'ipc'), ('central_value', 'ipc'),
                                             ('mlp', 'fallback'), ('mlp', 'ipc-two-phase'), ('lstm', 'fallback')])
def test_two_rank_training_keeps_ranks_in_sync(kind, collective):
    """2 ranks on this box's single GPU (RLG_TEST_SINGLE_GPU=1: gloo collectives), different data per
    rank, 4 epochs of multi_gpu training for each agent kind: parameters, normaliser statistics
    (pooled merge: ONE collective for all normalisers) and learning rate end bit-identical on both ranks.
    collective: the in-graph hipIpc all-reduce kernel (default), its reduce-scatter + all-gather variant, or the
    forced torch.distributed fallback (`native_allreduce: False` - RCCL in production, its gloo stand-in here) through
    the same agent code."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    over = {'ipc': '{}', 'fallback': '{"native_allreduce": false}',
            'ipc-two-phase': '{"native_allreduce_two_phase": true}'}[collective]
    env = dict(os.environ, RLG_TEST_SINGLE_GPU='1', RLG_TWO_RANK_CONFIG=over)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.join(root, 'tools', 'two_rank_check.py'), kind]
    res = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    assert f'TWO_RANK_CHECK {kind} in_sync' in res.stdout
    want = {'ipc': 'ipc one_shot', 'fallback': 'rccl one_shot', 'ipc-two-phase': 'ipc two_phase'}[collective]
    assert f'TWO_RANK_ALLREDUCE {want}' in res.stdout, res.stdout[-600:]


@pytest.mark.parametrize('shape,collective', [('tiny', 'ipc'), ('humanoid', 'ipc'), ('tiny', 'fallback')])
def test_two_rank_steps_match_the_mean_of_two_oracle_ranks(shape, collective):
    """Rank-vs-oracle, not only rank-vs-rank: 2 ranks on this box's single GPU, different env shards, the optimiser steps
    of the first mini-epoch through the multi_gpu code path (gradient arena + KL slot through the in-graph all-reduce or
    the torch.distributed fallback, then clip + Adam + the lr rule on the averaged KL) against two oracle ranks stepped
    together by oracle.ppo_epoch_oracle.data_parallel_minibatch_step - the restatement of a2c_common.py:493-514 /
    :1557-1563 that tests/test_distributed_cpu.py pins to the reference method under gloo.  Per step: this rank's
    losses, the averaged clipped gradients (1e-5 of each tensor's scale), the parameters, the learning rate (exact).
    'humanoid': a rank of 8's 4,096-row minibatches of BASELINE config #4 (the lean 16-row kernels)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    over = {'ipc': '{}', 'fallback': '{"native_allreduce": false}'}[collective]
    os.environ['RLG_TWO_RANK_CONFIG'] = over
    try:
        res = _run_two_ranks([os.path.join(root, 'tools', 'two_rank_oracle_check.py'), shape], timeout=360)
    finally:
        os.environ.pop('RLG_TWO_RANK_CONFIG', None)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert f'TWO_RANK_ORACLE_CHECK {shape} ok' in res.stdout, res.stdout[-1500:]
    want = {'ipc': 'allreduce ipc', 'fallback': 'allreduce rccl'}[collective]
    assert want in res.stdout, res.stdout[-1500:]


def test_value_size_two_runs_on_the_torch_forms_and_matches_the_oracle_functions():
    """value_size > 1 (rewards / values / returns [B, V]; a2c_common.py:1622 sums the advantages over V) no longer raises:
    the rollout's kernels carry V columns, GAE runs per (env, value), and the update takes autograd through
    rl_games_amd/torch_fallback.py (pinned to the reference's own calc_losses on the CPU, tests/test_vs_reference_cpu.py).
    Here, on the device, V = 2: returns against the oracle scan bit for bit, the prepared dataset against the oracle's
    prepare_dataset, the first optimiser step's losses / KL / clipped gradients against the oracle's loss functions
    applied to a copy of the model, then two whole epochs."""
    from rl_games_amd import configs
    from rl_games_amd.agent import A2CAgent
    N, H, V = 64, 8, 2
    params = configs.tiny(num_actors=N, horizon=H, obs_dim=10, act_dim=4)
    params['config']['env_config']['value_size'] = V
    torch.manual_seed(3)
    agent = A2CAgent('v2', copy.deepcopy(params))
    assert agent.value_size == V and agent._engine is None
    agent.init_tensors()
    agent.obs = agent.env_reset()
    agent.set_eval()
    with torch.no_grad():
        batch = agent.play_steps()
    assert batch['values'].shape == (N * H, V) and batch['returns'].shape == (N * H, V)
    tb = agent.experience_buffer.tensor_dict
    assert tb['rewards'].shape == (H, N, V)
    last_values = agent.get_values(agent.obs)
    advs = O.gae_scan(tb['rewards'].cpu(), tb['values'].cpu(), tb['dones'].cpu().float(), last_values.cpu(),
                      agent.dones.cpu().float(), 0.99, 0.95)
    assert torch.equal(batch['returns'].cpu(), O.flatten_env_major(advs + tb['values'].cpu()))
    vms = agent.value_mean_std
    stats0 = {k: getattr(vms, k).detach().cpu().clone() for k in ('running_mean', 'running_var', 'count')}
    assert stats0['running_mean'].shape == (V,)
    cpu = {k: v.detach().cpu().clone() for k, v in batch.items() if isinstance(v, torch.Tensor)}
    agent.set_train()
    agent.prepare_dataset(batch)
    want = O.prepare_dataset(cpu['returns'], cpu['values'], stats0, normalize_value=True, normalize_advantage=True)
    vd = agent.dataset.values_dict
    for key in ('old_values', 'returns', 'advantages'):
        assert vd[key].shape == want[key].shape
        assert torch.allclose(vd[key].cpu(), want[key], rtol=1e-5, atol=2e-6), key
    assert torch.allclose(vms.running_mean.cpu(), want['value_stats']['running_mean'], rtol=1e-6, atol=1e-9)
    assert vms.count.item() == 1 + 2 * N * H
    # ---- the first optimiser step against the oracle's loss functions on a copy of the model
    twin = copy.deepcopy(agent._plain_model())
    twin.train()
    item = agent.dataset[0]
    old_mu, old_sigma = item['mu'].clone(), item['sigma'].clone()
    mu, logstd, values, _ = twin.forward_heads({'is_train': True, 'prev_actions': item['actions'], 'obs': item['obs']})
    sigma = torch.exp(logstd)
    ent = O.normal_entropy(mu, mu * 0 + sigma)
    nlp = torch.squeeze(O.neglogp(item['actions'], mu, mu * 0 + sigma, mu * 0 + logstd))
    cfg = params['config']
    loss, a, c, e, b = O.ppo_losses(item['old_logp_actions'], nlp, item['advantages'], item['old_values'], values,
                                    item['returns'], mu, ent, cfg['e_clip'], cfg['critic_coef'], cfg['entropy_coef'],
                                    cfg['bounds_loss_coef'], cfg['clip_value'])
    loss.backward()
    want_kl = O.policy_kl(mu.detach(), (mu * 0 + sigma).detach(), old_mu, old_sigma)
    norm = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in twin.parameters() if p.grad is not None)).item()
    coef = min(1.0, cfg['grad_norm'] / (norm + 1e-6))
    res = agent.train_actor_critic(item)
    for got, ref in ((res[0], a), (res[1], c), (res[2], e), (res[8], b)):
        assert np.isclose(got.item(), ref.item(), rtol=1e-5, atol=2e-6), (got.item(), ref.item())
    assert np.isclose(res[3].item(), want_kl.item(), rtol=1e-4, atol=2e-6)
    assert torch.allclose(item['mu'], mu.detach(), rtol=1e-6, atol=1e-7)              # update_mu_sigma (datasets.py:33-43)
    refp = dict(twin.named_parameters())
    for name, p in agent._plain_model().named_parameters():
        g_ref = refp[name].grad * coef
        scale = g_ref.abs().max().item()
        assert (p.grad - g_ref).abs().max().item() <= 1e-4 * scale + 1e-9, name
    # ---- whole epochs through the public entry point
    for _ in range(2):
        agent.update_epoch()
        out = agent.train_epoch()
    assert all(torch.isfinite(x).all() for x in out[4] + out[5])
    assert torch.isfinite(agent.optimizer.flat_params).all()
    assert agent.game_rewards.mean.shape[-1] == V


class _HostVecEnv:
    """A CPU vector env behind the IVecEnv seam (rl_games/common/ivecenv.py:1-36).  numpy=True hands out
    what a gym-style CPU env does (float64 observations / rewards, bool dones, numpy actions expected,
    a2c_common.py:663-673,:705-722); numpy=False hands the SAME numbers out as device tensors."""

    def __init__(self, numpy_io, num_envs, obs_dim, act_dim=0, discrete_actions=None, autoreset_mode='same_step'):
        self.inner = SyntheticTensorEnv(num_envs, obs_dim, act_dim, device='cpu', seed=77,
                                        discrete_actions=discrete_actions, autoreset_mode=autoreset_mode)
        self.numpy_io = numpy_io
        self.action_kinds = set()

    def _out(self, t, dtype=None):
        if self.numpy_io:
            a = t.numpy()
            return a.astype(dtype) if dtype is not None else a
        return t.to(DEV)

    def reset(self):
        return self._out(self.inner.reset(), np.float64)

    def step(self, actions):
        self.action_kinds.add(type(actions).__name__)
        if self.numpy_io:
            assert isinstance(actions, np.ndarray) and actions.shape[0] == self.inner.num_envs
        obs, rewards, dones, infos = self.inner.step(None)
        time_outs = infos['time_outs']
        return (self._out(obs, np.float64), self._out(rewards, np.float64), self._out(dones.bool()),
                {'time_outs': time_outs.numpy() if self.numpy_io else time_outs.to(DEV)})

    def get_env_info(self):
        return self.inner.get_env_info()

    def __getattr__(self, name):          # has_action_masks, set_train_info, get_env_state ...
        return getattr(self.inner, name)


@pytest.mark.parametrize('kind', ['continuous', 'discrete'])
def test_numpy_vec_env_plumbing_equals_the_tensor_env(kind):
    """BASELINE.json config #1 as stated - a CPU env speaking numpy (cast_obs / env_step branches,
    a2c_common.py:663-673,:705-722): two epochs on numpy observations / rewards / dones must leave the very
    same parameters as the same numbers handed over as device tensors."""
    from rl_games_amd import configs
    from rl_games_amd.agent import A2CAgent
    from rl_games_amd.discrete_agent import DiscreteA2CAgent

    def run(numpy_io):
        torch.manual_seed(5)
        if kind == 'discrete':
            params = configs.cartpole_discrete(num_actors=16, device=DEV)
            env = _HostVecEnv(numpy_io, 16, 4, discrete_actions=2, autoreset_mode='next_step')
            cls = DiscreteA2CAgent
        else:
            params = configs.tiny(num_actors=64, horizon=8, obs_dim=12, act_dim=3, device=DEV)
            env = _HostVecEnv(numpy_io, 64, 12, 3)
            cls = A2CAgent
        params['config']['vec_env'] = env
        params['config']['env_info'] = env.get_env_info()
        params['config']['seed'] = 11
        agent = cls('host_env', params)
        agent.init_tensors()
        agent.obs = agent.env_reset()
        assert agent.is_tensor_obses == (not numpy_io)
        assert agent.obs['obs'].dtype == torch.float32 and agent.obs['obs'].device.type == 'cuda'
        results = []
        for _ in range(2):
            agent.epoch_num += 1
            results.append(agent.train_epoch())
        assert env.action_kinds == ({'ndarray'} if numpy_io else {'Tensor'})
        return agent, results

    a_np, r_np = run(True)
    a_t, r_t = run(False)
    for x, y in zip(r_np, r_t):
        for u, v in zip(x[4] + x[5] + x[6] + x[7], y[4] + y[5] + y[6] + y[7]):     # losses, entropies, KLs
            assert torch.isfinite(u).all() and torch.equal(u, v)
    for (k, p), (_, q) in zip(a_np.model.state_dict().items(), a_t.model.state_dict().items()):
        assert torch.equal(p, q), k
    buf_np, buf_t = a_np.experience_buffer.tensor_dict, a_t.experience_buffer.tensor_dict
    for k in ('obses', 'rewards', 'dones', 'actions'):
        assert torch.equal(buf_np[k], buf_t[k]), k
