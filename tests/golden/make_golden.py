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
sys.path.insert(0, HERE)
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


def _clone(x):
    if isinstance(x, torch.Tensor):
        return x.detach().clone()
    if isinstance(x, dict):
        return {k: _clone(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_clone(v) for v in x]
    return x


def make_epoch(variants=None, filename='epoch.pt'):
    """One full train_epoch of the REAL reference A2CAgent (a2c_continuous.py / a2c_common.py) on
    CPU, driven by the synthetic tensor env, recorded tensor by tensor."""
    import copy
    import ref_import
    ref_import.enable()
    from rl_games.torch_runner import Runner
    from rl_games_amd import configs
    from rl_games_amd.synthetic_env import SyntheticTensorEnv

    variants = variants or {
        'default': dict(),
        'smooth_reg_ema': dict(use_smooth_clamp=True, bound_loss_type='regularisation', bounds_loss_coef=0.01,
                               normalize_rms_advantage=True, entropy_coef=0.01, critic_coef=1.0),
        # BASELINE.json config #5 in miniature: LSTM policy, play_steps_rnn, seq_length chunks
        'lstm': dict(seq_length=4, _rnn={'name': 'lstm', 'units': 16, 'layers': 1}),
    }
    out = {}
    for name, over in variants.items():
        N, H, O_, A = 64, 8, 12, 3
        over = dict(over)
        rnn = over.pop('_rnn', None)
        separate = over.pop('_separate', False)
        space = over.pop('_space', None)
        net_over = over.pop('_network', None)
        params = configs.tiny(num_actors=N, horizon=H, obs_dim=O_, act_dim=A, device='cpu',
                              train_dir='/tmp/rlg_golden_runs', games_to_track=100, **over)
        if rnn is not None:
            params['network']['rnn'] = rnn
        if separate:
            params['network']['separate'] = True
        if space is not None:
            params['network']['space']['continuous'].update(space)
        if net_over is not None:
            for k, v in net_over.items():
                if isinstance(v, dict) and isinstance(params['network'].get(k), dict):
                    params['network'][k].update(v)
                else:
                    params['network'][k] = v
        params['seed'] = 7
        env = SyntheticTensorEnv(N, O_, A, device='cpu', seed=1234)
        params['config']['env_info'] = env.get_env_info()
        stored_params = copy.deepcopy({k: v for k, v in params.items()})
        stored_params['config'].pop('env_info')
        runner = Runner()
        runner.load({'params': copy.deepcopy(params)})
        runner.params['config']['vec_env'] = env
        runner.params['config']['env_info'] = env.get_env_info()
        agent = runner.algo_factory.create(runner.algo_name, base_name='golden', params=runner.params)
        torch.manual_seed(11)
        agent.init_tensors()
        agent.obs = agent.env_reset()
        cap = {'init_state': _clone(agent.model.state_dict()), 'lrs': []}
        play_name = 'play_steps_rnn' if agent.is_rnn else 'play_steps'
        orig_play = getattr(agent, play_name)

        def play():
            b = orig_play()
            cap['batch'] = _clone({k: v for k, v in b.items() if isinstance(v, torch.Tensor)})
            if 'rnn_states' in b:
                cap['batch']['rnn_states'] = _clone(b['rnn_states'])
            cap['buffers'] = _clone({k: agent.experience_buffer.tensor_dict[k]
                                     for k in ('rewards', 'values', 'dones')})
            cap['last_dones'] = agent.dones.clone()
            cap['last_values'] = agent.get_values(agent.obs).clone()
            cap['cur'] = _clone([agent.current_rewards, agent.current_shaped_rewards, agent.current_lengths])
            cap['meters'] = {'mean': _clone([agent.game_rewards.mean, agent.game_shaped_rewards.mean,
                                             agent.game_lengths.mean]),
                             'n': [agent.game_rewards.current_size, agent.game_shaped_rewards.current_size,
                                   agent.game_lengths.current_size]}
            cap['state_after_rollout'] = _clone(agent.model.state_dict())
            return b
        setattr(agent, play_name, play)
        orig_update_lr = agent.update_lr

        def update_lr(lr):
            cap['lrs'].append(float(lr))
            return orig_update_lr(lr)
        agent.update_lr = update_lr
        agent.epoch_num = 1
        res = agent.train_epoch()
        (_, _, _, _, a_losses, c_losses, b_losses, entropies, kls, last_lr, lr_mul) = res
        cap['a_losses'] = torch.stack([x.detach() for x in a_losses])
        cap['c_losses'] = torch.stack([x.detach() for x in c_losses])
        cap['b_losses'] = torch.stack([x.detach() for x in b_losses]) if b_losses else None
        cap['entropies'] = torch.stack([x.detach() for x in entropies])
        cap['mini_epoch_kls'] = torch.stack([x.detach() for x in kls])
        cap['last_lr'] = float(last_lr)
        vd = agent.dataset.values_dict
        cap['dataset'] = _clone({k: vd[k] for k in ('old_values', 'returns', 'advantages', 'mu', 'sigma')})
        cap['final_state'] = _clone(agent.model.state_dict())
        opt = agent.optimizer.state_dict()
        cap['opt_state'] = _clone({i: {k: v for k, v in s.items()} for i, s in opt['state'].items()})
        if over.get('normalize_rms_advantage'):
            cap['adv_ema'] = _clone(agent.advantage_mean_std.state_dict())
        cap['params'] = stored_params
        cap['env'] = {'num_envs': N, 'obs_dim': O_, 'act_dim': A, 'seed': 1234}
        out[name] = cap
        print('epoch', name, 'minibatches', len(a_losses), 'lrs', cap['lrs'][:4], 'kl', cap['mini_epoch_kls'].tolist())
    torch.save(out, os.path.join(HERE, filename))
    print(filename, 'written', os.path.getsize(os.path.join(HERE, filename)) // 1024, 'KiB')


def make_epoch_extra():
    """Round 6: epochs of the real reference agent for the API corners that left the NotImplementedError list -
    a state-dependent sigma head (fixed_sigma False, network_builder.py:341-344) and the plain A2C loss (ppo False,
    common_losses.py:80) - and a D2RL trunk with layer normalisation (d2rl.py, network_builder.py:105-145).  A file of its own:
    epoch.pt keeps its bytes."""
    make_epoch({
        'state_sigma': dict(_space={'fixed_sigma': False, 'sigma_init': {'name': 'const_initializer', 'val': -0.5}}),
        'ppo_false': dict(ppo=False),
        'd2rl_layer_norm': dict(_network={'mlp': {'d2rl': True}, 'normalization': 'layer_norm'}),
    }, 'epoch_extra.pt')


def make_epoch_sigma_forms():
    """Round 6: the sigma head's parametrisations (network_builder.py:311-322, models.py:272-301) - softplus with a floor on
    a state-dependent head, exp with log-sigma bounds (active: the initial 0.3 sits above the upper bound) and a floor on
    the parameter vector, and the linear ('scalar') form."""
    const = lambda v: {'name': 'const_initializer', 'val': v}
    make_epoch({
        'softplus_state_sigma': dict(_space={'fixed_sigma': False, 'sigma_parametrization': 'softplus', 'min_sigma': 0.05,
                                             'sigma_init': const(0.5)}),
        'bounded_exp_floor': dict(_space={'logstd_bounds': [-1.0, 0.1], 'min_sigma': 0.1, 'sigma_init': const(0.3)}),
        'linear_sigma': dict(_space={'sigma_parametrization': 'scalar', 'sigma_init': const(0.8)}, entropy_coef=0.01),
    }, 'epoch_sigma_forms.pt')


def make_epoch_batch_norm():
    """Round 6: BatchNorm1d behind every trunk layer (network_builder.py:128-129) - batch statistics in the update, the
    running ones in the rollout - and inside a D2RL trunk (d2rl.py:19-20)."""
    make_epoch({
        'batch_norm': dict(_network={'normalization': 'batch_norm'}),
        'd2rl_batch_norm': dict(_network={'mlp': {'d2rl': True}, 'normalization': 'batch_norm'}),
    }, 'epoch_batch_norm.pt')


def make_epoch_separate_rnn():
    """Round 6: separate actor / critic trunks, each with its own RNN (network_builder.py:272-277, :372-421) - four LSTM
    state tensors per environment, two with a GRU."""
    make_epoch({
        'separate_lstm': dict(seq_length=4, _separate=True, _rnn={'name': 'lstm', 'units': 16, 'layers': 1}),
        'separate_gru_layer_norm': dict(seq_length=4, _separate=True,
                                        _rnn={'name': 'gru', 'units': 12, 'layers': 1, 'layer_norm': True}),
    }, 'epoch_separate_rnn.pt')


def make_discrete(variants=None, filename='discrete.pt'):
    """One train_epoch of the REAL reference DiscreteA2CAgent (a2c_discrete.py) on CPU: CartPole-like
    shapes (BASELINE.json config #1 in miniature), separate actor/critic MLPs, next_step autoreset
    (masked filler rows), adaptive lr stepped once per mini-epoch."""
    import copy
    import ref_import
    ref_import.enable()
    from rl_games.torch_runner import Runner
    from rl_games_amd import configs
    from rl_games_amd.synthetic_env import SyntheticTensorEnv

    variants = variants or {
        'masked_adaptive': dict(normalize_input=True, normalize_value=True, lr_schedule='adaptive',
                                learning_rate=3e-4, kl_threshold=0.002, p_done=0.15),
        'plain': dict(_autoreset='same_step', entropy_coef=0.02, p_done=0.05),
        # ModelA2CMultiDiscrete + CategoricalMasked: two heads (3 and 4 actions), random action masks
        'multi_discrete_masked': dict(_autoreset='same_step', _heads=[3, 4], _masks=True, entropy_coef=0.02,
                                      p_done=0.05, normalize_input=True),
    }
    out = {}
    for name, over in variants.items():
        over = dict(over)
        N, H, O_, n_act = 16, 8, 4, 3
        p_done = over.pop('p_done')
        mode = over.pop('_autoreset', 'next_step')
        heads = over.pop('_heads', None)
        masks = over.pop('_masks', False)
        rnn = over.pop('_rnn', None)
        separate = over.pop('_separate', False)
        params = configs.cartpole_discrete(num_actors=N, horizon_length=H, minibatch_size=32, mini_epochs=2,
                                           device='cpu', train_dir='/tmp/rlg_golden_runs', **over)
        env_kw = dict(obs_dim=O_, discrete_actions=heads or n_act, autoreset_mode=mode, p_done=p_done, seed=99,
                      action_masks=masks)
        if heads:
            params['network']['space'] = {'multi_discrete': None}
            params['model']['name'] = 'multi_discrete_a2c'
        if masks:
            params['config']['use_action_masks'] = True
        if rnn:
            params['network']['rnn'] = dict(rnn)
            params['network']['separate'] = separate
        params['config']['env_config'] = dict(env_kw)
        params['seed'] = 5
        env = SyntheticTensorEnv(N, device='cpu', **env_kw)
        stored_params = copy.deepcopy(params)
        runner = Runner()
        runner.load({'params': copy.deepcopy(params)})
        runner.params['config']['vec_env'] = env
        runner.params['config']['env_info'] = env.get_env_info()
        agent = runner.algo_factory.create(runner.algo_name, base_name='golden', params=runner.params)
        assert type(agent).__name__ == 'DiscreteA2CAgent'
        torch.manual_seed(13)
        agent.init_tensors()
        agent.obs = agent.env_reset()
        cap = {'init_state': _clone(agent.model.state_dict()), 'lrs': []}
        play_name = 'play_steps_rnn' if rnn else 'play_steps'
        orig_play = getattr(agent, play_name)

        def play():
            b = orig_play()
            cap['batch'] = _clone({k: v for k, v in b.items() if isinstance(v, torch.Tensor)})
            if rnn:
                cap['batch']['rnn_states'] = _clone(b['rnn_states'])
            cap['buffers'] = _clone({k: agent.experience_buffer.tensor_dict[k]
                                     for k in ('rewards', 'values', 'dones')})
            cap['state_after_rollout'] = _clone(agent.model.state_dict())
            return b
        setattr(agent, play_name, play)
        orig_update_lr = agent.update_lr

        def update_lr(lr):
            cap['lrs'].append(float(lr))
            return orig_update_lr(lr)
        agent.update_lr = update_lr
        mb_results = []
        orig_calc = agent.calc_gradients

        def calc(input_dict):
            orig_calc(input_dict)
            mb_results.append([x.detach().clone() for x in agent.train_result[:4]])
        agent.calc_gradients = calc
        agent.epoch_num = 1
        res = agent.train_epoch()
        (_, _, _, _, a_losses, c_losses, entropies, kls, last_lr, lr_mul) = res
        cap['a_losses'] = torch.stack([x.detach() for x in a_losses])
        cap['c_losses'] = torch.stack([x.detach() for x in c_losses])
        cap['entropies'] = torch.stack([x.detach() for x in entropies])
        cap['mb_kls'] = torch.stack([r[3] for r in mb_results])
        cap['mini_epoch_kls'] = torch.stack([x.detach() for x in kls])
        cap['last_lr'] = float(last_lr)
        vd = agent.dataset.values_dict
        cap['dataset'] = _clone({k: vd[k] for k in ('old_values', 'returns', 'advantages', 'actions',
                                                      'old_logp_actions')})
        if masks:
            cap['dataset']['action_masks'] = _clone(vd['action_masks'])
        cap['final_state'] = _clone(agent.model.state_dict())
        cap['params'] = stored_params
        cap['num_envs'] = N
        out[name] = cap
        print('discrete', name, 'minibatches', len(a_losses), 'masked rows',
              None if 'rnn_masks' not in cap['batch'] else int((cap['batch']['rnn_masks'] == 0).sum()),
              'lrs', cap['lrs'], 'kl', cap['mini_epoch_kls'].tolist())
    torch.save(out, os.path.join(HERE, filename))
    print(filename, 'written', os.path.getsize(os.path.join(HERE, filename)) // 1024, 'KiB')


def make_discrete_rnn():
    """Recurrent categorical policies (round 6): the reference agent's play_steps_rnn + sequence minibatches
    (a2c_common.py:1071-1202, a2c_discrete.py:138-144) with an LSTM behind a shared trunk - plain, and multi-discrete
    with action masks and filler rows (next_step autoreset)."""
    lstm = dict(name='lstm', units=16, layers=1, before_mlp=False)
    make_discrete({
        'lstm': dict(_autoreset='same_step', _rnn=lstm, seq_length=4, entropy_coef=0.02, p_done=0.1,
                     normalize_input=True, normalize_value=True),
        'lstm_multi_masked': dict(_rnn=lstm, _heads=[3, 4], _masks=True, seq_length=4, entropy_coef=0.01, p_done=0.15,
                                  normalize_input=True, lr_schedule='adaptive', learning_rate=3e-4, kl_threshold=0.002),
        'gru_before_mlp': dict(_autoreset='same_step', _rnn=dict(name='gru', units=12, layers=1, before_mlp=True),
                               seq_length=4, p_done=0.1),
        # separate actor / critic trunks, each with its own LSTM + layer norm: four state tensors (network_builder.py:272-277)
        'lstm_separate': dict(_rnn=dict(name='lstm', units=8, layers=1, layer_norm=True), _separate=True, seq_length=4,
                              p_done=0.15, normalize_input=True, normalize_value=True),
    }, 'discrete_rnn.pt')


def make_checkpoint():
    """A checkpoint written by the REAL reference agent (`A2CBase.save` -> torch_ext.save_checkpoint,
    a2c_common.py:921-925, torch_ext.py:89-91) after one epoch, for the interop test."""
    import copy
    import shutil
    import ref_import
    ref_import.enable()
    from rl_games.torch_runner import Runner
    from rl_games_amd import configs
    from rl_games_amd.synthetic_env import SyntheticTensorEnv
    N, H, O_, A = 32, 8, 6, 2
    params = configs.tiny(num_actors=N, horizon=H, obs_dim=O_, act_dim=A, device='cpu',
                          train_dir='/tmp/rlg_golden_runs')
    params['seed'] = 3
    env = SyntheticTensorEnv(N, O_, A, device='cpu', seed=77)
    stored = copy.deepcopy(params)
    runner = Runner()
    runner.load({'params': copy.deepcopy(params)})
    runner.params['config']['vec_env'] = env
    runner.params['config']['env_info'] = env.get_env_info()
    agent = runner.algo_factory.create(runner.algo_name, base_name='golden', params=runner.params)
    agent.init_tensors()
    agent.obs = agent.env_reset()
    agent.epoch_num = 4
    agent.frame = 4 * N * H
    agent.train_epoch()
    agent.last_mean_rewards = 1.25
    agent.save('/tmp/rlg_golden_ref_ckpt')
    shutil.copy('/tmp/rlg_golden_ref_ckpt.pth', os.path.join(HERE, 'ref_checkpoint.pth'))
    torch.save({'params': stored, 'env': {'num_envs': N, 'obs_dim': O_, 'act_dim': A, 'seed': 77}},
               os.path.join(HERE, 'ref_checkpoint_meta.pt'))
    ck = torch.load(os.path.join(HERE, 'ref_checkpoint.pth'), weights_only=False)
    print('ref_checkpoint.pth keys', sorted(ck), 'lr', ck['optimizer']['param_groups'][0]['lr'],
          os.path.getsize(os.path.join(HERE, 'ref_checkpoint.pth')) // 1024, 'KiB')


def make_central_value(variants=None, filename='central_value.pt'):
    """One train_epoch of the REAL reference A2CAgent with a central (asymmetric) value function
    (central_value.py CentralValueTrain): privileged states, own optimiser/minibatches."""
    import copy
    import ref_import
    ref_import.enable()
    from rl_games.torch_runner import Runner
    from rl_games_amd import configs
    from rl_games_amd.synthetic_env import SyntheticTensorEnv
    out = {}
    variants = variants or {'experimental_cv': dict(), 'no_actor_value_loss': dict(use_experimental_cv=False)}
    for name, over in variants.items():
        N, H, O_, A, S_ = 64, 8, 12, 3, 20
        over = dict(over)
        rnn, cv_rnn, agents = over.pop('_rnn', None), over.pop('_cv_rnn', None), over.pop('_agents', 1)
        if agents > 1:
            N = 32
        params = configs.tiny(num_actors=N, horizon=H, obs_dim=O_, act_dim=A, device='cpu',
                              train_dir='/tmp/rlg_golden_runs', **over)
        params['config']['central_value_config'] = {
            'minibatch_size': N * H // 4, 'mini_epochs': 2, 'learning_rate': 5e-4, 'clip_value': True,
            'normalize_input': True, 'truncate_grads': True, 'grad_norm': 1.0,
            'network': {'name': 'actor_critic', 'central_value': True,
                        'mlp': {'units': [24, 16], 'activation': 'elu', 'initializer': {'name': 'default'}}},
        }
        if rnn is not None:
            params['network']['rnn'] = dict(rnn)
        if cv_rnn is not None:
            params['config']['central_value_config']['network']['rnn'] = dict(cv_rnn)
        params['config']['env_config']['state_dim'] = S_
        if agents > 1:
            params['config']['env_config']['agents'] = agents
        params['seed'] = 9
        env = SyntheticTensorEnv(N, O_, A, device='cpu', seed=4321, state_dim=S_, agents=agents)
        stored = copy.deepcopy(params)
        runner = Runner()
        runner.load({'params': copy.deepcopy(params)})
        runner.params['config']['vec_env'] = env
        runner.params['config']['env_info'] = env.get_env_info()
        agent = runner.algo_factory.create(runner.algo_name, base_name='golden', params=runner.params)
        assert agent.has_central_value
        torch.manual_seed(17)
        agent.init_tensors()
        agent.obs = agent.env_reset()
        cap = {'lrs': []}
        play_name = 'play_steps_rnn' if agent.is_rnn else 'play_steps'
        orig_play = getattr(agent, play_name)

        def play():
            b = orig_play()
            cap['batch'] = _clone({k: v for k, v in b.items() if isinstance(v, torch.Tensor)})
            if 'rnn_states' in b:
                cap['batch']['rnn_states'] = _clone(b['rnn_states'])
            if agent.central_value_net.is_rnn:                   # what update_dataset reads (central_value.py:163-170)
                cap['cv_mb_rnn_states'] = _clone(agent.central_value_net.mb_rnn_states)
                cap['cv_rnn_states'] = _clone(agent.central_value_net.rnn_states)
            cap['state_after_rollout'] = _clone(agent.model.state_dict())
            cap['cv_state_after_rollout'] = _clone(agent.central_value_net.state_dict())
            return b
        setattr(agent, play_name, play)
        orig_update_lr = agent.update_lr

        def update_lr(lr):
            cap['lrs'].append(float(lr))
            return orig_update_lr(lr)
        agent.update_lr = update_lr
        cv_losses = []
        orig_cv = agent.central_value_net.calc_gradients

        def cv_calc(batch):
            loss = orig_cv(batch)
            cv_losses.append(loss.detach().clone())
            return loss
        agent.central_value_net.calc_gradients = cv_calc
        agent.epoch_num = 1
        res = agent.train_epoch()
        (_, _, _, _, a_losses, c_losses, b_losses, entropies, kls, last_lr, lr_mul) = res
        cap['a_losses'] = torch.stack([x.detach() for x in a_losses])
        cap['c_losses'] = torch.stack([x.detach().reshape(()) for x in c_losses])
        cap['entropies'] = torch.stack([x.detach() for x in entropies])
        cap['mini_epoch_kls'] = torch.stack([x.detach() for x in kls])
        cap['cv_losses'] = torch.stack(cv_losses)
        vd = agent.dataset.values_dict
        cap['dataset'] = _clone({k: vd[k] for k in ('old_values', 'returns', 'advantages')})
        cap['final_state'] = _clone(agent.model.state_dict())
        cap['cv_final_state'] = _clone(agent.central_value_net.state_dict())
        cap['params'] = stored
        cap['env'] = {'num_envs': N, 'obs_dim': O_, 'act_dim': A, 'seed': 4321, 'state_dim': S_, 'agents': agents}
        cvd = agent.central_value_net.dataset.values_dict
        cap['cv_dataset'] = _clone({k: cvd[k] for k in ('old_values', 'returns', 'dones')})
        out[name] = cap
        print('central_value', name, 'cv losses', cap['cv_losses'].tolist()[:3], 'c_losses', cap['c_losses'].tolist()[:2],
              'keys', sorted(cap['batch'])[:12])
    torch.save(out, os.path.join(HERE, filename))
    print(filename, 'written', os.path.getsize(os.path.join(HERE, filename)) // 1024, 'KiB')


def make_central_value_rnn():
    """Round 6: a recurrent actor with a central value function (the rollout of play_steps_rnn stores the privileged
    states, a2c_common.py:1119-1120) - the critic a plain MLP, and the critic with an RNN of its own
    (central_value.py:96-107,163-205: its states advance with the rollout and train on sequence minibatches)."""
    lstm = {'name': 'lstm', 'units': 16, 'layers': 1}
    make_central_value({
        'rnn_actor_mlp_critic': dict(seq_length=4, _rnn=lstm),
        'rnn_actor_rnn_critic': dict(seq_length=4, _rnn=lstm, _cv_rnn={'name': 'lstm', 'units': 12, 'layers': 1}),
        'rnn_actor_gru_critic_layer_norm': dict(seq_length=4, _rnn=lstm, use_experimental_cv=False,
                                                _cv_rnn={'name': 'gru', 'units': 12, 'layers': 1, 'layer_norm': True}),
    }, 'central_value_rnn.pt')


def make_central_value_multi_agent():
    """Round 6: multi-agent envs under a central value function (central_value.py:153-158,223-234) - 3 agents per env
    with an MLP critic, and 2 agents with a recurrent actor and a recurrent critic."""
    lstm = {'name': 'lstm', 'units': 16, 'layers': 1}
    make_central_value({
        'three_agents': dict(_agents=3),
        'two_agents_recurrent': dict(_agents=2, seq_length=4, _rnn=lstm, _cv_rnn={'name': 'lstm', 'units': 12, 'layers': 1}),
    }, 'central_value_multi_agent.pt')


def make_lstm_full():
    """BASELINE.json config #5 AT ITS OWN SIZE - 4,096 envs x seq_len 16, obs 3, act 1, MLP [64,64] + LSTM 64,
    minibatch 16,384, 4 mini-epochs = 16 optimiser steps - one train_epoch of the REAL reference agent
    (play_steps_rnn, a2c_common.py:1071-1202; recurrent.py:26-83) on the synthetic env.  Stored: the rollout batch
    with the rnn states, the model state it was played with, and the reference's per-minibatch results.
    gzip-compressed (the initial rnn states are zeros): tests/golden/lstm_full.pt.gz."""
    import copy
    import gzip
    import io
    import ref_import
    ref_import.enable()
    from rl_games.torch_runner import Runner
    from rl_games_amd import configs
    from rl_games_amd.synthetic_env import SyntheticTensorEnv
    params = configs.pendulum_lstm_4096(device='cpu', train_dir='/tmp/rlg_golden_runs', games_to_track=100)
    N, O_, A = 4096, 3, 1
    params['seed'] = 7
    env = SyntheticTensorEnv(N, O_, A, device='cpu', seed=1234)
    params['config']['env_info'] = env.get_env_info()
    stored_params = copy.deepcopy({k: v for k, v in params.items()})
    stored_params['config'].pop('env_info')
    runner = Runner()
    runner.load({'params': copy.deepcopy(params)})
    runner.params['config']['vec_env'] = env
    runner.params['config']['env_info'] = env.get_env_info()
    agent = runner.algo_factory.create(runner.algo_name, base_name='golden', params=runner.params)
    torch.manual_seed(11)
    agent.init_tensors()
    agent.obs = agent.env_reset()
    cap = {'lrs': []}
    orig_play = agent.play_steps_rnn

    def play():
        b = orig_play()
        cap['batch'] = _clone({k: v for k, v in b.items() if isinstance(v, torch.Tensor)})
        cap['batch']['rnn_states'] = _clone(b['rnn_states'])
        cap['state_after_rollout'] = _clone(agent.model.state_dict())
        return b
    agent.play_steps_rnn = play
    orig_update_lr = agent.update_lr

    def update_lr(lr):
        cap['lrs'].append(float(lr))
        return orig_update_lr(lr)
    agent.update_lr = update_lr
    agent.epoch_num = 1
    res = agent.train_epoch()
    (_, _, _, _, a_losses, c_losses, b_losses, entropies, kls, last_lr, lr_mul) = res
    cap['a_losses'] = torch.stack([x.detach() for x in a_losses])
    cap['c_losses'] = torch.stack([x.detach() for x in c_losses])
    cap['b_losses'] = torch.stack([x.detach() for x in b_losses])
    cap['entropies'] = torch.stack([x.detach() for x in entropies])
    cap['mini_epoch_kls'] = torch.stack([x.detach() for x in kls])
    cap['last_lr'] = float(last_lr)
    cap['final_state'] = _clone(agent.model.state_dict())
    cap['params'] = stored_params
    cap['env'] = {'num_envs': N, 'obs_dim': O_, 'act_dim': A, 'seed': 1234}
    buf = io.BytesIO()
    torch.save(cap, buf)
    path = os.path.join(HERE, 'lstm_full.pt.gz')
    with gzip.open(path, 'wb', compresslevel=9) as f:
        f.write(buf.getvalue())
    print('lstm_full: minibatches', len(a_losses), 'lrs', cap['lrs'][:4], 'kl', cap['mini_epoch_kls'].tolist())
    print('lstm_full.pt.gz written', os.path.getsize(path) // 1024, 'KiB (raw', len(buf.getvalue()) // 1024, 'KiB)')


SECTIONS = {'gae': make_gae, 'epoch': make_epoch, 'discrete': make_discrete, 'discrete_rnn': make_discrete_rnn, 'epoch_separate_rnn': make_epoch_separate_rnn, 'epoch_sigma_forms': make_epoch_sigma_forms, 'epoch_batch_norm': make_epoch_batch_norm, 'central_value_rnn': make_central_value_rnn, 'central_value_multi_agent': make_central_value_multi_agent, 'checkpoint': make_checkpoint,
            'central_value': make_central_value, 'lstm_full': make_lstm_full, 'epoch_extra': make_epoch_extra}

if __name__ == '__main__':
    only = sys.argv[1:] or list(SECTIONS)
    for name in only:
        SECTIONS[name]()
