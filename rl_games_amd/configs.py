"""Parameter dictionaries for the BASELINE.json configurations (synthetic, device-resident
environments of the named shapes).  Hyper-parameters follow SURVEY 8(d) / BASELINE.md 3 (they
mirror rl_games/configs/mujoco/ant.yaml:36-57): adaptive lr, kl_threshold 0.008, e_clip 0.2,
clip_value, critic_coef 2, normalize input/value/advantage, truncate_grads, fp32."""
import copy


def _network(units, rnn=None):
    net = {
        'name': 'actor_critic', 'separate': False,
        'space': {'continuous': {
            'mu_activation': 'None', 'sigma_activation': 'None',
            'mu_init': {'name': 'default'},
            'sigma_init': {'name': 'const_initializer', 'val': 0},
            'fixed_sigma': True}},
        'mlp': {'units': list(units), 'activation': 'elu', 'initializer': {'name': 'default'}},
    }
    if rnn is not None:
        net['rnn'] = dict(rnn)
    return net


def _config(name, num_actors, horizon, minibatch, mini_epochs, env_config, **over):
    cfg = {
        'name': name, 'env_name': 'synthetic', 'env_config': dict(env_config),
        'normalize_input': True, 'normalize_value': True, 'normalize_advantage': True,
        'value_bootstrap': True, 'reward_shaper': {'scale_value': 1.0},
        'gamma': 0.99, 'tau': 0.95, 'learning_rate': 3e-4, 'lr_schedule': 'adaptive',
        'kl_threshold': 0.008, 'grad_norm': 1.0, 'entropy_coef': 0.0, 'truncate_grads': True,
        'e_clip': 0.2, 'clip_value': True, 'critic_coef': 2, 'bounds_loss_coef': 1e-4,
        'bound_loss_type': 'bound', 'num_actors': num_actors, 'horizon_length': horizon,
        'minibatch_size': minibatch, 'mini_epochs': mini_epochs, 'max_epochs': -1,
        'mixed_precision': False, 'print_stats': False, 'save_frequency': 0,
        'save_best_after': 10 ** 9, 'device': 'cuda:0', 'multi_gpu': False,
        'train_dir': '/tmp/rl_games_amd_runs',
    }
    cfg.update(over)
    return cfg


def humanoid_65536(num_actors=65536, minibatch_size=32768, **over):
    """BASELINE.json config #3/#4: Isaac-Humanoid-shaped obs 108 / act 21, 65,536 x 32."""
    return {'algo': {'name': 'a2c_continuous'}, 'model': {'name': 'continuous_a2c_logstd'},
            'network': _network([400, 200, 100]),
            'config': _config('humanoid_shaped', num_actors, 32, minibatch_size, 5,
                              {'obs_dim': 108, 'act_dim': 21}, **over)}


def ant_4096(num_actors=4096, **over):
    """BASELINE.json config #2: Ant-v5-shaped obs 60 / act 8, 4,096 x 16."""
    return {'algo': {'name': 'a2c_continuous'}, 'model': {'name': 'continuous_a2c_logstd'},
            'network': _network([256, 128, 64]),
            'config': _config('ant_shaped', num_actors, 16, min(32768, num_actors * 16), 4,
                              {'obs_dim': 60, 'act_dim': 8}, **over)}


def pendulum_lstm_4096(num_actors=4096, **over):
    """BASELINE.json config #5: LSTM policy, Pendulum-shaped obs 3 / act 1, 4,096 x seq_len 16."""
    return {'algo': {'name': 'a2c_continuous'}, 'model': {'name': 'continuous_a2c_logstd'},
            'network': _network([64, 64], rnn={'name': 'lstm', 'units': 64, 'layers': 1}),
            'config': _config('pendulum_lstm', num_actors, 16, min(16384, num_actors * 16), 4,
                              {'obs_dim': 3, 'act_dim': 1}, seq_length=16, **over)}


def tiny(num_actors=256, horizon=8, obs_dim=12, act_dim=3, **over):
    """Small config for smoke tests."""
    return {'algo': {'name': 'a2c_continuous'}, 'model': {'name': 'continuous_a2c_logstd'},
            'network': _network([32, 16]),
            'config': _config('tiny', num_actors, horizon, num_actors * horizon // 2, 2,
                              {'obs_dim': obs_dim, 'act_dim': act_dim}, **over)}


def cartpole_discrete(num_actors=16, **over):
    """BASELINE.json config #1 (rl_games/configs/ppo_cartpole.yaml): discrete PPO, CartPole-shaped
    obs 4 / 2 actions, separate actor/critic MLPs [32,32] relu, next_step autoreset (masked rows)."""
    net = {'name': 'actor_critic', 'separate': True, 'space': {'discrete': None},
           'mlp': {'units': [32, 32], 'activation': 'relu', 'initializer': {'name': 'default'}}}
    cfg = _config('cartpole_shaped', num_actors, 32, 64, 4,
                  {'obs_dim': 4, 'discrete_actions': 2, 'autoreset_mode': 'next_step'},
                  normalize_input=False, normalize_value=False, learning_rate=2e-4, lr_schedule=None,
                  entropy_coef=0.01, critic_coef=1, tau=0.9, reward_shaper={'scale_value': 0.1})
    cfg.pop('bounds_loss_coef')
    cfg.pop('bound_loss_type')
    cfg.update(over)
    return {'algo': {'name': 'a2c_discrete'}, 'model': {'name': 'discrete_a2c'}, 'network': net, 'config': cfg}


def clone(params):
    return copy.deepcopy(params)
