"""TEST INFRASTRUCTURE ONLY - CPU restatement of one full PPO epoch of the reference.

`OracleAgent` strings the leaf functions of ppo_oracle.py together in the order of
ContinuousA2CBase.train_epoch (rl_games/common/a2c_common.py:1517-1584): play_steps (:985-1069),
prepare_dataset (:1586-1660), then mini_epochs x minibatches of calc_gradients
(rl_games/algos_torch/a2c_continuous.py:136-234), trancate_gradients_and_step
(a2c_common.py:493-514) and the per-minibatch adaptive learning rate (:1557-1563).  The model
is the reference's `continuous_a2c_logstd` over an `actor_critic` MLP with a fixed-sigma
parameter (models.py:329-359, network_builder.py:447-512), written with plain torch.nn on
the CPU; eager op for eager op it issues the same PyTorch calls as the reference, so
  * tests/test_oracle_epoch.py can pin it against golden vectors recorded from the real
    reference agent (tests/golden/make_golden.py, section "epoch"), and
  * bench.py can time it on the GPU box's host cores as the `cpu_baseline` ("port").
Used by tests/, __graft_entry__.smoke() and bench.py only.
"""
import time

import numpy as np
import torch
from torch import nn

from . import ppo_oracle as O

_ACT = {'elu': nn.ELU, 'relu': nn.ReLU, 'tanh': nn.Tanh, 'None': nn.Identity}


class OracleNet(nn.Module):
    """Parameter names equal the reference's (`a2c_network.*`)."""

    def __init__(self, obs_dim, act_dim, units, activation='elu'):
        super().__init__()
        layers, last = [], obs_dim
        for u in units:
            layers += [nn.Linear(last, u), _ACT[activation]()]
            last = u
        self.actor_mlp = nn.Sequential(*layers)
        self.value = nn.Linear(last, 1)
        self.mu = nn.Linear(last, act_dim)
        self.sigma = nn.Parameter(torch.zeros(act_dim))
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.zeros_(m.bias)

    def forward(self, obs):
        out = self.actor_mlp(obs)
        mu = self.mu(out)
        return mu, mu * 0 + self.sigma, self.value(out)          # network_builder.py:506-512


class OracleModel(nn.Module):
    def __init__(self, obs_dim, act_dim, units, normalize_input=True, normalize_value=True):
        super().__init__()
        self.a2c_network = OracleNet(obs_dim, act_dim, units)
        self.normalize_input, self.normalize_value = normalize_input, normalize_value
        self.obs_stats = O.new_running_stats(obs_dim)
        self.value_stats = O.new_running_stats(1)
        self.obs_stats_training = False

    # state_dict in the reference's layout (buffers under running_mean_std.* / value_mean_std.*)
    def full_state_dict(self):
        sd = {'a2c_network.' + k: v.detach().clone() for k, v in self.a2c_network.state_dict().items()}
        if self.normalize_value:
            for k, v in self.value_stats.items():
                sd['value_mean_std.' + k] = v.clone()
        if self.normalize_input:
            for k, v in self.obs_stats.items():
                sd['running_mean_std.' + k] = v.clone()
        return sd

    def load_full_state_dict(self, sd):
        net = {k[len('a2c_network.'):]: v for k, v in sd.items() if k.startswith('a2c_network.')}
        self.a2c_network.load_state_dict(net)
        for prefix, stats in (('value_mean_std.', self.value_stats), ('running_mean_std.', self.obs_stats)):
            for k in list(stats):
                if prefix + k in sd:
                    stats[k] = sd[prefix + k].clone()

    def norm_obs(self, obs):
        if not self.normalize_input:
            return obs
        with torch.no_grad():
            y, self.obs_stats = O.running_stats_forward(self.obs_stats, obs, self.obs_stats_training)
            return y

    def act(self, obs):
        """Rollout forward (is_train False): models.py:348-359."""
        mu, logstd, value = self.a2c_network(self.norm_obs(obs))
        sigma = torch.exp(logstd)
        action = torch.distributions.Normal(mu, sigma, validate_args=False).sample()
        nlp = O.neglogp(action, mu, sigma, logstd)
        if self.normalize_value:
            value, _ = O.running_stats_forward(self.value_stats, value, False, denorm=True)
        return {'neglogpacs': torch.squeeze(nlp), 'values': value, 'actions': action, 'mus': mu,
                'sigmas': sigma}


class OracleAgent:
    def __init__(self, params, env, seed=None):
        cfg = params['config']
        net = params['network']
        self.cfg = cfg
        self.env = env
        info = env.get_env_info()
        self.obs_dim = info['observation_space'].shape[0]
        self.act_dim = info['action_space'].shape[0]
        self.N, self.H = cfg['num_actors'], cfg['horizon_length']
        self.mb = cfg['minibatch_size']
        self.B = self.N * self.H
        self.mini_epochs = cfg['mini_epochs']
        if seed is not None:
            torch.manual_seed(seed)
        self.model = OracleModel(self.obs_dim, self.act_dim, net['mlp']['units'], cfg['normalize_input'],
                                 cfg.get('normalize_value', False))
        self.lr = float(cfg['learning_rate'])
        self.optimizer = torch.optim.Adam(self.model.a2c_network.parameters(), self.lr, eps=1e-08)
        self.gamma, self.tau = cfg['gamma'], cfg['tau']
        self.hp = {'e_clip': cfg['e_clip'], 'critic_coef': cfg['critic_coef'], 'entropy_coef': cfg['entropy_coef'],
                   'bounds_loss_coef': cfg.get('bounds_loss_coef'), 'clip_value': cfg['clip_value'],
                   'use_smooth_clamp': cfg.get('use_smooth_clamp', False),
                   'bound_loss_type': cfg.get('bound_loss_type', 'bound'), 'ppo': cfg.get('ppo', True)}
        self.adaptive = cfg.get('lr_schedule') == 'adaptive'
        self.ema_state = O.new_moving_stats(1) if (cfg['normalize_advantage'] and
                                                    cfg.get('normalize_rms_advantage', False)) else None
        self.low = torch.from_numpy(np.asarray(info['action_space'].low).copy()).float()
        self.high = torch.from_numpy(np.asarray(info['action_space'].high).copy()).float()
        shaper = cfg.get('reward_shaper', {})
        self.shaper = dict(scale=shaper.get('scale_value', 1.0), shift=shaper.get('shift_value', 0.0))
        self.cur_r = torch.zeros(self.N, 1)
        self.cur_s = torch.zeros(self.N, 1)
        self.cur_l = torch.zeros(self.N)
        self.dones = torch.ones(self.N, dtype=torch.uint8)
        self.meters = {'mean': [torch.zeros(1), torch.zeros(1), torch.zeros(1)], 'n': [0, 0, 0]}
        self.games_to_track = cfg.get('games_to_track', 100)
        self.obs = None

    # ------------------------------------------------------------------ rollout
    def play_steps(self):
        N, H = self.N, self.H
        buf = {'obses': torch.zeros(H, N, self.obs_dim), 'rewards': torch.zeros(H, N, 1),
               'values': torch.zeros(H, N, 1), 'neglogpacs': torch.zeros(H, N),
               'dones': torch.zeros(H, N, dtype=torch.uint8), 'actions': torch.zeros(H, N, self.act_dim),
               'mus': torch.zeros(H, N, self.act_dim), 'sigmas': torch.zeros(H, N, self.act_dim)}
        self.model.obs_stats_training = False
        with torch.no_grad():
            for n in range(H):
                res = self.model.act(self.obs)
                buf['obses'][n, :] = self.obs
                buf['dones'][n, :] = self.dones
                for k in ('actions', 'neglogpacs', 'values', 'mus', 'sigmas'):
                    buf[k][n, :] = res[k]
                act = torch.clamp(res['actions'], -1.0, 1.0)
                d, m = (self.high - self.low) / 2.0, (self.high + self.low) / 2.0
                self.obs, rewards, dones, infos = self.env.step(act * d + m)
                rewards = rewards.unsqueeze(1)
                self.dones = dones
                shaped = O.shape_rewards(rewards, **self.shaper)
                if self.cfg.get('value_bootstrap', True) and 'time_outs' in infos:
                    shaped = O.bootstrap_timeouts(shaped, res['values'], infos['time_outs'], self.gamma)
                buf['rewards'][n, :] = shaped
                self.cur_r, self.cur_s, self.cur_l, fin = O.episode_bookkeeping(
                    self.cur_r, self.cur_s, self.cur_l, rewards, shaped, self.dones)
                for j, vals in enumerate(fin[:3]):
                    vals = vals.reshape(vals.shape[0], 1)      # value_size 1: rewards [n,1], lengths [n]
                    self.meters['mean'][j], self.meters['n'][j] = O.average_meter_update(
                        self.meters['mean'][j], self.meters['n'][j], vals, self.games_to_track)
            last_values = self.model.act(self.obs)['values']
            fdones = self.dones.float()
            advs = O.gae_scan(buf['rewards'], buf['values'], buf['dones'].float(), last_values, fdones,
                              self.gamma, self.tau)
            returns = O.returns_from_advantages(advs, buf['values'])
        batch = {k: O.flatten_env_major(buf[k]) for k in ('actions', 'neglogpacs', 'values', 'mus', 'sigmas',
                                                          'obses', 'dones')}
        batch['returns'] = O.flatten_env_major(returns)
        self.last_buffers = buf
        self.last_values, self.last_dones = last_values, self.dones
        return batch

    # ------------------------------------------------------------------ update
    def prepare_dataset(self, batch):
        cfg = self.cfg
        out = O.prepare_dataset(batch['returns'], batch['values'], self.model.value_stats,
                                normalize_value=cfg.get('normalize_value', False),
                                normalize_advantage=cfg['normalize_advantage'],
                                mask=batch.get('rnn_masks'), adv_ema_state=self.ema_state,
                                adv_ema_decay=cfg.get('adv_rms_momentum', 0.5))
        self.model.value_stats = out['value_stats']
        if 'adv_ema_state' in out:
            self.ema_state = out['adv_ema_state']
        self.dataset = {'old_values': out['old_values'], 'old_logp_actions': batch['neglogpacs'],
                        'advantages': out['advantages'], 'returns': out['returns'],
                        'actions': batch['actions'], 'obs': batch['obses'], 'mu': batch['mus'].clone(),
                        'sigma': batch['sigmas'].clone(), 'rnn_masks': batch.get('rnn_masks')}
        return self.dataset

    def minibatch_backward(self, i):
        """calc_gradients up to loss.backward() (a2c_continuous.py:136-211): the unclipped gradients of minibatch i are left
        in p.grad; KL (:215-221) and the mu / sigma write-back (a2c_common.py:1556) as well - they only use this forward's
        outputs.  Nothing is stepped."""
        cfg, ds = self.cfg, self.dataset
        lo, hi = i * self.mb, (i + 1) * self.mb
        mbd = {k: (None if v is None else v[lo:hi]) for k, v in ds.items()}
        self.model.obs_stats_training = True
        net = self.model.a2c_network
        mu, logstd, values = net(self.model.norm_obs(mbd['obs']))
        sigma = torch.exp(logstd)
        entropy = O.normal_entropy(mu, sigma)
        nlp = torch.squeeze(O.neglogp(mbd['actions'], mu, sigma, logstd))
        mask = mbd.get('rnn_masks')
        loss, a, c, e, b = O.ppo_losses(
            mbd['old_logp_actions'], nlp, mbd['advantages'], mbd['old_values'], values, mbd['returns'],
            mu, entropy, self.hp['e_clip'], self.hp['critic_coef'], self.hp['entropy_coef'],
            self.hp['bounds_loss_coef'], self.hp['clip_value'], mask, self.hp['use_smooth_clamp'],
            self.hp['bound_loss_type'], self.hp['ppo'])
        for p in net.parameters():
            p.grad = None
        loss.backward()
        with torch.no_grad():
            kl = O.policy_kl(mu.detach(), sigma.detach(), mbd['mu'], mbd['sigma'], mask)
        ds['mu'][lo:hi] = mu.detach()
        ds['sigma'][lo:hi] = sigma.detach()
        return {'a_loss': a.detach(), 'c_loss': c.detach(), 'entropy': e.detach(), 'b_loss': b.detach(), 'kl': kl}

    def minibatch_apply(self, kl_value):
        """trancate_gradients_and_step behind the (optional) gradient exchange (a2c_common.py:510-514) + the per-minibatch
        lr rule on `kl_value` (:1557-1563, python floats).  Returns the lr the step used."""
        cfg = self.cfg
        net = self.model.a2c_network
        if cfg.get('truncate_grads', False):
            nn.utils.clip_grad_norm_(net.parameters(), cfg['grad_norm'])
        self.optimizer.step()
        lr_used = self.lr
        if self.adaptive:
            self.lr = O.adaptive_lr(self.lr, kl_value, cfg['kl_threshold'], cfg.get('min_lr', 1e-6),
                                    cfg.get('max_lr', 1e-2), cfg.get('lr_multiplier', 1.5))
            for g in self.optimizer.param_groups:
                g['lr'] = self.lr
        return lr_used

    def minibatch_step(self, i):
        """calc_gradients + trancate_gradients_and_step + per-minibatch lr (python floats)."""
        out = self.minibatch_backward(i)
        out['lr'] = self.minibatch_apply(out['kl'].item())
        return out

    def update(self, batch):
        self.prepare_dataset(batch)
        results = []
        for _ in range(self.mini_epochs):
            for i in range(self.B // self.mb):
                results.append(self.minibatch_step(i))
        return results

    def train_epoch(self):
        if self.obs is None:
            self.obs = self.env.reset()
        t0 = time.perf_counter()
        batch = self.play_steps()
        t1 = time.perf_counter()
        results = self.update(batch)
        t2 = time.perf_counter()
        return {'play_time': t1 - t0, 'update_time': t2 - t1, 'total_time': t2 - t0, 'results': results}


def data_parallel_minibatch_step(agents, i):
    """One optimiser step of `len(agents)` data-parallel ranks (one OracleAgent per rank, each on its own env shard's
    dataset, the same parameters): what A2CBase.trancate_gradients_and_step does under multi_gpu
    (rl_games/common/a2c_common.py:493-514) - every gradient summed over the ranks (the all_reduce of the flattened
    buffer), divided by world_size, copied back into p.grad, THEN clip_grad_norm_ and the Adam step on every rank - and
    the per-minibatch lr rule on the KL averaged over the ranks (:1557-1563: all_reduce(kl, SUM); av_kls /= world_size).
    The observation statistics stay per rank inside the epoch (merge_rank_stats pools them once per epoch, :1578).
    Returns the per-rank dicts of minibatch_backward plus 'lr' and 'kl_mean'."""
    world = len(agents)
    outs = [a.minibatch_backward(i) for a in agents]
    for ps in zip(*[list(a.model.a2c_network.parameters()) for a in agents]):
        total = ps[0].grad.detach().clone()
        for p in ps[1:]:
            total += p.grad                      # (rank order; two ranks: commutative, any order gives these bits)
        for p in ps:
            p.grad = total / world
    kl = outs[0]['kl'].clone()
    for o in outs[1:]:
        kl += o['kl']
    kl /= world
    for a, o in zip(agents, outs):
        o['kl_mean'] = kl
        o['lr'] = a.minibatch_apply(kl.item())
    return outs
