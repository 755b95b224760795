"""Actor-critic policy used by the MI355X PPO agent.

Host-side mirror of the part of the reference's model zoo that the BASELINE configs use:
`ModelA2CContinuousLogStd` (rl_games/algos_torch/models.py:305-364) over the `actor_critic`
network of `A2CBuilder` (rl_games/algos_torch/network_builder.py:218-589) - shared MLP trunk,
`mu` / `value` heads, fixed-sigma parameter, optional LSTM/GRU after the MLP.  Module and
parameter names match the reference (`a2c_network.actor_mlp.<2i>.weight`, `a2c_network.mu`,
`a2c_network.value`, `a2c_network.sigma`, `running_mean_std.*`, `value_mean_std.*`), so
`state_dict()`s are interchangeable with reference checkpoints.

The module tree only HOLDS the parameters (views of the optimiser's flat arena): for the MLP and LSTM
configurations every product of rollout and update runs in this library's fused MFMA kernels
(`mlp_engine.ManualMlpEngine` over `ops.MlpChain` / `ops.MlpDwPlan`, csrc/mlp_chain*.hip, mlp_dw.hip), the
normalisers and - in training - the whole distribution/loss epilogue in its HIP kernels as well.  The
`torch.nn.Linear` forward of the modules (library GEMMs) is what runs for shapes outside the engines' envelope
(`rl_games_amd: manual MLP engine unavailable (...)` is printed when that happens) and on the CPU for checkpoints.
`forward(input_dict)` keeps the reference's dict-in/dict-out contract; `forward_heads` is the
raw (mu, logstd, value) path the fused loss kernel consumes.
"""
import math

import numpy as np
import torch
from torch import nn

from .normalizers import RunningMeanStd

_ACTIVATIONS = {
    'relu': nn.ReLU, 'tanh': nn.Tanh, 'sigmoid': nn.Sigmoid, 'elu': nn.ELU, 'selu': nn.SELU,
    'swish': nn.SiLU, 'gelu': nn.GELU, 'softplus': nn.Softplus, 'None': nn.Identity, None: nn.Identity,
}


def _activation(name):
    if name not in _ACTIVATIONS:
        raise ValueError(f'unknown activation {name}')
    return _ACTIVATIONS[name]()


def _initializer(spec):
    """network_builder.py:60-70: name -> in-place init function."""
    spec = dict(spec or {'name': 'default'})
    name = spec.pop('name', 'default')
    if name == 'default':
        return lambda t: t
    if name == 'const_initializer':
        val = spec.get('val', spec.get('value', 0))
        return lambda t: nn.init.constant_(t, val)
    table = {
        'orthogonal_initializer': nn.init.orthogonal_, 'orthogonal': nn.init.orthogonal_,
        'glorot_normal_initializer': nn.init.xavier_normal_,
        'glorot_uniform_initializer': nn.init.xavier_uniform_,
        'random_uniform_initializer': nn.init.uniform_, 'kaiming_normal': nn.init.kaiming_normal_,
    }
    if name not in table:
        raise NotImplementedError(f'initializer {name}')
    fn = table[name]
    return lambda t: fn(t, **spec)


class RnnWithDones(nn.Module):
    """LSTM/GRU that resets its state where `dones` is set, without the per-call `.cpu()`
    done search of the reference (rl_games/common/layers/recurrent.py:26-58): the state is
    multiplied by (1 - done) before every timestep, which yields the same outputs as running
    the fused RNN between done positions.  Parameter names follow torch.nn.LSTM (`rnn.rnn.*`),
    like the reference's wrapper."""

    def __init__(self, kind, input_size, hidden_size, num_layers):
        super().__init__()
        self.kind = kind
        cls = nn.LSTM if kind == 'lstm' else nn.GRU
        self.rnn = cls(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers)

    def forward(self, x, states, dones=None, bptt_len=0):
        # x: [T, B, F]; states: tuple (h, c) or (h,) / tensor
        if self.kind == 'lstm':
            st = tuple(states) if isinstance(states, (tuple, list)) else states
        else:
            st = states[0] if isinstance(states, (tuple, list)) else states
        if dones is None:
            return self.rnn(x, st)
        outs = []
        T = x.shape[0]
        for t in range(T):
            keep = (1.0 - dones[t].float()).reshape(1, -1, 1)
            if self.kind == 'lstm':
                st = (st[0] * keep, st[1] * keep)
            else:
                st = st * keep
            o, st = self.rnn(x[t:t + 1], st)
            outs.append(o)
        return torch.cat(outs, dim=0), st


def _norm_layer(name, width):
    """network_builder.py:126-129, d2rl.py:17-22: LayerNorm, BatchNorm1d (train / eval as the module's mode says: batch
    statistics in the update, running ones in the rollout), or nothing - any other name builds no layer there either."""
    if name == 'layer_norm':
        return nn.LayerNorm(width)
    if name == 'batch_norm':
        return nn.BatchNorm1d(width)
    return None


class D2RLNet(nn.Module):
    """Dense-to-dense trunk (rl_games/algos_torch/d2rl.py:3-33): every layer behind the first sees the previous layer's
    output concatenated with the network input.  Parameter names as in the reference (`linears.N`, `norm_layers.N`)."""

    def __init__(self, input_size, units, activation, norm_func_name=None):
        super().__init__()
        self.activations = nn.ModuleList([_activation(activation) for _ in units])
        self.linears = nn.ModuleList([])
        self.norm_layers = nn.ModuleList([])
        self.num_layers = len(units)
        last = input_size
        for u in units:
            self.linears.append(nn.Linear(last, u))
            last = u + input_size
            self.norm_layers.append(_norm_layer(norm_func_name, u) or nn.Identity())

    def forward(self, x0):
        x = self.norm_layers[0](self.activations[0](self.linears[0](x0)))            # (:25-27: act, then norm)
        for i in range(1, self.num_layers):
            x = self.activations[i](self.norm_layers[i](self.linears[i](torch.cat([x, x0], dim=1))))   # (:29-32: norm, then act)
        return x


def build_trunk(input_size, units, activation, norm_func_name=None, norm_only_first_layer=False, d2rl=False):
    """network_builder.py:105-145 (_build_sequential_mlp / _build_mlp), statement for statement - including that `in_size` is
    only advanced while a normalisation layer is still to come when `norm_only_first_layer` is set."""
    if d2rl:
        return D2RLNet(input_size, units, activation, norm_func_name)
    in_size, layers, need_norm = input_size, [], True
    for unit in units:
        layers.append(nn.Linear(in_size, unit))
        layers.append(_activation(activation))
        if not need_norm:
            continue
        if norm_only_first_layer and norm_func_name is not None:
            need_norm = False
        norm = _norm_layer(norm_func_name, unit)
        if norm is not None:
            layers.append(norm)
        in_size = unit
    return nn.Sequential(*layers)


class ActorCriticNetwork(nn.Module):
    """`a2c_network`: MLP trunk (+ optional RNN) with mu / value heads.  The plain layouts - Linear + activation trunks,
    a single-layer LSTM behind the trunk - run on this library's fused kernels (`plain_trunk`); layer normalisation, D2RL
    trunks, an RNN in front of the MLP, concatenated RNN inputs / outputs and layer norm behind the RNN (round 6) are
    built like the reference builds them and run as torch modules with autograd between the HIP rollout, dataset, loss
    and optimiser kernels."""

    def __init__(self, net_params, actions_num, input_shape, value_size=1, num_seqs=1):
        super().__init__()
        if 'cnn' in net_params:
            raise NotImplementedError('CNN encoders are outside the MI355X PPO hot path (SURVEY 2, row 15)')
        self.separate = bool(net_params.get('separate', False))
        self.central_value = bool(net_params.get('central_value', False))
        if self.central_value:
            self._init_central_value(net_params, input_shape, value_size, num_seqs)
            return
        space = net_params.get('space', {})
        self.is_multi_discrete = 'multi_discrete' in space
        self.is_discrete = 'discrete' in space or self.is_multi_discrete
        if self.is_discrete:
            self._init_discrete(net_params, actions_num, input_shape, value_size, num_seqs)
            return
        if 'continuous' not in space:
            raise NotImplementedError("network 'space' must be continuous or discrete")
        self.space_config = space['continuous']
        # fixed_sigma False (round 6): a Linear sigma head over the trunk's features (network_builder.py:341-344, :508-511).
        # The fused kernels carry log sigma as a parameter vector, so such a policy runs the reference's operation sequence
        # as torch ops with autograd around this library's rollout, GAE, dataset and optimiser kernels
        # (agent._forward_loss_backward_general, torch_fallback.py; pinned to the reference on the CPU).
        self.fixed_sigma = bool(self.space_config['fixed_sigma'])
        # how the sigma head's raw output becomes sigma (network_builder.py:311-322, models.py:272-301): the fused kernels
        # know `exp` of an unbounded log sigma; a floor, bounds or the softplus / linear forms run as torch ops like a
        # state-dependent head does (`plain_sigma` False)
        self.min_sigma = float(self.space_config.get('min_sigma', 0.0))
        self.logstd_bounds = self.space_config.get('logstd_bounds', None)
        self.sigma_parametrization = self.space_config.get('sigma_parametrization', 'exp')
        if self.sigma_parametrization == 'scalar':
            self.sigma_parametrization = 'linear'
        if self.sigma_parametrization not in ('exp', 'softplus', 'linear'):
            raise NotImplementedError(f"sigma_parametrization '{self.sigma_parametrization}'")
        self.plain_sigma = (self.sigma_parametrization == 'exp' and self.logstd_bounds is None and self.min_sigma <= 0)
        mlp = net_params['mlp']
        self.value_size = value_size
        self.num_seqs = num_seqs
        self.actions_num = actions_num
        assert len(input_shape) == 1, 'flat observations only'
        in_size = input_shape[0]
        out_size = self._build_body(net_params, in_size)
        self.value = nn.Linear(out_size, value_size)
        self.value_act = _activation(net_params.get('value_activation', 'None'))
        self.mu = nn.Linear(out_size, actions_num)
        self.mu_act = _activation(self.space_config['mu_activation'])
        self.sigma_act = _activation(self.space_config['sigma_activation'])
        if self.fixed_sigma:
            self.sigma = nn.Parameter(torch.zeros(actions_num, dtype=torch.float32), requires_grad=True)
        else:
            self.sigma = nn.Linear(out_size, actions_num)

        mlp_init = _initializer(mlp['initializer'])
        for m in self.modules():
            if isinstance(m, nn.Linear):
                mlp_init(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
        _initializer(self.space_config['mu_init'])(self.mu.weight)
        if self.fixed_sigma:
            _initializer(self.space_config['sigma_init'])(self.sigma)
        elif (self.space_config.get('sigma_init') or {}).get('name') == 'const_initializer':
            # init_state_dependent_sigma_head (network_builder.py:14-25): a constant initialiser means a uniform initial
            # log sigma - the BIAS - over zeroed weights
            si = self.space_config['sigma_init']
            nn.init.zeros_(self.sigma.weight)
            nn.init.constant_(self.sigma.bias, si.get('val', si.get('value', 0.0)))
        else:
            _initializer(self.space_config['sigma_init'])(self.sigma.weight)

    def _build_body(self, net_params, in_size):
        """MLP trunk(s) + optional RNN of a policy network (network_builder.py:250-296); returns the width the heads read."""
        mlp = net_params['mlp']
        self._trunk_kw = dict(activation=mlp['activation'], norm_func_name=net_params.get('normalization'),
                              norm_only_first_layer=mlp.get('norm_only_first_layer', False), d2rl=mlp.get('d2rl', False))
        self.units = list(mlp['units'])
        self.has_rnn = 'rnn' in net_params
        # sizes as network_builder.py:250-272
        mlp_in = in_size
        out_size = self.units[-1] if self.units else in_size
        self.rnn_before_mlp = self.rnn_concat_input = self.rnn_concat_output = self.rnn_ln = False
        if self.has_rnn:
            rnn = net_params['rnn']
            self.rnn_name = rnn['name']
            self.rnn_units = rnn['units']
            self.rnn_layers = rnn['layers']
            self.rnn_ln = bool(rnn.get('layer_norm', False))
            self.rnn_before_mlp = bool(rnn.get('before_mlp', False))
            self.rnn_concat_input = bool(rnn.get('concat_input', False))
            self.rnn_concat_output = bool(rnn.get('concat_output', False))
            if not self.rnn_before_mlp:
                rnn_in = out_size + (in_size if self.rnn_concat_input else 0)
                out_size = self.rnn_units + (in_size if self.rnn_concat_output else 0)
            else:
                rnn_in = in_size
                mlp_in = self.rnn_units + (in_size if self.rnn_concat_output else 0)
            if self.separate:                                  # one RNN per trunk (network_builder.py:272-277)
                self.a_rnn = RnnWithDones(self.rnn_name, rnn_in, self.rnn_units, self.rnn_layers)
                self.c_rnn = RnnWithDones(self.rnn_name, rnn_in, self.rnn_units, self.rnn_layers)
                if self.rnn_ln:
                    self.a_layer_norm = nn.LayerNorm(self.rnn_units)
                    self.c_layer_norm = nn.LayerNorm(self.rnn_units)
            else:
                self.rnn = RnnWithDones(self.rnn_name, rnn_in, self.rnn_units, self.rnn_layers)
                if self.rnn_ln:
                    self.layer_norm = nn.LayerNorm(self.rnn_units)
        self.actor_mlp = build_trunk(mlp_in, self.units, **self._trunk_kw)
        if self.separate:                                      # network_builder.py:292-293
            self.critic_mlp = build_trunk(mlp_in, self.units, **self._trunk_kw)
        # what the fused engines take (mlp_engine.ManualMLP): Linear + activation pairs, the RNN plainly behind them
        self.plain_trunk = (not self._trunk_kw['d2rl'] and self._trunk_kw['norm_func_name'] is None and
                            not (self.rnn_before_mlp or self.rnn_concat_input or self.rnn_concat_output or self.rnn_ln)
                            and not (self.separate and self.has_rnn))
        return out_size

    def _init_central_value(self, net_params, input_shape, value_size, num_seqs):
        """`central_value: True` networks (network_builder.py:497,556): MLP trunk + value head only."""
        mlp = net_params['mlp']
        if 'cnn' in net_params:
            raise NotImplementedError('central value network: MLP (+ RNN) only on this path')
        self.is_discrete = False
        self.separate = False
        self.value_size, self.num_seqs, self.actions_num = value_size, num_seqs, 0
        assert len(input_shape) == 1, 'flat states only'
        last = self._build_body(net_params, input_shape[0])      # (round 6: layer norm / D2RL trunks and an RNN as well)
        self.value = nn.Linear(last, value_size)
        self.value_act = _activation(net_params.get('value_activation', 'None'))
        mlp_init = _initializer(mlp['initializer'])
        for m in self.modules():
            if isinstance(m, nn.Linear):
                mlp_init(m.weight)
                nn.init.zeros_(m.bias)

    def _init_discrete(self, net_params, actions_num, input_shape, value_size, num_seqs):
        """Categorical head (network_builder.py:298-299): `logits` Linear in place of mu/sigma."""
        mlp = net_params['mlp']
        self.value_size, self.num_seqs, self.actions_num = value_size, num_seqs, actions_num
        assert len(input_shape) == 1, 'flat observations only'
        last = self._build_body(net_params, input_shape[0])      # (an RNN as for the continuous policy, round 6)
        self.value = nn.Linear(last, value_size)
        self.value_act = _activation(net_params.get('value_activation', 'None'))
        if self.is_multi_discrete:                              # network_builder.py:303-304
            self.logits = nn.ModuleList([nn.Linear(last, int(n)) for n in actions_num])
        else:
            self.logits = nn.Linear(last, actions_num)
        mlp_init = _initializer(mlp['initializer'])
        for m in self.modules():
            if isinstance(m, nn.Linear):
                mlp_init(m.weight)
                nn.init.zeros_(m.bias)

    def head_logits(self, out):
        """Single head: tensor [B, n]; multi-discrete: list of [B, n_b] (network_builder.py:431-436)."""
        if self.is_multi_discrete:
            return [head(out) for head in self.logits]
        return self.logits(out)

    def _mlp_like(self, in_size, activation):
        layers, last = [], in_size
        for u in self.units:
            layers += [nn.Linear(last, u), _activation(activation)]
            last = u
        return nn.Sequential(*layers)

    def is_rnn(self):
        return self.has_rnn

    def is_separate_critic(self):
        return self.separate

    def get_value_layer(self):
        return self.value

    def get_aux_loss(self):
        return None

    def get_default_rnn_state(self):
        if not self.has_rnn:
            return None
        z = lambda: torch.zeros((self.rnn_layers, self.num_seqs, self.rnn_units))
        per = 2 if self.rnn_name == 'lstm' else 1               # separate trunks: the actor's states, then the critic's
        return tuple(z() for _ in range(per * (2 if self.separate else 1)))

    def _body(self, mlp, rnn, norm, obs, states, dones, seq_length):
        """One trunk with its RNN (network_builder.py:447-500; :372-421 runs two of them side by side)."""
        out = obs
        if not self.rnn_before_mlp:
            out = mlp(out)
            if self.rnn_concat_input:
                out = torch.cat([out, obs], dim=1)
        batch = out.shape[0]
        num_seqs = batch // seq_length
        out = out.reshape(num_seqs, seq_length, -1).transpose(0, 1)
        if dones is not None:
            dones = dones.reshape(num_seqs, seq_length, -1).transpose(0, 1)
        if states is None:
            states = tuple()
        out, states = rnn(out, states, dones)
        out = out.transpose(0, 1).contiguous().reshape(batch, -1)
        if norm is not None:
            out = norm(out)
        if self.rnn_concat_output:
            out = torch.cat([out, obs], dim=1)
        if self.rnn_before_mlp:
            out = mlp(out)
        if not isinstance(states, tuple):
            states = (states,)
        return out, states

    def bodies(self, obs, states=None, dones=None, seq_length=1):
        """(actor features, critic features, next rnn states): one shared trunk, or `separate` ones - each with its own
        RNN, the states of the actor's in front of the critic's (network_builder.py:372-421, :447-500)."""
        if not self.has_rnn:
            out = self.actor_mlp(obs)
            return out, (self.critic_mlp(obs) if self.separate else out), states
        if not self.separate:
            out, states = self._body(self.actor_mlp, self.rnn, self.layer_norm if self.rnn_ln else None, obs, states,
                                     dones, seq_length)
            return out, out, states
        half = len(states) // 2
        a_out, a_states = self._body(self.actor_mlp, self.a_rnn, self.a_layer_norm if self.rnn_ln else None, obs,
                                     tuple(states[:half]), dones, seq_length)
        c_out, c_states = self._body(self.critic_mlp, self.c_rnn, self.c_layer_norm if self.rnn_ln else None, obs,
                                     tuple(states[half:]), dones, seq_length)
        return a_out, c_out, a_states + c_states

    def forward(self, obs_dict):
        """(mu, logstd_broadcast, value, states) - network_builder.py:447-512."""
        out, c_out, states = self.bodies(obs_dict['obs'], obs_dict.get('rnn_states'), obs_dict.get('dones'),
                                         obs_dict.get('seq_length', 1))
        value = self.value_act(self.value(c_out))
        if self.central_value:                                  # (value, states) :497-498
            return value, states
        if self.is_discrete:                                   # (logits, value, states) :431-436,:500-504
            return self.head_logits(out), value, states
        mu = self.mu_act(self.mu(out))
        return mu, mu * 0 + self.logstd_of(out), value, states

    def sigma_and_logstd(self, raw):
        """(sigma, log sigma) of the sigma head's raw output - `apply_sigma_parametrization` (models.py:272-301)."""
        if self.sigma_parametrization == 'softplus':
            sigma = torch.nn.functional.softplus(raw) + self.min_sigma
        elif self.sigma_parametrization == 'linear':            # sigma ~ raw above a smooth floor
            floor = max(self.min_sigma, 1e-3)
            sigma = floor + torch.nn.functional.softplus(raw - floor)
        else:
            if self.logstd_bounds is not None:
                raw = torch.clamp(raw, self.logstd_bounds[0], self.logstd_bounds[1])
            sigma = torch.exp(raw)
            if self.min_sigma <= 0:
                return sigma, raw
            sigma = sigma + self.min_sigma
        return sigma, torch.log(sigma)

    def logstd_of(self, out):
        """log sigma: the parameter vector [A], or the sigma head of the trunk's features [B, A] (network_builder.py:508-511)."""
        return self.sigma_act(self.sigma) if self.fixed_sigma else self.sigma_act(self.sigma(out))


class ContinuousA2CLogStdModel(nn.Module):
    """The reference's `ModelA2CContinuousLogStd.Network` contract (models.py:311-364)."""

    def __init__(self, a2c_network, obs_shape, normalize_value, normalize_input, value_size):
        super().__init__()
        self.obs_shape = obs_shape
        self.normalize_value = normalize_value
        self.normalize_input = normalize_input
        self.value_size = value_size
        if normalize_value:
            self.value_mean_std = RunningMeanStd((value_size,))
        if normalize_input:
            if isinstance(obs_shape, dict):
                raise NotImplementedError('dict observations are not implemented on this path')
            self.running_mean_std = RunningMeanStd(obs_shape)
        self.a2c_network = a2c_network        # after the normalisers, as in models.py:27-45,:311-316

    # -- reference API ---------------------------------------------------------------
    def is_rnn(self):
        return self.a2c_network.is_rnn()

    def get_default_rnn_state(self):
        return self.a2c_network.get_default_rnn_state()

    def get_value_layer(self):
        return self.a2c_network.get_value_layer()

    def get_aux_loss(self):
        return None

    def norm_obs(self, observation):
        with torch.no_grad():
            return self.running_mean_std(observation) if self.normalize_input else observation

    def denorm_value(self, value):
        with torch.no_grad():
            return self.value_mean_std(value, denorm=True) if self.normalize_value else value

    @staticmethod
    def neglogp(x, mean, std, logstd):
        return 0.5 * (((x - mean) / std) ** 2).sum(dim=-1) \
            + 0.5 * np.log(2.0 * np.pi) * x.size(-1) + logstd.sum(dim=-1)

    def forward_heads(self, input_dict):
        """Training fast path: normalise (and update) observations, run the trunk, return the
        raw heads (mu [B,A], logstd - the parameter [A] or, with fixed_sigma False, the sigma head's [B,A] - value [B,V],
        rnn states)."""
        obs = self.norm_obs(input_dict['obs'])
        net = self.a2c_network
        out, c_out, states = net.bodies(obs, input_dict.get('rnn_states'), input_dict.get('dones'),
                                        input_dict.get('seq_length', 1))
        value = net.value_act(net.value(c_out))
        mu = net.mu_act(net.mu(out))
        return mu, net.logstd_of(out), value, states

    # The reference Runner wraps `agent.model` in torch.compile unless the config says otherwise (torch_runner.py:283-307).
    # The normalisers of this model launch HIP kernels through ctypes on torch's current stream - nothing Dynamo can trace -
    # so the forward is marked opaque: the wrapper then calls it eagerly (the rollout's slow path, get_action_values /
    # get_values of an agent outside the fused envelope, players).
    @torch.compiler.disable
    def forward(self, input_dict):
        is_train = input_dict.get('is_train', True)
        prev_actions = input_dict.get('prev_actions', None)
        input_dict = dict(input_dict)
        input_dict['obs'] = self.norm_obs(input_dict['obs'])
        mu, logstd, value, states = self.a2c_network(input_dict)
        sigma, logstd = self.a2c_network.sigma_and_logstd(logstd)
        if is_train:
            entropy = (0.5 + 0.5 * math.log(2 * math.pi) + torch.log(sigma)).sum(dim=-1)
            prev_neglogp = self.neglogp(prev_actions, mu, sigma, logstd)
            return {'prev_neglogp': torch.squeeze(prev_neglogp), 'values': value, 'entropy': entropy,
                    'rnn_states': states, 'mus': mu, 'sigmas': sigma}
        selected_action = torch.normal(mu, sigma)
        neglogp = self.neglogp(selected_action, mu, sigma, logstd)
        return {'neglogpacs': torch.squeeze(neglogp), 'values': self.denorm_value(value),
                'actions': selected_action, 'rnn_states': states, 'mus': mu, 'sigmas': sigma}


class DiscreteA2CModel(ContinuousA2CLogStdModel):
    """The reference's `ModelA2C.Network` / `ModelA2CMultiDiscrete.Network` contracts
    (models.py:66-125, :128-206): Categorical heads (one, or one per sub-action), optional action
    masks with CategoricalMasked semantics (common/extensions/distributions.py:24-47)."""

    @staticmethod
    def _dist(logits, masks):
        if masks is None:
            cat = torch.distributions.Categorical(logits=logits)
            return cat, cat.entropy()
        floor = torch.tensor(-1e+8, dtype=logits.dtype, device=logits.device)
        cat = torch.distributions.Categorical(logits=torch.where(masks, logits, floor))
        p_log_p = torch.where(masks, cat.logits * cat.probs, torch.zeros((), dtype=logits.dtype, device=logits.device))
        return cat, -p_log_p.sum(-1)

    def branch_sizes(self):
        net = self.a2c_network
        if net.is_multi_discrete:
            return [head.out_features for head in net.logits]
        return [net.logits.out_features]

    def forward_heads(self, input_dict):
        """Training fast path: (logits [B, sum(branch sizes)] with the heads concatenated, value
        [B, V]) for the fused categorical loss kernel."""
        obs = self.norm_obs(input_dict['obs'])
        net = self.a2c_network
        out, c_out, _ = net.bodies(obs, input_dict.get('rnn_states'), input_dict.get('dones'),
                                   input_dict.get('seq_length', 1))
        logits = net.head_logits(out)
        if net.is_multi_discrete:
            logits = torch.cat(logits, dim=1)
        return logits, net.value_act(net.value(c_out))

    @torch.compiler.disable       # (see ContinuousA2CLogStdModel: launches behind ctypes are opaque to Dynamo)
    def forward(self, input_dict):
        is_train = input_dict.get('is_train', True)
        action_masks = input_dict.get('action_masks', None)
        prev_actions = input_dict.get('prev_actions', None)
        input_dict = dict(input_dict)
        input_dict['obs'] = self.norm_obs(input_dict['obs'])
        logits, value, states = self.a2c_network(input_dict)
        multi = self.a2c_network.is_multi_discrete
        heads = logits if multi else [logits]
        if action_masks is None:
            masks = [None] * len(heads)
        else:
            masks = torch.split(action_masks.bool(), [h.shape[-1] for h in heads], dim=1)
        dists = [self._dist(h, m) for h, m in zip(heads, masks)]
        norm_logits = [d.logits for d, _ in dists]
        if is_train:
            acts = prev_actions.reshape(prev_actions.shape[0], len(heads))
            neglogp = sum(-d.log_prob(acts[:, b]) for b, (d, _) in enumerate(dists))
            entropy = sum(h for _, h in dists)
            return {'prev_neglogp': torch.squeeze(neglogp), 'logits': norm_logits if multi else norm_logits[0],
                    'values': value, 'entropy': torch.squeeze(entropy), 'rnn_states': states}
        selected = [d.sample().long() for d, _ in dists]
        neglogp = sum(-d.log_prob(a) for (d, _), a in zip(dists, selected))
        actions = torch.stack(selected, dim=-1) if multi else selected[0]
        return {'neglogpacs': torch.squeeze(neglogp), 'values': self.denorm_value(value), 'actions': actions,
                'logits': norm_logits if multi else norm_logits[0], 'rnn_states': states}


class CentralValueModel(ContinuousA2CLogStdModel):
    """The reference's `ModelCentralValue.Network` contract (models.py:425-464): normalised states in,
    {'values', 'rnn_states'} out; values de-normalised outside training."""

    @torch.compiler.disable       # (see ContinuousA2CLogStdModel: launches behind ctypes are opaque to Dynamo)
    def forward(self, input_dict):
        is_train = input_dict.get('is_train', True)
        inputs = dict(input_dict)
        inputs['obs'] = self.norm_obs(input_dict['obs'])
        value, states = self.a2c_network(inputs)                 # (rnn_states / dones / seq_length ride along)
        if not is_train:
            value = self.denorm_value(value)
        return {'values': value, 'rnn_states': states}

    def forward_values(self, obs, rnn=None):
        """Training fast path: normalise (and update) the states, return raw values [B, V].  rnn: None or the
        {'rnn_states', 'seq_length', 'dones'} of a sequence minibatch."""
        inputs = dict(rnn or {})
        inputs['obs'] = self.norm_obs(obs)
        value, _ = self.a2c_network(inputs)
        return value


class PolicyBuilder:
    """Stands in for `model_builder.ModelBuilder().load(params)` (rl_games/algos_torch/
    model_builder.py:56-60): `.build(config)` returns the model for one agent."""

    def __init__(self, params):
        model_name = params.get('model', {}).get('name', 'continuous_a2c_logstd')
        net_name = params.get('network', {}).get('name', 'actor_critic')
        if model_name not in ('continuous_a2c_logstd', 'discrete_a2c', 'multi_discrete_a2c', 'central_value'):
            raise NotImplementedError(f"model '{model_name}' is not implemented on the MI355X PPO path")
        self.model_name = model_name
        if net_name != 'actor_critic':
            raise NotImplementedError(f"network '{net_name}' is not implemented on the MI355X PPO path")
        self.net_params = params['network']

    def build(self, config):
        net = ActorCriticNetwork(self.net_params, actions_num=config['actions_num'],
                                 input_shape=config['input_shape'],
                                 value_size=config.get('value_size', 1),
                                 num_seqs=config.get('num_seqs', 1))
        if self.model_name == 'central_value' or net.central_value:
            if not net.central_value:
                raise ValueError("model 'central_value' needs a network with `central_value: True`")
            return CentralValueModel(net, obs_shape=config['input_shape'],
                                     normalize_value=config.get('normalize_value', False),
                                     normalize_input=config.get('normalize_input', False),
                                     value_size=config.get('value_size', 1))
        discrete_model = self.model_name in ('discrete_a2c', 'multi_discrete_a2c')
        cls = DiscreteA2CModel if discrete_model else ContinuousA2CLogStdModel
        if discrete_model != net.is_discrete or (self.model_name == 'multi_discrete_a2c') != net.is_multi_discrete:
            raise ValueError(f"model '{self.model_name}' does not match the network's action space")
        return cls(net, obs_shape=config['input_shape'],
                                        normalize_value=config.get('normalize_value', False),
                                        normalize_input=config.get('normalize_input', False),
                                        value_size=config.get('value_size', 1))
