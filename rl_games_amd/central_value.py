"""Central (asymmetric) value function on MI355X - host mirror of `CentralValueTrain`
(rl_games/algos_torch/central_value.py:14-383): a value-only network over the privileged `states`,
trained on its own minibatches before the actor's, whose predictions replace the actor's values in
the rollout (a2c_common.py:593-615) and therefore in GAE.

Same kernels as the actor path: the state normaliser and value de-normaliser are `RunningMeanStd`
(csrc/running_stats.hip), the clipped value loss + its gradient + masked mean is one fused kernel
(`rlg_value_loss`) reduced by `rlg_ppo_loss_finalize`, clipping + Adam run on a flat parameter arena
(`FlatAdam`, csrc/optim.hip) with one in-place all-reduce per step in multi-GPU runs.  Round 5: the MLP itself runs on
the actor path's fused chain kernels - state normaliser + hidden layers + value head as ONE forward launch, the dX /
activation-backward / bias-sum chain as ONE backward launch (csrc/mlp_chain*.hip), the hidden layers' weight gradients as
the MFMA launch of csrc/mlp_dw.hip (chain_net.ChainNet) - where the network has that form (plain Linear + ELU / ReLU /
tanh trunk, widths that are multiples of 4, one value column); anything else keeps autograd around the loss kernel
(`fused_mlp: False` in the central-value config forces that path).

Multi-agent envs (round 6, central_value.py:153-158,223-234): one state row per env, its value repeated for the env's agents
in the rollout, the critic trained on agent 0's values and returns.  Recurrent critics (round 6,
central_value.py:96-107,163-205): the network's RNN advances with the rollout (`pre_step_rnn` keeps the state every sequence
starts from, `post_step_rnn` / `zero_states_where` zero it where an episode ended) and trains on sequence minibatches as a
torch module between the loss and optimiser kernels.
"""
import torch
from torch import nn

from . import distributed as rdist
from . import ops
from .chain_net import ChainNet
from .flat_optim import FlatAdam
from .lr_control import IdentityScheduler, LinearScheduler
from .minibatch import PPODataset


def _value_chain(net, arena, max_rows):
    """Trunk + value head of a central value network on the fused chain kernels (chain_net.ChainNet): what autograd does
    for `CentralValueTrain.calc_gradients` (rl_games/algos_torch/central_value.py:278-335) between the loss kernel and
    the optimiser.  Raises NotImplementedError for networks outside the kernels' envelope (the caller keeps autograd)."""
    if not isinstance(net.value_act, nn.Identity) or net.value.out_features != 1:
        raise NotImplementedError('one linear value column only')
    if not getattr(net, 'plain_trunk', True) or net.is_rnn():
        raise NotImplementedError('plain Linear + activation trunks only')
    return ChainNet(net.actor_mlp, [net.value], arena, max_rows)


class CentralValueTrain(nn.Module):
    def __init__(self, state_shape, value_size, ppo_device, num_agents, horizon_length, num_actors,
                 num_actions, seq_length, normalize_value, network, config, writter, max_epochs, multi_gpu,
                 zero_rnn_on_done):
        super().__init__()
        self.ppo_device = ppo_device
        self.num_agents, self.horizon_length, self.num_actors = num_agents, horizon_length, num_actors
        self.seq_length, self.normalize_value, self.num_actions = seq_length, normalize_value, num_actions
        self.state_shape, self.value_size, self.max_epochs = state_shape, value_size, max_epochs
        self.multi_gpu, self.config = multi_gpu, config
        self.normalize_input = config['normalize_input']
        self.zero_rnn_on_done = zero_rnn_on_done
        self.model = network.build({
            'value_size': value_size, 'input_shape': state_shape, 'actions_num': num_actions,
            'num_agents': num_agents, 'num_seqs': num_actors, 'normalize_input': self.normalize_input,
            'normalize_value': self.normalize_value,
        }).to(ppo_device)
        self.is_rnn = self.model.is_rnn()
        self.rnn_states = None
        self.lr = float(config['learning_rate'])
        self.linear_lr = config.get('lr_schedule') == 'linear'
        if self.linear_lr:
            self.scheduler = LinearScheduler(self.lr, max_steps=self.max_epochs, apply_to_entropy=False,
                                             start_entropy_coef=0)
        else:
            self.scheduler = IdentityScheduler()
        self.mini_epoch = config['mini_epochs']
        if 'minibatch_size' not in config and 'minibatch_size_per_env' not in config:
            raise ValueError("Configuration must include either 'minibatch_size' or 'minibatch_size_per_env'. "
                             'Neither was found in the provided config.')
        self.minibatch_size_per_env = config.get('minibatch_size_per_env', 0)
        self.minibatch_size = config.get('minibatch_size', self.num_actors * self.minibatch_size_per_env)
        self.num_minibatches = self.horizon_length * self.num_actors // self.minibatch_size
        self.clip_value = config['clip_value']
        self.writter = writter
        self.weight_decay = config.get('weight_decay', 0.0)
        self.optimizer = FlatAdam(self.model.parameters(), self.lr, eps=1e-08, weight_decay=self.weight_decay)
        self._engine = None
        if config.get('fused_mlp', True) and value_size == 1:
            try:
                self._engine = _value_chain(self.model.a2c_network, self.optimizer, self.minibatch_size)
            except NotImplementedError as e:
                print(f'rl_games_amd: central value network outside the fused chain kernels ({e}); using autograd')
        self.frame = 0
        self.epoch_num = 0
        self.grad_norm = config.get('grad_norm', 1)
        self.truncate_grads = config.get('truncate_grads', False)
        self.e_clip = config.get('e_clip', 0.2)
        self.batch_size = self.horizon_length * self.num_actors
        self.local_rank = self.global_rank = 0
        self.world_size = 1
        if self.multi_gpu:
            self.local_rank, self.global_rank, self.world_size = rdist.env_ranks()
        if self.is_rnn:                                            # central_value.py:96-107
            self.rnn_states = [s.to(ppo_device) for s in self.model.get_default_rnn_state()]
            num_seqs = self.horizon_length // self.seq_length
            assert (self.horizon_length * self.num_actors // self.num_minibatches) % self.seq_length == 0
            self.mb_rnn_states = [torch.zeros((num_seqs, s.size()[0], self.num_actors, s.size()[2]), dtype=torch.float32,
                                              device=ppo_device) for s in self.rnn_states]
        self.dataset = PPODataset(self.batch_size, self.minibatch_size, True, self.is_rnn, ppo_device, self.seq_length)
        mb = self.minibatch_size
        self._d_val = torch.empty(mb, dtype=torch.float32, device=ppo_device)
        self._partials = torch.empty((mb + 255) // 256, 7, dtype=torch.float64, device=ppo_device)
        self._rows = torch.zeros(max(1, self.mini_epoch * self.num_minibatches), 8, dtype=torch.float32,
                                 device=ppo_device)
        self._no_logstd = torch.zeros(1, dtype=torch.float32, device=ppo_device)
        self._row_index = 0

    # ------------------------------------------------------------------ reference API
    def update_lr(self, lr):
        self.optimizer.set_lr(float(lr))           # identical on every rank by construction

    def get_stats_weights(self, model_stats=False):
        state = {}
        if model_stats:
            if self.normalize_input:
                state['running_mean_std'] = self.model.running_mean_std.state_dict()
            if self.normalize_value:
                state['reward_mean_std'] = self.model.value_mean_std.state_dict()
        return state

    def set_stats_weights(self, weights):
        pass

    def update_dataset(self, batch_dict):
        if self.num_agents > 1:                                    # central_value.py:153-158
            res = self.update_multiagent_tensors(batch_dict['old_values'], batch_dict['returns'], batch_dict['actions'],
                                                 batch_dict['dones'])
            batch_dict['old_values'], batch_dict['returns'], batch_dict['actions'], batch_dict['dones'] = res
        if self.is_rnn:                                            # central_value.py:163-170
            states = []
            for mb_s in self.mb_rnn_states:
                t_size = mb_s.size()[0] * mb_s.size()[2]
                states.append(mb_s.permute(1, 2, 0, 3).reshape(-1, t_size, mb_s.size()[3]))
            batch_dict['rnn_states'] = states
        self.dataset.update_values_dict(batch_dict)

    def _preproc_obs(self, obs_batch):
        if obs_batch.dtype == torch.uint8:
            obs_batch = obs_batch.float() / 255.0
        return obs_batch

    def pre_step_rnn(self, n):
        """central_value.py:189-194: the state each sequence of the rollout starts from."""
        if self.is_rnn and n % self.seq_length == 0:
            for s, mb_s in zip(self.rnn_states, self.mb_rnn_states):
                mb_s[n // self.seq_length, :, :, :] = s

    def post_step_rnn(self, all_done_indices, zero_rnn_on_done=True):
        """central_value.py:196-203 (indices of finished rows)."""
        if not self.is_rnn or not self.zero_rnn_on_done:
            return
        idx = all_done_indices[::self.num_agents] // self.num_agents
        for s in self.rnn_states:
            s[:, idx, :] = 0

    def zero_states_where(self, done_mask):
        """post_step_rnn for a device-side mask [num_actors * num_agents] instead of an index list (nothing waits for the
        host).  Multi-agent: post_step_rnn looks at every num_agents-th finished row, whichever agents those are; this
        form asks for agent 0 of an env - the same thing wherever the agents of an env finish together."""
        if self.is_rnn and self.zero_rnn_on_done:
            if self.num_agents > 1:
                done_mask = done_mask[::self.num_agents].contiguous()
            for s in self.rnn_states:
                ops.rnn_zero_done_states(s, done_mask)

    def update_multiagent_tensors(self, value_preds, returns, actions, dones):
        """central_value.py:225-234: the critic trains on one row per env and step - agent 0's values and returns in
        env-major order (the order of the flattened states); `dones` are cut, not re-ordered, exactly as there."""
        batch_size = self.batch_size
        ma_batch_size = self.num_actors * self.num_agents * self.horizon_length
        shape = (self.num_actors, self.num_agents, self.horizon_length, self.value_size)
        value_preds = value_preds.reshape(shape).transpose(0, 1).contiguous().view(ma_batch_size, self.value_size)[:batch_size]
        returns = returns.reshape(shape).transpose(0, 1).contiguous().view(ma_batch_size, self.value_size)[:batch_size]
        dones = dones.contiguous().view(ma_batch_size, self.value_size)[:batch_size]
        return value_preds, returns, actions, dones

    def forward(self, input_dict):
        return self.model(input_dict)

    def get_value(self, input_dict):
        self.eval()
        obs_batch = self._preproc_obs(input_dict['states'])
        with torch.no_grad():
            res = self.forward({'obs': obs_batch, 'actions': input_dict.get('actions', None),
                                'rnn_states': self.rnn_states, 'is_train': False})
        if self.is_rnn:                                            # central_value.py:222
            self.rnn_states = [s.contiguous() for s in res['rnn_states']]
        value = res['values']
        if self.num_agents > 1:                                    # every agent of an env gets the env's value (:223-225)
            value = value.repeat(1, self.num_agents)
            value = value.view(value.size()[0] * self.num_agents, -1)
        return value

    def train_critic(self, input_dict):
        self.train()
        return self.calc_gradients(input_dict)

    def train_net(self):
        """central_value.py:236-260.  Returns the mean loss as a 0-dim device tensor (one host read
        per epoch instead of the reference's `.item()` per minibatch)."""
        self.train()
        self._row_index = 0
        count = 0
        for _ in range(self.mini_epoch):
            if self.config.get('freeze_critic', False):
                break
            for i in range(len(self.dataset)):
                self.train_critic(self.dataset[i])
                count += 1
            if self.normalize_input:
                self.model.running_mean_std.eval()
        total = self._rows[:count, 5].sum() if count else torch.zeros((), device=self.ppo_device)
        avg_loss = total / (self.mini_epoch * self.num_minibatches)
        self.epoch_num += 1
        self.lr, _ = self.scheduler.update(self.lr, 0, self.epoch_num, self.frame, 0)
        self.update_lr(self.lr)
        self.frame += self.batch_size
        if self.writter is not None:
            self.writter.add_scalar('losses/cval_loss', avg_loss.item(), self.frame)
            self.writter.add_scalar('info/cval_lr', self.lr, self.frame)
        return avg_loss

    def calc_gradients(self, batch):
        """central_value.py:278-335: forward, fused value loss + gradient, backward, (all-reduce,)
        clip, Adam.  Returns the minibatch loss (0-dim device tensor)."""
        opt = self.optimizer
        obs_batch = self._preproc_obs(batch['obs'])
        rnn_masks = batch.get('rnn_masks')
        opt.zero_grad()
        eng = self._engine
        if eng is not None:
            # models.py:54-56 (norm_obs: training mode updates the statistics first), then the whole network in one launch
            rms, eps = None, 1e-5
            if self.normalize_input:
                m = self.model.running_mean_std
                if m.training:
                    m.update(obs_batch)
                rms, eps = (m.running_mean, m.running_var), m.epsilon
            values = eng.forward(obs_batch, rms, eps)                       # [mb, 1]
        else:
            rnn = None
            if self.is_rnn:                                                 # central_value.py:300-307
                rnn = {'rnn_states': batch['rnn_states'], 'seq_length': self.seq_length, 'dones': batch['dones']}
            values = self.model.forward_values(obs_batch, rnn)              # [mb, V], autograd graph
        mb = values.shape[0]
        mask = mask_sum = None
        if rnn_masks is not None:
            mask = rnn_masks.reshape(-1).float().contiguous()
            mask_sum = mask.sum().reshape(1)
        row = self._rows[self._row_index % self._rows.shape[0]]
        self._row_index += 1
        d_val = self._d_val[:mb] if eng is None else eng.d_heads[:mb].view(-1)
        with torch.no_grad():
            ops.value_loss(values.detach().reshape(-1), batch['old_values'].reshape(-1).contiguous(),
                           batch['returns'].reshape(-1).contiguous(), d_val, self._partials, self.e_clip,
                           self.clip_value, mask, mask_sum)
            ops.ppo_loss_finalize(self._partials, (mb + 255) // 256, 0, mb, mask is not None, 2.0, 0.0, 0.0,
                                  row, self._no_logstd)
        if eng is not None:
            eng.backward()
        else:
            torch.autograd.backward([values], [d_val.view(mb, 1)])
        if self.multi_gpu:
            rdist.all_reduce_sum(opt.flat_grads)
        scale = 1.0 / self.world_size if self.multi_gpu else 1.0
        opt.step(grad_scale=scale, max_norm=self.grad_norm if self.truncate_grads else None)
        return row[5]
