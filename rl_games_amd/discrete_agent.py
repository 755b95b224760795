"""Discrete-action PPO agent on MI355X - host mirror of `DiscreteA2CAgent`
(rl_games/algos_torch/a2c_discrete.py:13-209 over `DiscreteA2CBase`,
rl_games/common/a2c_common.py:1204-1359), BASELINE.json config #1 (CartPole-shaped).

Shares the rollout / GAE / dataset-preparation / optimiser path with `A2CAgent`; what differs:
  * `Discrete` or `Tuple`-of-`Discrete` (multi-discrete) action spaces, int64 actions, update_list
    without mus/sigmas, optional action masks from `vec_env.get_action_masks()` with
    CategoricalMasked semantics (a2c_common.py:995-997,1224-1229; a2c_discrete.py:92-114);
  * the minibatch loss is the categorical one, a single fused HIP kernel
    (csrc/ppo_loss.hip `ppo_loss_discrete_kernel`) that emits d loss/d logits and d loss/d value;
  * the network around it runs on the fused chain kernels of the continuous hot path (chain_net.ChainNet: observation
    normaliser + trunk + heads as one forward launch, one backward launch, MFMA weight gradients) - one chain over
    [value | logits] behind a shared trunk, one chain per trunk with `separate: True` (ppo_cartpole.yaml:17) - where the
    network has that form (plain Linear + ELU / ReLU / tanh trunks, widths that are multiples of 4, value_size 1);
    anything else, or `fused_mlp: False`, keeps autograd around the loss kernel (`torch.autograd.backward` on the heads);
  * the lr schedule is stepped once per mini-epoch on the mean KL (a2c_common.py:1271-1278),
    whatever `schedule_type` says, and `train_epoch` returns the 10-tuple without bound losses.
"""
import torch

from torch import nn

from . import ops
from .agent import A2CAgent
from .chain_net import ChainNet, arena_layout


class DiscreteA2CAgent(A2CAgent):
    def _init_action_space(self, config):
        """DiscreteA2CBase.__init__ (a2c_common.py:1206-1222)."""
        action_space = self.env_info['action_space']
        kind = type(action_space).__name__
        rows = self.num_agents * self.num_actors
        if kind == 'Discrete':
            self.actions_shape = (self.horizon_length, rows)
            self.actions_num = int(action_space.n)
            self.branch_sizes = [self.actions_num]
            self.is_multi_discrete = False
        elif kind == 'Tuple':
            self.actions_shape = (self.horizon_length, rows, len(action_space))
            self.actions_num = [int(a.n) for a in action_space]
            self.branch_sizes = list(self.actions_num)
            self.is_multi_discrete = True
        else:
            raise ValueError(f'Unsupported action space type for DiscreteA2CBase: {type(action_space)}')
        self.is_discrete = True
        self.bounds_loss_coef = None
        self.clip_actions = False
        # the reference's discrete train_epoch has no per-minibatch scheduling: it always updates
        # the lr once per mini-epoch with the mean KL - the base class's 'standard' schedule
        self.schedule_type = 'standard'

    def _supports_action_masks(self):
        return True

    def _chain_heads(self):
        """Head groups of the network, one per chain: [[value, logits...]] behind a shared trunk, [[logits...], [value]]
        for separate actor / critic trunks."""
        net = self.model.a2c_network
        logits = list(net.logits) if net.is_multi_discrete else [net.logits]
        return [logits, [net.value]] if net.is_separate_critic() else [[net.value] + logits]

    def _arena_layout(self, config):
        if not config.get('fused_mlp', True):
            return None
        net = self.model.a2c_network
        rest = [p for p in self.model.parameters() if all(p is not q for q in net.parameters())]
        return arena_layout(list(net.parameters()), self._chain_heads()) + rest

    def _init_chains(self, config):
        self._chains = None
        net = self.model.a2c_network
        if not config.get('fused_mlp', True):
            return
        try:
            if self.value_size != 1 or not isinstance(net.value_act, nn.Identity):
                raise NotImplementedError('one linear value column only')
            if not getattr(net, 'plain_trunk', True) or net.is_rnn():
                raise NotImplementedError('plain Linear + activation trunks only')
            rows = self.minibatch_size
            groups = self._chain_heads()
            trunks = [net.actor_mlp, net.critic_mlp] if net.is_separate_critic() else [net.actor_mlp]
            self._chains = [ChainNet(t, g, self.optimizer, rows) for t, g in zip(trunks, groups)]
        except NotImplementedError as e:
            print(f'rl_games_amd: discrete network outside the fused chain kernels ({e}); using autograd')
            self._chains = None

    def _alloc_loss_scratch(self, mb, dev):
        self._d_logits = torch.empty(mb, sum(self.branch_sizes), dtype=torch.float32, device=dev)
        self._d_val = torch.empty(mb, dtype=torch.float32, device=dev)
        self._loss_blocks = ops.ppo_loss_discrete_blocks(mb)
        self._loss_partials = torch.empty(self._loss_blocks, ops.ppo_loss_partials_per_block(0),
                                          dtype=torch.float64, device=dev)
        self._no_logstd = torch.zeros(1, dtype=torch.float32, device=dev)

    def _rollout_fields(self):
        fields = ['actions', 'neglogpacs', 'values']               # a2c_common.py:1224-1229
        return fields + ['action_masks'] if self.use_action_masks else fields

    def get_masked_action_values(self, obs, action_masks):
        """a2c_discrete.py:92-114.  action_masks: bool [rows, sum(head sizes)] (numpy or tensor)."""
        processed_obs = self._preproc_obs(obs['obs'])
        action_masks = torch.as_tensor(action_masks, dtype=torch.bool, device=self.ppo_device)
        self.model.eval()
        with torch.no_grad():
            res_dict = self.model({'is_train': False, 'prev_actions': None, 'obs': processed_obs,
                                   'action_masks': action_masks, 'rnn_states': self.rnn_states})
            if self.has_central_value:
                res_dict['values'] = self.get_central_value({'is_train': False, 'states': obs['states']})
        res_dict['action_masks'] = action_masks
        return res_dict

    def preprocess_actions(self, actions):
        """a2c_common.py:736-739 - discrete actions go to the env as they are."""
        if not self.is_tensor_obses:
            actions = actions.cpu().numpy()
        return actions

    def calc_gradients(self, input_dict):
        """a2c_discrete.py:121-209."""
        row = self._mb_scalars[self._mb_index % self._mb_scalars.shape[0]]
        self._mb_index += 1
        self._forward_loss_backward(input_dict, row)
        self.trancate_gradients_and_step()
        self.train_result = (row[0], row[1], row[2], row[4], self._host_lr, 1.0)

    def _forward_loss_backward(self, input_dict, row):
        opt = self.optimizer
        obs_batch = self._preproc_obs(input_dict['obs'])
        rnn_masks = input_dict.get('rnn_masks', None)
        opt.zero_grad()
        chains = self._chains if obs_batch.dtype == torch.float32 else None
        if chains is not None:
            # models.py:54-56 (norm_obs: training mode updates the statistics first), then each trunk + its heads as
            # one launch that normalises on the way in
            rms, eps = None, 1e-5
            if self.normalize_input:
                m = self.model.running_mean_std
                if m.training:
                    m.update(obs_batch)
                rms, eps = (m.running_mean, m.running_var), m.epsilon
            outs = [c.forward(obs_batch, rms, eps) for c in chains]
            mb = obs_batch.shape[0]
            if len(chains) == 1:                                   # [value | logits]
                logits, values = outs[0][:, 1:], outs[0][:, 0]
                d_logits, d_val = chains[0].d_heads[:mb, 1:], chains[0].d_heads[:mb, 0]
            else:
                logits, values = outs[0], outs[1][:, 0]
                d_logits, d_val = chains[0].d_heads[:mb], chains[1].d_heads[:mb, 0]
        else:
            batch = {'is_train': True, 'obs': obs_batch}
            if self.is_rnn:                                        # a2c_discrete.py:138-144
                batch.update(rnn_states=input_dict['rnn_states'], seq_length=self.seq_length)
                if self.zero_rnn_on_done:
                    batch['dones'] = input_dict['dones']
            logits, values = self.model.forward_heads(batch)
            mb = logits.shape[0]
            d_logits, d_val = self._d_logits[:mb], self._d_val[:mb]
        mask = mask_sum = None
        if rnn_masks is not None:
            mask = rnn_masks.reshape(-1).float().contiguous()
            mask_sum = mask.sum().reshape(1)
        with torch.no_grad():
            lg = logits.detach()
            ops.ppo_loss_discrete(lg if lg.stride(1) == 1 else lg.contiguous(), values.detach().reshape(-1),
                                  input_dict['actions'], input_dict['old_logp_actions'],
                                  input_dict['advantages'], input_dict['old_values'].reshape(-1),
                                  input_dict['returns'].reshape(-1), d_logits, d_val, self._loss_partials,
                                  self.e_clip, self.critic_coef if self.has_value_loss else 0.0,
                                  self.entropy_coef, self.clip_value, self.surrogate, mask, mask_sum,
                                  branch_sizes=self.branch_sizes, action_masks=input_dict.get('action_masks'))
            ops.ppo_loss_finalize(self._loss_partials, ops.ppo_loss_discrete_blocks(mb), 0, mb,
                                  mask is not None, self.critic_coef if self.has_value_loss else 0.0,
                                  self.entropy_coef, 0.0, row,
                                  self._no_logstd, opt.kl_slot)
        if chains is not None:
            for c in chains:
                c.backward()
        else:
            torch.autograd.backward([logits, values], [d_logits, d_val.view(mb, 1)])

    def train_epoch(self):
        """a2c_common.py:1232-1289: (step_time, play_time, update_time, total_time, a_losses,
        c_losses, entropies, kls, last_lr, lr_mul)."""
        res = super().train_epoch()
        return res[:6] + res[7:]
