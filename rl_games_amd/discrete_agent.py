"""Discrete-action PPO agent on MI355X - host mirror of `DiscreteA2CAgent`
(rl_games/algos_torch/a2c_discrete.py:13-209 over `DiscreteA2CBase`,
rl_games/common/a2c_common.py:1204-1359), BASELINE.json config #1 (CartPole-shaped).

Shares the rollout / GAE / dataset-preparation / optimiser path with `A2CAgent`; what differs:
  * `Discrete` or `Tuple`-of-`Discrete` (multi-discrete) action spaces, int64 actions, update_list
    without mus/sigmas, optional action masks from `vec_env.get_action_masks()` with
    CategoricalMasked semantics (a2c_common.py:995-997,1224-1229; a2c_discrete.py:92-114);
  * the minibatch loss is the categorical one, a single fused HIP kernel
    (csrc/ppo_loss.hip `ppo_loss_discrete_kernel`) that emits d loss/d logits and d loss/d value;
    the MLP backward runs through autograd (`torch.autograd.backward` on the two heads);
  * the lr schedule is stepped once per mini-epoch on the mean KL (a2c_common.py:1271-1278),
    whatever `schedule_type` says, and `train_epoch` returns the 10-tuple without bound losses.
"""
import torch

from . import ops
from .agent import A2CAgent


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
        logits, values = self.model.forward_heads({'is_train': True, 'obs': obs_batch})
        mb, n = logits.shape
        mask = mask_sum = None
        if rnn_masks is not None:
            mask = rnn_masks.reshape(-1).float().contiguous()
            mask_sum = mask.sum().reshape(1)
        d_logits, d_val = self._d_logits[:mb], self._d_val[:mb]
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
        torch.autograd.backward([logits, values], [d_logits, d_val.view(mb, 1)])

    def train_epoch(self):
        """a2c_common.py:1232-1289: (step_time, play_time, update_time, total_time, a_losses,
        c_losses, entropies, kls, last_lr, lr_mul)."""
        res = super().train_epoch()
        return res[:6] + res[7:]
