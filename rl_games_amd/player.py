"""Inference-side players for checkpoints written by `A2CAgent` / `DiscreteA2CAgent` or by the
reference agents (same `.pth` format): host mirror of `PpoPlayerContinuous` / `PpoPlayerDiscrete`
(rl_games/algos_torch/players.py:17-160) over the `BasePlayer` essentials
(rl_games/common/player.py:17-102, :242-330): `restore(fn)`, `get_action(obs, is_deterministic)`,
`reset()`, `run()`.  The policy is the same module the agents train (`policy.PolicyBuilder`), on
the GPU; observation normalisation runs through the `RunningMeanStd` HIP kernels in eval mode.
Not mirrored: the evaluation-worker checkpoint watcher, rendering, self-play hooks.
"""
import numpy as np
import torch

from .agent import rescale_actions
from .policy import PolicyBuilder


class _BasePlayer:
    def __init__(self, params):
        self.config = config = params['config']
        self.network = config.get('network')
        if self.network is None or not hasattr(self.network, 'build'):
            self.network = config['network'] = PolicyBuilder(params)
        self.player_config = config.get('player', {})
        self.env_info = config.get('env_info')
        self.env = config.get('vec_env')
        if self.env_info is None:
            if self.env is None:
                from .synthetic_env import SyntheticTensorEnv
                self.env = SyntheticTensorEnv(config['num_actors'], device=config.get('device', 'cuda:0'),
                                              **config.get('env_config', {}))
            self.env_info = self.env.get_env_info()
        self.clip_actions = config.get('clip_actions', True)
        self.num_agents = self.env_info.get('agents', 1)
        self.value_size = self.env_info.get('value_size', 1)
        self.action_space = self.env_info['action_space']
        self.observation_space = self.env_info['observation_space']
        if type(self.observation_space).__name__ == 'Dict':
            raise NotImplementedError('dict observations are not implemented on the MI355X path')
        self.obs_shape = self.observation_space.shape
        self.device = torch.device(config.get('device', config.get('device_name', 'cuda:0')))
        if self.device.type != 'cuda':
            raise RuntimeError('rl_games_amd players run on an MI355X HIP device only')
        self.states = None
        self.batch_size = 1
        self.has_batch_dimension = False
        self.is_tensor_obses = False
        self.games_num = self.player_config.get('games_num', 1000000000)
        self.is_deterministic = self.player_config.get('deterministic', True)
        self.print_stats = self.player_config.get('print_stats', True)
        self.max_steps = 108000 // 4
        self.normalize_input = config['normalize_input']
        self.normalize_value = config.get('normalize_value', False)

    def _build_model(self, actions_num):
        self.model = self.network.build({
            'actions_num': actions_num, 'input_shape': self.obs_shape, 'num_seqs': self.num_agents,
            'value_size': self.value_size, 'normalize_value': self.normalize_value,
            'normalize_input': self.normalize_input}).to(self.device)
        self.model.eval()
        self.is_rnn = self.model.is_rnn()

    # ------------------------------------------------------------------ reference API
    def restore(self, fn):
        checkpoint = torch.load(fn, map_location=self.device, weights_only=False)
        state = {k.replace('_orig_mod.', ''): v for k, v in checkpoint['model'].items()}
        self.model.load_state_dict(state)
        if self.normalize_input and 'running_mean_std' in checkpoint:
            self.model.running_mean_std.load_state_dict(checkpoint['running_mean_std'])
        env_state = checkpoint.get('env_state', None)
        if self.env is not None and env_state is not None:
            self.env.set_env_state(env_state)

    def get_weights(self):
        return {'model': self.model.state_dict()}

    def set_weights(self, weights):
        self.model.load_state_dict(weights['model'])
        if self.normalize_input and 'running_mean_std' in weights:
            self.model.running_mean_std.load_state_dict(weights['running_mean_std'])

    def reset(self):
        self.init_rnn()

    def init_rnn(self):
        if self.is_rnn:
            self.states = [torch.zeros((s.size()[0], self.batch_size, s.size()[2]), dtype=torch.float32,
                                       device=self.device) for s in self.model.get_default_rnn_state()]

    def _to_device(self, obs):
        if isinstance(obs, dict):
            obs = obs['obs']
        if isinstance(obs, np.ndarray):
            obs = torch.from_numpy(obs)
        else:
            self.is_tensor_obses = True
        obs = obs.to(self.device)
        if obs.dtype == torch.uint8:
            obs = obs.float() / 255.0
        elif obs.dtype != torch.float32:
            obs = obs.float()
        return obs

    def _forward(self, obs, action_masks=None):
        obs = self._to_device(obs)
        if not self.has_batch_dimension and obs.dim() == len(self.obs_shape):
            obs = obs.unsqueeze(0)
        inputs = {'is_train': False, 'prev_actions': None, 'obs': obs.contiguous(), 'rnn_states': self.states}
        if action_masks is not None:
            inputs['action_masks'] = action_masks
        with torch.no_grad():
            res = self.model(inputs)
        self.states = res['rnn_states']
        return res

    def run(self):
        """Plays `games_num` episodes (counted over all envs of the vec-env) and returns
        (mean reward, mean episode length)."""
        obs = self.env.reset()
        first = self._to_device(obs)
        self.batch_size = first.shape[0] if first.dim() > len(self.obs_shape) else 1
        self.has_batch_dimension = first.dim() > len(self.obs_shape)
        self.init_rnn()
        cur_r = torch.zeros(self.batch_size, device=self.device)
        cur_n = torch.zeros(self.batch_size, device=self.device)
        sum_r = sum_n = 0.0
        games = 0
        mask_fn = getattr(self.env, 'has_action_mask', None)       # the reference's player-side name (player.py:284-294)
        has_masks = bool(mask_fn()) if mask_fn is not None else False
        for _ in range(self.max_steps):
            if has_masks:
                action = self.get_masked_action(obs, self.env.get_action_mask(), self.is_deterministic)
            else:
                action = self.get_action(obs, self.is_deterministic)
            if not self.is_tensor_obses:
                action = action.cpu().numpy()
            obs, rewards, dones, _ = self.env.step(action)
            rewards = torch.as_tensor(rewards, device=self.device, dtype=torch.float32).reshape(self.batch_size, -1)[:, 0]
            dones = torch.as_tensor(dones, device=self.device).reshape(-1).bool()
            cur_r += rewards
            cur_n += 1
            if self.is_rnn and dones.any():
                for s in self.states:
                    s[:, dones, :] = 0.0
            finished = int(dones.sum().item())
            if finished:
                sum_r += float(cur_r[dones].sum().item())
                sum_n += float(cur_n[dones].sum().item())
                games += finished
                cur_r[dones] = 0
                cur_n[dones] = 0
                if games >= self.games_num:
                    break
        if self.print_stats and games:
            print(f'av reward: {sum_r / games:.4f} av steps: {sum_n / games:.2f} games: {games}')
        return (sum_r / max(games, 1), sum_n / max(games, 1))


class PpoPlayerContinuous(_BasePlayer):
    def __init__(self, params):
        super().__init__(params)
        self.actions_num = self.action_space.shape[0]
        self.actions_low = torch.from_numpy(np.asarray(self.action_space.low).copy()).float().to(self.device)
        self.actions_high = torch.from_numpy(np.asarray(self.action_space.high).copy()).float().to(self.device)
        self._build_model(self.actions_num)

    def get_action(self, obs, is_deterministic=False):
        res = self._forward(obs)
        current = res['mus'] if is_deterministic else res['actions']
        if not self.has_batch_dimension:
            current = torch.squeeze(current.detach())
        if self.clip_actions:
            return rescale_actions(self.actions_low, self.actions_high, torch.clamp(current, -1.0, 1.0))
        return current


class PpoPlayerDiscrete(_BasePlayer):
    """`Discrete` and `Tuple`-of-`Discrete` action spaces (players.py:85-181).  Deterministic play takes the arg-max
    of every head (multi-discrete: stacked on the last axis), otherwise the model's own sample."""

    def __init__(self, params):
        super().__init__(params)
        kind = type(self.action_space).__name__
        if kind == 'Discrete':
            self.actions_num = self.action_space.n
            self.is_multi_discrete = False
        elif kind == 'Tuple':
            self.actions_num = [a.n for a in self.action_space]
            self.is_multi_discrete = True
        else:
            raise NotImplementedError(f'{kind} action spaces are not played by PpoPlayerDiscrete')
        self.mask = [False]
        self._build_model(self.actions_num)

    def _pick(self, res, is_deterministic):
        if not is_deterministic:
            action = res['actions']
        elif self.is_multi_discrete:
            action = torch.stack([torch.argmax(l, dim=-1) for l in res['logits']], dim=-1)
        else:
            action = torch.argmax(res['logits'], dim=-1)
        return action if self.has_batch_dimension else torch.squeeze(action.detach())

    def get_action(self, obs, is_deterministic=True):
        return self._pick(self._forward(obs), is_deterministic)

    def get_masked_action(self, obs, action_masks, is_deterministic=True):
        """action_masks: bool [rows, sum of the head sizes] (numpy or tensor), True = allowed (players.py:117-148)."""
        masks = torch.as_tensor(action_masks).to(self.device).bool()
        if not self.has_batch_dimension and masks.dim() == 1:
            masks = masks.unsqueeze(0)
        return self._pick(self._forward(obs, action_masks=masks), is_deterministic)
