"""Device-resident synthetic vector environment (the `IVecEnv` data-source seam,
rl_games/common/ivecenv.py:1-36) producing the shapes of BASELINE.json's configs.

SURVEY 8(d): obs = 3*randn(N,O)+1, rewards = randn(N), dones = rand(N) < 0.05,
time_outs = dones & (rand(N) < 0.5); `autoreset_mode: 'same_step'` (Isaac-style), torch-tensor
observations (zero-copy mode of the agent, a2c_common.py:672-673,:712-715).  Works on any
torch device so that the CPU baseline drives the very same environment."""
import numpy as np
import torch

from .spaces import Box, Discrete, Tuple


class SyntheticTensorEnv:
    def __init__(self, num_envs, obs_dim, act_dim=0, device='cuda:0', seed=1234, p_done=0.05,
                 value_size=1, discrete_actions=None, autoreset_mode='same_step', state_dim=0, agents=1,
                 action_masks=False):
        self.num_envs, self.obs_dim, self.act_dim = num_envs, obs_dim, act_dim
        self.autoreset_mode = autoreset_mode
        self.agents = agents            # agents per env: every per-step tensor has num_envs * agents rows
        self.state_dim = state_dim      # > 0: privileged `states` for a central value function
        self.device = torch.device(device)
        self.p_done = p_done
        self.value_size = value_size
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed)
        self.observation_space = Box(-np.inf, np.inf, (obs_dim,), np.float32)
        self.action_masks = bool(action_masks)    # discrete only: random masks, >= 1 allowed action per head
        if isinstance(discrete_actions, (list, tuple)):
            self.action_space = Tuple([Discrete(n) for n in discrete_actions])      # multi-discrete
            self.head_sizes = [int(n) for n in discrete_actions]
        elif discrete_actions is not None:        # CartPole-like (BASELINE config #1)
            self.action_space = Discrete(discrete_actions)
            self.head_sizes = [int(discrete_actions)]
        else:
            self.action_space = Box(-1.0, 1.0, (act_dim,), np.float32)

    def _obs(self):
        if self.device.type == 'cuda':
            # one generation pass (N(1, 3^2) written once) instead of randn + mul + add: 28 MB instead
            # of 140 MB of traffic per step at 65,536 x 108 - the env should not evict the rollout's own
            # rewards / values / done flags from the Infinity Cache before the GAE launch reads them
            obs = torch.empty(self.num_envs * self.agents, self.obs_dim, device=self.device).normal_(
                1.0, 3.0, generator=self.gen)
        else:
            obs = torch.randn(self.num_envs * self.agents, self.obs_dim, device=self.device, generator=self.gen) * 3.0 + 1.0
        if self.state_dim > 0:
            states = torch.randn(self.num_envs, self.state_dim, device=self.device, generator=self.gen) * 2.0 - 0.5
            return {'obs': obs, 'states': states}
        return obs

    def reset(self):
        return self._obs()

    def step(self, actions):
        n = self.num_envs * self.agents
        obs = self._obs()
        if self.value_size == 1:
            rewards = torch.randn(n, device=self.device, generator=self.gen)
        else:
            rewards = torch.randn(n, self.value_size, device=self.device, generator=self.gen)
        u = torch.rand(2, n, device=self.device, generator=self.gen)
        dones = u[0] < self.p_done
        time_outs = dones & (u[1] < 0.5)
        return obs, rewards, dones.to(torch.uint8), {'time_outs': time_outs}

    def get_env_info(self):
        info = {'observation_space': self.observation_space, 'action_space': self.action_space,
                'agents': self.agents, 'value_size': self.value_size, 'autoreset_mode': self.autoreset_mode}
        if self.state_dim > 0:
            info['state_space'] = Box(-np.inf, np.inf, (self.state_dim,), np.float32)
        return info

    def has_action_masks(self):
        return self.action_masks

    def get_action_masks(self):
        """bool [rows, sum(head sizes)] - numpy on the CPU (the reference wraps it in torch.BoolTensor,
        a2c_discrete.py:94), a device tensor otherwise."""
        rows = self.num_envs * self.agents
        total = sum(self.head_sizes)
        masks = torch.rand(rows, total, device=self.device, generator=self.gen) > 0.4
        pick = torch.rand(rows, len(self.head_sizes), device=self.device, generator=self.gen)
        at = 0
        for b, n in enumerate(self.head_sizes):
            forced = at + (pick[:, b] * n).long().clamp(max=n - 1)
            masks[torch.arange(rows, device=self.device), forced] = True
            at += n
        return masks.cpu().numpy() if self.device.type == 'cpu' else masks

    # the names the reference's players ask for (rl_games/common/player.py:284-329); agents use the plural ones
    has_action_mask = has_action_masks
    get_action_mask = get_action_masks

    def get_number_of_agents(self):
        return self.agents

    def set_train_info(self, env_frames, *args, **kwargs):
        pass

    def get_env_state(self):
        return None

    def set_env_state(self, env_state):
        pass

    def seed(self, seed):
        self.gen.manual_seed(seed)
