"""A2CAgent: the rl_games continuous-PPO agent with the per-epoch hot path on MI355X kernels.

Drop-in for `rl_games.algos_torch.a2c_continuous.A2CAgent` (rl_games/algos_torch/
a2c_continuous.py:12-257) and its bases `ContinuousA2CBase` / `A2CBase`
(rl_games/common/a2c_common.py:168-1202, :1482-1782) behind the same plugin seam:

    runner.algo_factory.register_builder('a2c_continuous', lambda **kw: A2CAgent(**kw))
    agent = A2CAgent(base_name='run', params=params);  agent.train()

Same constructor (`base_name`, `params` dict with the reference's config keys), same public
methods and attributes: init_tensors, env_reset, play_steps, play_steps_rnn,
prepare_dataset, dataset[i], train_actor_critic, calc_gradients, train_epoch, train,
get_action_values, get_values, update_lr, save/restore, get/set_weights, get/set_param,
experience_buffer.tensor_dict, game_rewards/game_lengths, value_mean_std, ...

What runs where (per epoch):
  rollout   : model forward on rocBLAS/hipBLASLt via torch; every buffer write of a step is ONE
              launch (csrc/experience.hip); reward shaping, time-out bootstrap, episode
              accumulators and meters are one launch per step + one per rollout, with no host
              sync (the reference syncs twice per step: a2c_common.py:1039, algo_observer.py:42)
  GAE       : one launch -> returns, advantages, fp64 moments (csrc/gae.hip)
  dataset   : 2 launches (statistics, normalise); swap_and_flatten01 is a view
  minibatch : obs normaliser 3 launches, fused loss fwd+bwd+KL 2 launches, clip+Adam+adaptive-lr
              2 launches, one all-reduce; learning rate and KL stay on the device.
"""
import gc
import os
import time
from datetime import datetime

import numpy as np
import torch

from . import distributed as rdist
from . import ops
from .flat_optim import FlatAdam
from .gae import compute_gae, gae_returns_advantages
from .lr_control import AdaptiveScheduler, IdentityScheduler, LinearScheduler
from .minibatch import PPODataset
from .normalizers import GeneralizedMovingStats
from .policy import PolicyBuilder
from .rollout_buffer import ExperienceBuffer


def swap_and_flatten01(arr):
    """[H, N, ...] -> [N*H, ...], flat index env*H + t (a2c_common.py:33-40).  On the env-major
    buffer views this is a zero-copy reshape."""
    if arr is None:
        return arr
    s = arr.size()
    return arr.transpose(0, 1).reshape(s[0] * s[1], *s[2:])


def rescale_actions(low, high, action):
    d = (high - low) / 2.0
    m = (high + low) / 2.0
    return action * d + m


class GraphCaptureError(RuntimeError):
    """A HIP graph capture failed; nothing ran on the device and host mirrors were rolled back."""


class NullObserver:
    """AlgoObserver interface (rl_games/common/algo_observer.py:6-26) with no-op hooks."""

    def before_init(self, base_name, config, experiment_name):
        pass

    def after_init(self, algo):
        pass

    def process_infos(self, infos, done_indices):
        pass

    def after_steps(self):
        pass

    def after_clear_stats(self):
        pass

    def after_print_stats(self, frame, epoch_num, total_time):
        pass


class NullWriter:
    def add_scalar(self, *args, **kwargs):
        pass

    def flush(self):
        pass


def _make_writer(summaries_dir):
    try:
        from tensorboardX import SummaryWriter  # optional, like the reference (a2c_common.py:21)
        return SummaryWriter(summaries_dir)
    except Exception:
        return NullWriter()


class DeviceAverageMeter:
    """AverageMeter (rl_games/algos_torch/torch_ext.py:326-352) whose running mean and size
    live on the device and are updated by rlg_episode_meters_update; host reads are lazy."""

    def __init__(self, in_shape, max_size, device, sizes, slot):
        self.max_size = max_size
        self.mean = torch.zeros(in_shape, dtype=torch.float32, device=device)
        self._sizes = sizes
        self._slot = slot

    @property
    def current_size(self):
        return int(self._sizes[self._slot].item())

    def __len__(self):
        return self.current_size

    def clear(self):
        self._sizes[self._slot] = 0
        self.mean.fill_(0)

    def get_mean(self):
        return self.mean.squeeze(0).cpu().numpy()

    def update(self, values):
        size = values.size()[0]
        if size == 0:
            return
        new_mean = torch.mean(values.float(), dim=0)
        size = int(np.clip(size, 0, self.max_size))
        old_size = min(self.max_size - size, self.current_size)
        size_sum = old_size + size
        self._sizes[self._slot] = size_sum
        self.mean = (self.mean * old_size + new_mean * size) / size_sum


class _LeanChains:
    """The fp32 weight fragments of the lean 16-row kernels of one or two ops.MlpChain objects, handled together."""

    def __init__(self, chains):
        self.chains = chains

    def pack_frags(self, stream_of):
        for c in self.chains:
            c.pack_frags(stream_of)

    def mark_frags(self, version):
        for c in self.chains:
            c.mark_frags(version)

    def ensure_frags(self, stream_of):
        for c in self.chains:
            c.ensure_frags(stream_of)


class A2CAgent:
    def __init__(self, base_name, params):
        self.config = config = params['config']
        self.params = params
        self.name = base_name
        full_name = config.get('full_experiment_name', None)
        self.experiment_name = full_name or (config['name'] + datetime.now().strftime('_%d-%H-%M-%S'))
        features = config.get('features') or {}
        self.algo_observer = features.get('observer') or NullObserver()
        self.algo_observer.before_init(base_name, config, self.experiment_name)
        self.network = config['network'] = PolicyBuilder(params)          # load_networks :516-525
        self.central_value_config = config.get('central_value_config', None)

        # ---- ranks / device (a2c_common.py:193-220) ----
        self.multi_gpu = config.get('multi_gpu', False)
        self.multi_gpu_sync_stats = config.get('multi_gpu_sync_stats', True)
        self.multi_gpu_sync_stats_mode = rdist.resolve_stats_sync_mode(
            config.get('multi_gpu_sync_stats_mode', 'pooled'))
        # once per epoch: do all ranks hold the same parameter / Adam-moment bits?  'raise' (default) | 'rebroadcast'
        # (rank 0's optimiser state wins, a warning is printed) | False
        self.multi_gpu_param_check = config.get('multi_gpu_param_check', 'raise')
        if self.multi_gpu_param_check not in ('raise', 'rebroadcast', False, None):
            raise ValueError("multi_gpu_param_check must be 'raise', 'rebroadcast' or False")
        self.local_rank = self.global_rank = 0
        self.world_size = 1
        if self.multi_gpu:
            self.local_rank, self.global_rank, self.world_size = rdist.env_ranks()
            dev_index = rdist.local_device_index(self.local_rank)
            config['device'] = 'cuda:' + str(dev_index)
            torch.cuda.set_device(dev_index)
            rdist.init_process_group(True)
            if self.global_rank != 0:
                config['print_stats'] = False
        self.ppo_device = config.get('device', 'cuda:0')
        if config.get('gemm_tuning', True):
            from . import gemm_tuning
            gemm_tuning.enable(allow_tuning=bool(config.get('gemm_tuning_online', False)))
        if not str(self.ppo_device).startswith('cuda'):
            raise RuntimeError(f"rl_games_amd.A2CAgent runs on an MI355X HIP device only, got device "
                               f"'{self.ppo_device}' (there is no CPU fallback)")

        # ---- environment (a2c_common.py:228-241) ----
        self.env_config = config.get('env_config', {})
        self.num_actors = config['num_actors']
        self.env_name = config.get('env_name', 'synthetic')
        self.env_info = config.get('env_info')
        self.vec_env = config.get('vec_env', None)
        if self.env_info is None:
            if self.vec_env is None:
                from .synthetic_env import SyntheticTensorEnv
                self.vec_env = SyntheticTensorEnv(self.num_actors, device=self.ppo_device, **self.env_config)
            self.env_info = self.vec_env.get_env_info()
        self.value_size = self.env_info.get('value_size', 1)
        self.observation_space = self.env_info['observation_space']
        self.num_agents = self.env_info.get('agents', 1)
        self.weight_decay = config.get('weight_decay', 0.0)
        self.has_central_value = self.central_value_config is not None
        if self.has_central_value:                                   # a2c_common.py:254-266
            self.state_space = self.env_info.get('state_space', None)
            if self.state_space is None:
                self.state_space = self.observation_space
                self.env_info['state_space'] = self.state_space
            if type(self.state_space).__name__ == 'Dict':
                raise NotImplementedError('dict state spaces are not implemented on the MI355X hot path')
            self.state_shape = self.state_space.shape
        self.use_action_masks = bool(config.get('use_action_masks', False))
        if self.use_action_masks and not self._supports_action_masks():
            raise NotImplementedError('action masks are not implemented for continuous actions')
        self.is_train = config.get('is_train', True)

        self.save_freq = config.get('save_frequency', 0)
        self.save_best_after = config.get('save_best_after', 100)
        self.print_stats = config.get('print_stats', True)
        self.epochs_between_resets = config.get('epochs_between_resets', 0)
        self.rnn_states = None
        self.ppo = config.get('ppo', True)      # False: a_loss = neglogp * advantage (common_losses.py:59, 80) - surrogate kind 2
        self.max_epochs = config.get('max_epochs', -1)
        self.max_frames = max(config.get('max_frames', -1), config.get('max_steps', -1))
        self.stop_fn = config.get('stop_fn', None)
        if self.stop_fn is not None and not callable(self.stop_fn):
            raise ValueError(f"'stop_fn' must be callable, got {type(self.stop_fn).__name__}")

        # ---- lr schedule (a2c_common.py:286-332) ----
        lr_schedule = config.get('lr_schedule')
        self.is_adaptive_lr = lr_schedule == 'adaptive'
        self.linear_lr = lr_schedule == 'linear'
        self.schedule_type = config.get('schedule_type', 'per_minibatch')
        if self.schedule_type == 'legacy':
            self.schedule_type = 'per_minibatch'
        if self.is_adaptive_lr:
            self.kl_threshold = config['kl_threshold']
            self.scheduler = AdaptiveScheduler(self.kl_threshold, min_lr=config.get('min_lr', 1e-6),
                                               max_lr=config.get('max_lr', 1e-2),
                                               lr_multiplier=config.get('lr_multiplier', 1.5))
        elif self.linear_lr:
            if self.max_epochs == -1 and self.max_frames == -1:
                self.scheduler = IdentityScheduler()
            else:
                use_epochs = self.max_epochs != -1
                self.scheduler = LinearScheduler(
                    float(config['learning_rate']), min_lr=config.get('min_lr', 1e-6),
                    max_steps=self.max_epochs if use_epochs else self.max_frames, use_epochs=use_epochs,
                    apply_to_entropy=config.get('schedule_entropy', False),
                    start_entropy_coef=config.get('entropy_coef'))
        else:
            self.scheduler = IdentityScheduler()

        self.e_clip = config['e_clip']
        self.clip_value = config['clip_value']
        self.rewards_shaper = config.get('reward_shaper', {})
        self.autoreset_mode = (self.env_info or {}).get('autoreset_mode', 'same_step')
        self.mask_autoreset_rows = self.autoreset_mode == 'next_step'
        self._autoreset_prev_dones = None
        if self.mask_autoreset_rows and self.num_agents > 1:
            raise ValueError('PPO next_step autoreset masking does not support multi-agent envs; '
                             'wrap the env with a same_step autoreset adapter instead')
        self.horizon_length = config['horizon_length']
        self.seq_length = config.get('seq_length', 4)
        self.bptt_len = config.get('bptt_length', self.seq_length)
        self.zero_rnn_on_done = config.get('zero_rnn_on_done', True)
        self.normalize_advantage = config['normalize_advantage']
        self.normalize_rms_advantage = config.get('normalize_rms_advantage', False)
        self.normalize_input = config['normalize_input']
        self.normalize_value = config.get('normalize_value', False)
        self.truncate_grads = config.get('truncate_grads', False)
        if type(self.observation_space).__name__ == 'Dict':
            raise NotImplementedError('dict observations are not implemented on the MI355X hot path')
        self.obs_shape = self.observation_space.shape
        self.critic_coef = config['critic_coef']
        self.grad_norm = config['grad_norm']
        self.gamma = config['gamma']
        self.tau = config['tau']
        self.games_to_track = config.get('games_to_track', 100)

        self.batch_size = self.horizon_length * self.num_actors * self.num_agents
        self.batch_size_envs = self.horizon_length * self.num_actors
        if 'minibatch_size' not in config and 'minibatch_size_per_env' not in config:
            raise ValueError("Configuration must include either 'minibatch_size' or 'minibatch_size_per_env'. "
                             'Neither was found in the provided config.')
        self.minibatch_size_per_env = config.get('minibatch_size_per_env', 0)
        self.minibatch_size = config.get('minibatch_size', self.num_actors * self.minibatch_size_per_env)
        if self.minibatch_size <= 0:
            raise ValueError(f"'minibatch_size' must be greater than 0. Calculated value: {self.minibatch_size}.")
        self.games_num = self.minibatch_size // self.seq_length
        self.num_minibatches = self.batch_size // self.minibatch_size
        if self.batch_size % self.minibatch_size != 0:
            raise ValueError(f"'batch_size' ({self.batch_size}) must be divisible by 'minibatch_size' "
                             f'({self.minibatch_size}).')
        self.mini_epochs_num = config['mini_epochs']
        self.mixed_precision = False
        if config.get('mixed_precision', False):
            print('rl_games_amd: mixed_precision requested - this path computes in fp32 (parity); ignoring')
        self.last_lr = float(config['learning_rate'])
        self.frame = 0
        self.update_time = 0
        self.mean_rewards = self.last_mean_rewards = -float('inf')
        self.play_time = 0
        self.epoch_num = 0
        self.curr_frames = 0
        self.train_dir = config.get('train_dir', 'runs')
        self.experiment_dir = os.path.join(self.train_dir, self.experiment_name)
        self.nn_dir = os.path.join(self.experiment_dir, 'nn')
        self.summaries_dir = os.path.join(self.experiment_dir, 'summaries')
        self._dirs_made = False
        self.entropy_coef = config['entropy_coef']
        self.writer = None
        self.value_bootstrap = config.get('value_bootstrap', True)
        self.use_smooth_clamp = config.get('use_smooth_clamp', False)
        # which actor loss the kernels evaluate (ops.SURROGATE_*): clipped PPO / smooth clamp / none (ppo: False)
        self.surrogate = 2 if not self.ppo else (1 if self.use_smooth_clamp else 0)
        self.is_tensor_obses = False
        self.aux_loss_dict = {}

        self._init_action_space(config)
        dev = self.ppo_device

        # ---- A2CAgent.__init__ (a2c_continuous.py:18-76) ----
        build_config = {
            'actions_num': self.actions_num, 'input_shape': self.obs_shape,
            'num_seqs': self.num_actors * self.num_agents, 'value_size': self.value_size,
            'normalize_value': self.normalize_value, 'normalize_input': self.normalize_input,
        }
        self.model = self.network.build(build_config)
        self.model.to(dev)
        self.states = None
        self.is_rnn = self.model.is_rnn()
        self.bound_loss_type = config.get('bound_loss_type', 'bound')
        layout = None
        self._grads_overwritten = False
        # (value_size > 1: the fused loss kernels carry one value column - that agent takes autograd through
        #  rl_games_amd/torch_fallback.py instead, SURVEY 8a'-13)
        # (... and so does a state-dependent sigma head, fixed_sigma False: the kernels carry log sigma as a parameter vector)
        # (... and a sigma that is not exp of an unbounded log sigma: logstd_bounds, min_sigma, softplus / linear forms)
        self._general_forms = (not self.is_discrete) and (self.value_size != 1
                                                          or not getattr(self.model.a2c_network, 'fixed_sigma', True)
                                                          or not getattr(self.model.a2c_network, 'plain_sigma', True))
        self._use_engine = ((not self.is_discrete) and config.get('manual_mlp', True) and not self._general_forms
                            and getattr(self.model.a2c_network, 'plain_trunk', True)
                            and not self.model.a2c_network.is_separate_critic()
                            and (not self.is_rnn or config.get('manual_lstm', True)))
        if self._use_engine:
            from .mlp_engine import ManualMLP
            net = self.model.a2c_network
            rest = [p for p in self.model.parameters() if all(p is not q for q in net.parameters())]
            layout = ManualMLP.layout(net) + rest
            # with the engine every gradient slot is overwritten (GEMM out=, column-sum and loss
            # finalise kernels), so the per-step zero fill of the arena is skipped
            self._grads_overwritten = len(rest) == 0
        else:
            layout = self._arena_layout(config)
        self.optimizer = FlatAdam(self.model.parameters(), self.last_lr, eps=1e-08,
                                  weight_decay=self.weight_decay, layout=layout)
        self._engine = None
        if self._use_engine:
            try:
                self._engine = ManualMLP(self.model.a2c_network, self.optimizer,
                                         max(self.minibatch_size, self.num_actors * self.num_agents),
                                         mfma_dw=bool(config.get('mfma_dw', True)),
                                         inplace_act=bool(config.get('inplace_act', True)),
                                         fused_chain=bool(config.get('fused_mlp', True)))
            except NotImplementedError as e:
                print(f'rl_games_amd: manual MLP engine unavailable ({e}); using autograd')
                self._engine = None
                self._grads_overwritten = False
        self._init_chains(config)
        self.dataset = PPODataset(self.batch_size, self.minibatch_size, self.is_discrete, self.is_rnn,
                                  dev, self.seq_length)
        self.central_value_net = None
        if self.has_central_value:                                   # a2c_continuous.py:44-66
            from .central_value import CentralValueTrain
            cv_cfg = self.central_value_config
            cv_builder = PolicyBuilder({'model': cv_cfg.get('model', {'name': 'central_value'}),
                                        'network': cv_cfg['network']})
            self.central_value_net = CentralValueTrain(
                state_shape=self.state_shape, value_size=self.value_size, ppo_device=dev,
                num_agents=self.num_agents, horizon_length=self.horizon_length, num_actors=self.num_actors,
                num_actions=self.actions_num, seq_length=self.seq_length, normalize_value=self.normalize_value,
                network=cv_builder, config=cv_cfg, writter=self.writer, max_epochs=self.max_epochs,
                multi_gpu=self.multi_gpu, zero_rnn_on_done=self.zero_rnn_on_done)
        self.use_experimental_cv = config.get('use_experimental_cv', True)
        if self.normalize_value:
            self.value_mean_std = (self.central_value_net.model.value_mean_std if self.has_central_value
                                   else self.model.value_mean_std)
        if self.normalize_advantage and self.normalize_rms_advantage:
            momentum = config.get('adv_rms_momentum', 0.5)
            self.advantage_mean_std = GeneralizedMovingStats((1,), decay=momentum).to(dev)
        self.has_value_loss = self.use_experimental_cv or not self.has_central_value

        # device-side episode meters (game_rewards / game_shaped_rewards / game_lengths)
        self._meter_sizes = torch.zeros(3, dtype=torch.int32, device=dev)
        self._finished_total = torch.zeros(1, dtype=torch.int64, device=dev)
        self.game_rewards = DeviceAverageMeter(self.value_size, self.games_to_track, dev, self._meter_sizes, 0)
        self.game_shaped_rewards = DeviceAverageMeter(self.value_size, self.games_to_track, dev,
                                                      self._meter_sizes, 1)
        self.game_lengths = DeviceAverageMeter(1, self.games_to_track, dev, self._meter_sizes, 2)
        self.obs = None

        # per-minibatch scratch
        mb = self.minibatch_size
        self._alloc_loss_scratch(mb, dev)
        self._mb_scalars = torch.zeros(max(1, self.mini_epochs_num * self.num_minibatches), 8,
                                       dtype=torch.float32, device=dev)
        self._mb_index = 0
        self._prep_stats = ops.prepare_stats_buffer(dev)
        self._obs_norm = None
        self._host_lr = self.last_lr
        self.train_result = None
        self.kernel_timers = None
        # HIP graphs: one captured graph per minibatch index (forward/loss/backward) + one for the
        # optimiser kernels; the first epoch always runs eagerly (allocations, GEMM selection).
        self._hip_graphs = bool(config.get('hip_graphs', True))
        self._graphs, self._graph_opt, self._graph_sig, self._graph_pool = {}, None, None, None
        self._graph_epoch = None
        self._graph_norm_state = None
        self.last_allreduce = None    # 'ipc' | 'rccl' once a multi-GPU step has run (bench.py reports it)
        self._graph_failed = False
        self._fold_ready = False      # this epoch's minibatch observation moments are precomputed
        self._fin_norm_ok = None      # decided on first use (_norm_in_finalize)
        self._fin_norm_partials = None
        self._norm_ready = None       # (partials, count) when the finalise / all-reduce launch produced the gradient norm
        self._adam_pack = None        # decided on first use (_adam_pack_chain)
        self._lean_pack = None        # decided on first use (_lean_chain)
        self._roll_env_actions = None
        self._ar_norm_partials = None
        self._fold_index = None
        self._ipc_comm = None
        self._rollout_graphs, self._rollout_graph_key, self._rollout_static = {}, None, None
        self._rnn_state_store = None
        self._eager_epochs = 0
        self._graph_rows = torch.zeros(max(1, self.num_minibatches), 8, dtype=torch.float32, device=dev)
        self._obs_norm_mb = (torch.empty((mb,) + tuple(self.obs_shape), dtype=torch.float32, device=dev)
                             if self.normalize_input else None)
        self.algo_observer.after_init(self)

    def _init_action_space(self, config):
        """ContinuousA2CBase.__init__ (a2c_common.py:1484-1498)."""
        self.is_discrete = False
        action_space = self.env_info['action_space']
        if type(action_space).__name__ != 'Box':
            raise ValueError(f'A2CAgent (a2c_continuous) needs a Box action space, got '
                             f'{type(action_space).__name__}; use DiscreteA2CAgent for a2c_discrete')
        self.actions_num = action_space.shape[0]
        self.bounds_loss_coef = config.get('bounds_loss_coef', None)
        self.clip_actions = config.get('clip_actions', True)
        dev = self.ppo_device
        self.actions_low = torch.from_numpy(np.asarray(action_space.low).copy()).float().to(dev)
        self.actions_high = torch.from_numpy(np.asarray(action_space.high).copy()).float().to(dev)

    def _arena_layout(self, config):
        """Physical order of the parameters inside the optimiser's arena for agents outside the continuous engine
        (None: parameters() order)."""
        return None

    def _init_chains(self, config):
        """Hook behind the optimiser: the discrete agent puts its trunks on the fused chain kernels here."""

    def _alloc_loss_scratch(self, mb, dev):
        A = self.actions_num
        self._d_mu = torch.empty(mb, A, dtype=torch.float32, device=dev)
        self._d_val = torch.empty(mb, dtype=torch.float32, device=dev)
        self._loss_blocks = ops.ppo_loss_blocks(mb)
        # (a fused backward launch leaves one row per 16 / 32 / 64-row workgroup)
        self._loss_partials = torch.empty(max(self._loss_blocks, (mb + 15) // 16),
                                          ops.ppo_loss_partials_per_block(A), dtype=torch.float64, device=dev)

    def _rollout_fields(self):
        return ['actions', 'neglogpacs', 'values', 'mus', 'sigmas']

    def _supports_action_masks(self):
        return False

    # ================================================================== small helpers
    @property
    def device(self):
        return self.ppo_device

    def _ensure_dirs(self):
        if not self._dirs_made:
            for d in (self.train_dir, self.experiment_dir, self.nn_dir, self.summaries_dir):
                os.makedirs(d, exist_ok=True)
            if self.global_rank == 0:
                self.writer = _make_writer(self.summaries_dir)
            self._dirs_made = True

    def _shaper_params(self):
        s = self.rewards_shaper
        get = (lambda k, d: s.get(k, d)) if isinstance(s, dict) else (lambda k, d: getattr(s, k, d))
        return (float(get('shift_value', 0)), float(get('scale_value', 1)),
                float(get('min_val', -np.inf)), float(get('max_val', np.inf)), bool(get('log_val', False)))

    def _uses_observer_infos(self):
        fn = getattr(type(self.algo_observer), 'process_infos', None)
        return fn is not None and fn is not NullObserver.process_infos and \
            getattr(fn, '__qualname__', '').split('.')[0] != 'AlgoObserver'

    def update_epoch(self):
        self.epoch_num += 1
        return self.epoch_num

    def set_eval(self):
        self.model.eval()
        if self.normalize_rms_advantage:
            self.advantage_mean_std.eval()
        if self.epochs_between_resets > 0 and self.epoch_num % self.epochs_between_resets == 0:
            self.reset_envs()
            rows = self.num_agents * self.num_actors
            self.init_current_rewards(rows, (rows, self.value_size))

    def set_train(self):
        self.model.train()
        if self.normalize_rms_advantage:
            self.advantage_mean_std.train()

    def update_lr(self, lr):
        """a2c_common.py:564-576.  The per-minibatch adaptive schedule never calls this (the lr
        moves on the device); host-driven schedules and user code do."""
        self.last_lr = lr
        self._host_lr = lr
        self.optimizer.set_lr(lr)

    def _sync_lr_to_host(self):
        self.last_lr = self.optimizer.current_lr()
        return self.last_lr

    # ================================================================== model queries
    def _preproc_obs(self, obs_batch):
        if obs_batch.dtype == torch.uint8:
            obs_batch = obs_batch.float() / 255.0
        return obs_batch

    def get_action_values(self, obs):
        processed_obs = self._preproc_obs(obs['obs'])
        self.model.eval()
        input_dict = {'is_train': False, 'prev_actions': None, 'obs': processed_obs,
                      'rnn_states': self.rnn_states}
        with torch.no_grad():
            res_dict = self.model(input_dict)
            if self.has_central_value:                               # a2c_common.py:593-600
                res_dict['values'] = self.get_central_value({'is_train': False, 'states': obs['states']})
        return res_dict

    def get_central_value(self, obs_dict):
        return self.central_value_net.get_value(obs_dict)

    def train_central_value(self):
        return self.central_value_net.train_net()

    def get_values(self, obs):
        with torch.no_grad():
            if self.has_central_value:                               # a2c_common.py:605-614
                self.central_value_net.eval()
                return self.get_central_value({'is_train': False, 'states': obs['states'], 'actions': None,
                                               'is_done': self.dones})
            self.model.eval()
            processed_obs = self._preproc_obs(obs['obs'])
            input_dict = {'is_train': False, 'prev_actions': None, 'obs': processed_obs,
                          'rnn_states': self.rnn_states}
            return self.model(input_dict)['values']

    def get_masked_action_values(self, obs, action_masks):
        raise NotImplementedError('Masked action values are not implemented for continuous actions')

    # ================================================================== env plumbing
    def cast_obs(self, obs):
        if isinstance(obs, torch.Tensor):
            self.is_tensor_obses = True
            if obs.device != torch.device(self.ppo_device):
                obs = obs.to(self.ppo_device)
        elif isinstance(obs, np.ndarray):
            assert obs.dtype != np.int8
            if obs.dtype == np.uint8:
                obs = torch.from_numpy(obs).to(self.ppo_device)
            else:
                obs = torch.from_numpy(obs).float().to(self.ppo_device)
        return obs

    def obs_to_tensors(self, obs):
        if isinstance(obs, dict):
            upd = {k: self.cast_obs(v) for k, v in obs.items()}
            return upd if 'obs' in obs else {'obs': upd}
        return {'obs': self.cast_obs(obs)}

    def preprocess_actions(self, actions):
        if self.clip_actions:
            clamped = torch.clamp(actions, -1.0, 1.0)
            rescaled = rescale_actions(self.actions_low, self.actions_high, clamped)
        else:
            rescaled = actions
        if not self.is_tensor_obses:
            rescaled = rescaled.cpu().numpy()
        return rescaled

    def env_step(self, actions, preprocessed=None):
        """`preprocessed`: device tensor already clamped/rescaled by the fused rollout step."""
        if preprocessed is None:
            actions = self.preprocess_actions(actions)
        else:
            actions = preprocessed if self.is_tensor_obses else preprocessed.cpu().numpy()
        obs, rewards, dones, infos = self.vec_env.step(actions)
        dev = self.ppo_device
        if not isinstance(rewards, torch.Tensor):
            rewards = torch.from_numpy(np.asarray(rewards)).to(dev, dtype=torch.float32)
            dones = torch.from_numpy(np.asarray(dones)).to(dev)
        else:
            rewards, dones = rewards.to(dev), dones.to(dev)
        if self.value_size == 1 and rewards.dim() == 1:
            rewards = rewards.unsqueeze(1)
        return self.obs_to_tensors(obs), rewards, dones, infos

    def env_reset(self):
        obs = self.vec_env.reset()
        obs = self.obs_to_tensors(obs)
        self._autoreset_prev_dones = None
        return obs

    def reset_envs(self):
        if self.is_rnn:
            self.rnn_states = [s.to(self.ppo_device) for s in self.model.get_default_rnn_state()]
        self.obs = self.env_reset()

    # ================================================================== buffers
    def init_tensors(self):
        rows = self.num_agents * self.num_actors
        algo_info = {'num_actors': self.num_actors, 'horizon_length': self.horizon_length,
                     'has_central_value': self.has_central_value, 'use_action_masks': self.use_action_masks}
        self.experience_buffer = ExperienceBuffer(self.env_info, algo_info, self.ppo_device)
        self.init_current_rewards(rows, (rows, self.value_size))
        dev = self.ppo_device
        self._post_blocks = ops.post_step_num_blocks(rows)
        self._ep_partials = torch.zeros(self.horizon_length, self._post_blocks, 2 * self.value_size + 2,
                                        dtype=torch.float64, device=dev)
        B = self.batch_size
        self._returns = torch.empty(rows, self.horizon_length, dtype=torch.float32, device=dev)
        self._advantages = torch.empty(rows, self.horizon_length, dtype=torch.float32, device=dev)
        self._norm_values = torch.empty(B, 1, dtype=torch.float32, device=dev)
        self._norm_returns = torch.empty(B, 1, dtype=torch.float32, device=dev)
        self._norm_advantages = torch.empty(B, dtype=torch.float32, device=dev)
        from .gae import num_moment_partials
        self._gae_partials = torch.empty(num_moment_partials(rows), 6, dtype=torch.float64, device=dev)
        self.update_list = self._rollout_fields()
        self.tensor_list = self.update_list + ['obses', 'states', 'dones']
        if self._engine is not None:
            self._roll_obs_norm = torch.empty((rows,) + tuple(self.obs_shape), dtype=torch.float32, device=dev)
            self._roll_noise = torch.empty(rows, self.actions_num, dtype=torch.float32, device=dev)
            self._roll_actions = torch.empty(rows, self.actions_num, dtype=torch.float32, device=dev)
            self._roll_values = torch.empty(rows, dtype=torch.float32, device=dev)
        if self.is_rnn:
            self.rnn_states = [s.to(dev) for s in self.model.get_default_rnn_state()]
            num_seqs = self.horizon_length // self.seq_length
            if (self.horizon_length * rows // self.num_minibatches) % self.seq_length != 0:
                raise ValueError(f'Horizon length ({self.horizon_length}) times total agents ({rows}) divided by '
                                 f'num minibatches ({self.num_minibatches}) must be divisible by sequence '
                                 f'length ({self.seq_length})')
            self.mb_rnn_states = [torch.zeros((num_seqs, s.size()[0], rows, s.size()[2]),
                                              dtype=torch.float32, device=dev) for s in self.rnn_states]

    def init_current_rewards(self, batch_size, current_rewards_shape):
        dev = self.ppo_device
        self.current_rewards = torch.zeros(current_rewards_shape, dtype=torch.float32, device=dev)
        self.current_shaped_rewards = torch.zeros(current_rewards_shape, dtype=torch.float32, device=dev)
        self.current_lengths = torch.zeros(batch_size, dtype=torch.float32, device=dev)
        self.dones = torch.ones((batch_size,), dtype=torch.uint8, device=dev)

    def discount_values(self, fdones, last_extrinsic_values, mb_fdones, mb_extrinsic_values, mb_rewards):
        """a2c_common.py:729-734 (function seam kept for subclasses)."""
        return compute_gae(mb_rewards, mb_extrinsic_values, mb_fdones, last_extrinsic_values, fdones,
                           self.gamma, self.tau)

    def clear_stats(self, clean_rewards=True):
        self.game_rewards.clear()
        self.game_shaped_rewards.clear()
        self.game_lengths.clear()
        if clean_rewards:
            self.mean_rewards = self.last_mean_rewards = -float('inf')
        self.algo_observer.after_clear_stats()

    # ================================================================== rollout
    def _as_u8(self, dones):
        if dones.dtype == torch.uint8:
            return dones
        if dones.dtype == torch.bool:
            return dones.view(torch.uint8)
        return (dones != 0).to(torch.uint8)

    def _rollout_step_tail(self, n, res_dict, rewards, infos, mb_valid):
        """Everything after vec_env.step for step n (a2c_common.py:1021-1051)."""
        buf = self.experience_buffer
        time_outs = None
        if self.value_bootstrap and isinstance(infos, dict) and 'time_outs' in infos:
            time_outs = self.cast_obs(infos['time_outs'])
            if time_outs.dtype not in (torch.uint8, torch.bool, torch.float32):
                time_outs = time_outs.float()
        values = res_dict['values']
        if not values.is_contiguous():
            values = values.contiguous()
        if not rewards.is_contiguous():
            rewards = rewards.contiguous()
        live = None if mb_valid is None else mb_valid[n]
        ops.rollout_post_step(rewards.float() if rewards.dtype != torch.float32 else rewards, self.dones,
                              time_outs, values, live, buf.storage['rewards'], self.current_rewards,
                              self.current_shaped_rewards, self.current_lengths, self._ep_partials,
                              self._shaper, time_outs is not None, self.gamma, self.horizon_length, n,
                              num_agents=self.num_agents)
        if self._observer_needs_infos:
            done_indices = self.dones.nonzero(as_tuple=False)[::self.num_agents]
            self.algo_observer.process_infos(infos, done_indices)

    def _finish_rollout(self, batch_dict, mb_valid, step_time):
        buf = self.experience_buffer
        H = self.horizon_length
        rows = self.num_agents * self.num_actors
        ops.episode_meters_update(self._ep_partials, H, self._post_blocks, self.value_size,
                                  self.games_to_track, self.game_rewards.mean, self.game_shaped_rewards.mean,
                                  self.game_lengths.mean, self._meter_sizes, self._finished_total)
        if self._fast_rollout_ok():
            last_values = self._fast_values(self.obs)
        else:
            last_values = self.get_values(self.obs)
        if self.value_size == 1 and ops._lib.load().rlg_gae_envmajor_supported(H):
            lv = last_values.reshape(-1).contiguous()
            timers = self.kernel_timers
            ev = None
            if timers is not None:      # bench.py: HIP events on the launch stream around the kernel
                from .gae import HipEventPair
                ev = HipEventPair()
                timers.setdefault('gae_envmajor_fused', []).append(ev)
            gae_returns_advantages(buf.storage['rewards'].view(rows, H), buf.storage['values'].view(rows, H),
                                   buf.storage['dones'], lv, self.dones, self.gamma, self.tau,
                                   out_returns=self._returns, out_advantages=self._advantages,
                                   moment_partials=self._gae_partials, events=ev)
            batch_dict['returns'] = self._returns.view(rows * H, 1)
            batch_dict['_fused'] = {'advantages': self._advantages.view(rows * H), 'partials': self._gae_partials}
        else:
            mb_advs = self.discount_values(self.dones, last_values, buf.tensor_dict['dones'],
                                           buf.tensor_dict['values'], buf.tensor_dict['rewards'])
            batch_dict['returns'] = swap_and_flatten01(mb_advs + buf.tensor_dict['values'])
        batch_dict['played_frames'] = self.batch_size
        batch_dict['step_time'] = step_time
        if mb_valid is not None:
            batch_dict['rnn_masks'] = swap_and_flatten01(mb_valid)
        return batch_dict

    def _fast_policy_step(self, n):
        """Rollout forward of step n on the engine (see _policy_step_kernels).  From the second epoch
        on the launches of a step are replayed as one HIP graph per step index; its inputs must live at
        fixed addresses: the buffer slot of the step (plain float observations) or static copies of the
        env-owned tensors (recurrent policies, integer observations)."""
        if not self._rollout_graphs_usable():
            return self._policy_step_kernels(n, self.obs['obs'], self.dones, self.rnn_states)
        key = (id(self.experience_buffer), tuple(self.obs['obs'].shape), self.obs['obs'].dtype)
        buf = self.experience_buffer
        obs = self.obs['obs']
        # Plain float observations without recurrent state: the step's observations / done flags go from the
        # env's tensors straight into the buffer (the one launch that had to happen anyway) and the captured
        # forward reads the observations from that buffer slot - no staging copies in front of the graph.
        direct = (not self.is_rnn and obs.dtype == torch.float32 and obs.dim() == 2 and obs.is_contiguous()
                  and torch.is_tensor(buf.storage['obses']) and self.config.get('rollout_obs_from_buffer', True))
        if self._rollout_graph_key != key:
            self._rollout_graphs.clear()
            self._rollout_graph_key = key
            self._rollout_static = None
        if direct:
            buf.store_step(n, {'obses': obs, 'dones': self.dones})
            args = (n, buf.storage['obses'][:, n, :], None, None, False)
        else:
            if self._rollout_static is None:
                st = {'obs': torch.empty_like(obs).contiguous(), 'dones': torch.empty_like(self.dones)}
                if self.is_rnn:
                    st['rnn'] = [torch.empty_like(s) for s in self.rnn_states]
                self._rollout_static = st
            st = self._rollout_static
            st['obs'].copy_(obs)
            st['dones'].copy_(self.dones)
            if self.is_rnn:
                for dst, src in zip(st['rnn'], self.rnn_states):
                    dst.copy_(src)
            args = (n, st['obs'], st['dones'], st.get('rnn'), True)
        entry = self._rollout_graphs.get((n, direct))
        if entry is None:
            if self._graph_pool is None:
                self._graph_pool = torch.cuda.graph_pool_handle()
            g = torch.cuda.CUDAGraph()
            cached0 = self._chain_cache_states()
            try:
                with torch.cuda.graph(g, pool=self._graph_pool, capture_error_mode='thread_local'):
                    out = self._policy_step_kernels(*args)
            finally:
                self._restore_chain_cache_states(cached0)       # (launches of the body were recorded, not run)
            entry = self._rollout_graphs[(n, direct)] = (g, out)
        entry[0].replay()
        return entry[1]

    def _rollout_graphs_usable(self):
        return (self._hip_graphs and self.config.get('rollout_graphs', True) and self._eager_epochs >= 1
                and not self._graph_failed and isinstance(self.obs['obs'], torch.Tensor))

    def _policy_step_kernels(self, n, obs_raw, dones, rnn_states, store=True):
        """obs normalise -> engine GEMMs -> fused policy-head kernel that also writes actions / mus /
        sigmas / neglogpacs / values of the step into the buffer -> obs + dones into the buffer ->
        action clamp/rescale for the env.  Same maths as get_action_values + update_data +
        preprocess_actions; no autograd, nothing that depends on the host."""
        eng, buf = self._engine, self.experience_buffer
        obs = self._preproc_obs(obs_raw)
        if not obs.is_contiguous() and (eng.chain is None or obs.stride(-1) != 1):
            obs = obs.contiguous()          # (the fused forward takes any row stride: a buffer slot [:, n, :])
        rows = obs.shape[0]
        if eng.chain is not None:
            # normaliser + every layer + heads in one launch; nothing but the heads is written
            heads = eng.forward_obs(obs, self._obs_rms(), self._obs_eps(), keep=False)
        elif self.is_rnn and eng.chain_rnn is not None and obs.dtype == torch.float32:
            # recurrent policy: normaliser + trunk + gate-input product in one launch, then the LSTM step + heads
            heads = eng.forward(obs, keep=False, rnn_states=rnn_states, seq_length=1,
                                raw_rms=self._obs_rms() or (), eps=self._obs_eps())
        else:
            if self.normalize_input:
                m = self.model.running_mean_std
                obs_n = ops.rms_apply(obs, m.running_mean, m.running_var, m.epsilon, 0, out=self._roll_obs_norm)
            else:
                obs_n = obs
            if self.is_rnn:
                heads = eng.forward(obs_n, keep=False, rnn_states=rnn_states, seq_length=1)
            else:
                heads = eng.forward(obs_n, keep=False)
        torch.randn(self._roll_noise.shape, device=self._roll_noise.device, out=self._roll_noise)
        vs = None
        eps = 1e-5
        if self.normalize_value:
            vm = self.model.value_mean_std
            vs, eps = (vm.running_mean, vm.running_var), vm.epsilon
        env_actions = None
        if self.clip_actions:
            # clamp + rescale for the env ride along in the same launch (3 element-wise launches less per step)
            if self._roll_env_actions is None or self._roll_env_actions.shape != self._roll_actions.shape:
                self._roll_env_actions = torch.empty_like(self._roll_actions)
            env_actions = (self._roll_env_actions, self.actions_low, self.actions_high)
        ops.rollout_policy_head(heads, self.model.a2c_network.sigma.data, self._roll_noise, vs, eps,
                                self._roll_actions, self._roll_values, buf.storage, self.horizon_length, n,
                                env_actions=env_actions)
        # the buffer keeps the observation as the env delivered it (a2c_common.py:1000), not the
        # /255-preprocessed copy
        if store:
            buf.store_step(n, {'obses': obs_raw if obs_raw.is_contiguous() else obs_raw.contiguous(), 'dones': dones})
        res = {'actions': self._roll_actions, 'values': self._roll_values.view(rows, 1)}
        res['env_actions'] = self._roll_env_actions if self.clip_actions else self._roll_actions
        if self.is_rnn:
            res['rnn_states'] = eng.last_states
        return res

    def _fast_values(self, obs):
        """get_values on the engine: de-normalised critic values [N] of `obs`."""
        eng = self._engine
        x = self._preproc_obs(obs['obs'])
        if not x.is_contiguous():
            x = x.contiguous()
        if eng.chain is not None:
            heads = eng.forward_obs(x, self._obs_rms(), self._obs_eps(), keep=False)
        elif self.is_rnn and eng.chain_rnn is not None and x.dtype == torch.float32:
            heads = eng.forward(x, keep=False, rnn_states=self.rnn_states, seq_length=1,
                                raw_rms=self._obs_rms() or (), eps=self._obs_eps())
        else:
            if self.normalize_input:
                m = self.model.running_mean_std
                x = ops.rms_apply(x, m.running_mean, m.running_var, m.epsilon, 0, out=self._roll_obs_norm)
            if self.is_rnn:
                heads = eng.forward(x, keep=False, rnn_states=self.rnn_states, seq_length=1)
            else:
                heads = eng.forward(x, keep=False)
        v = heads[:, 0].contiguous()
        if self.normalize_value:
            vm = self.model.value_mean_std
            v = ops.rms_apply(v.view(-1, 1), vm.running_mean, vm.running_var, vm.epsilon, 1).view(-1)
        return v

    def _obs_rms(self):
        if not self.normalize_input:
            return None
        m = self.model.running_mean_std
        return (m.running_mean, m.running_var)

    def _obs_eps(self):
        return self.model.running_mean_std.epsilon if self.normalize_input else 1e-5

    def _fast_rollout_ok(self):
        return (self._engine is not None and self.value_size == 1 and not self.has_central_value
                and self.config.get('fused_rollout', True))

    def play_steps(self):
        """a2c_common.py:985-1069."""
        buf = self.experience_buffer
        self._shaper = self._shaper_params()
        self._observer_needs_infos = self._uses_observer_infos()
        step_time = 0.0
        mb_valid = None
        fast = self._fast_rollout_ok()
        if fast:
            # the step graphs contain no pack launch: the weights' derived forms (plane fragments of both kernel families) must belong
            # to the weights as they are - they do behind an optimiser step, not behind set_weights / a broadcast
            self._planes_before_replay()
        if self.mask_autoreset_rows:
            mb_valid = torch.ones((self.horizon_length, self.num_actors * self.num_agents),
                                  dtype=torch.float32, device=self.ppo_device)
        for n in range(self.horizon_length):
            if fast:
                res_dict = self._fast_policy_step(n)
            else:
                if self.use_action_masks:                        # a2c_common.py:995-997
                    res_dict = self.get_masked_action_values(self.obs, self.vec_env.get_action_masks())
                else:
                    res_dict = self.get_action_values(self.obs)
                fields = {'obses': self.obs['obs'], 'dones': self.dones}
                for k in self.update_list:
                    fields[k] = res_dict[k]
                if self.has_central_value:
                    fields['states'] = self.obs['states']
                buf.store_step(n, fields)
            if mb_valid is not None:
                prev = self._autoreset_prev_dones
                if prev is None:
                    prev = torch.zeros_like(self.dones)
                mb_valid[n] = 1.0 - prev.float()
            t0 = time.perf_counter()
            self.obs, rewards, dones, infos = self.env_step(res_dict['actions'], res_dict.get('env_actions'))
            self.dones = self._as_u8(dones)
            if self.mask_autoreset_rows:
                self._autoreset_prev_dones = self.dones.clone()
            step_time += time.perf_counter() - t0
            self._rollout_step_tail(n, res_dict, rewards, infos, mb_valid)
        batch_dict = buf.get_transformed_list(swap_and_flatten01, self.tensor_list)
        return self._finish_rollout(batch_dict, mb_valid, step_time)

    def play_steps_rnn(self):
        """a2c_common.py:1071-1202."""
        buf = self.experience_buffer
        self._shaper = self._shaper_params()
        self._observer_needs_infos = self._uses_observer_infos()
        step_time = 0.0
        mb_valid = None
        rows = self.num_actors * self.num_agents
        fast = self._fast_rollout_ok()
        if fast:
            self._planes_before_replay()          # (see play_steps)
        if self.mask_autoreset_rows:
            mb_valid = torch.ones((self.horizon_length, rows), dtype=torch.float32, device=self.ppo_device)
        for n in range(self.horizon_length):
            if n % self.seq_length == 0:
                for s, mb_s in zip(self.rnn_states, self.mb_rnn_states):
                    mb_s[n // self.seq_length, :, :, :] = s
            if self.has_central_value:
                self.central_value_net.pre_step_rnn(n)           # a2c_common.py:1085-1086
            if fast:
                res_dict = self._fast_policy_step(n)
            elif self.use_action_masks:                          # a2c_common.py:1088-1090
                res_dict = self.get_masked_action_values(self.obs, self.vec_env.get_action_masks())
            else:
                res_dict = self.get_action_values(self.obs)
            self.rnn_states = [s.contiguous() for s in res_dict['rnn_states']]
            fields = {'obses': self.obs['obs'], 'dones': self.dones}
            if mb_valid is not None:
                prev = self._autoreset_prev_dones
                if prev is None:
                    prev = torch.zeros_like(self.dones)
                mb_valid[n] = 1.0 - prev.float()
                if self.zero_rnn_on_done:
                    for s in self.rnn_states:
                        ops.rnn_zero_done_states(s, prev)
                    if self.has_central_value:                   # (the critic absorbed the same filler row: :1112-1115)
                        self.central_value_net.zero_states_where(prev)
            if not fast:
                for k in self.update_list:
                    fields[k] = res_dict[k]
                if self.has_central_value:
                    fields['states'] = self.obs['states']
                buf.store_step(n, fields)
            t0 = time.perf_counter()
            self.obs, rewards, dones, infos = self.env_step(res_dict['actions'], res_dict.get('env_actions'))
            self.dones = self._as_u8(dones)
            if self.mask_autoreset_rows:
                self._autoreset_prev_dones = self.dones.clone()
            step_time += time.perf_counter() - t0
            if self.zero_rnn_on_done:
                for s in self.rnn_states:
                    ops.rnn_zero_done_states(s, self.dones)
            if self.has_central_value:                           # a2c_common.py:1151-1152
                self.central_value_net.zero_states_where(self.dones)
            self._rollout_step_tail(n, res_dict, rewards, infos, mb_valid)
        batch_dict = buf.get_transformed_list(swap_and_flatten01, self.tensor_list)
        batch_dict = self._finish_rollout(batch_dict, mb_valid, step_time)
        if mb_valid is not None and self.zero_rnn_on_done:
            rnn_dones = buf.tensor_dict['dones'].clone()
            garbage = (mb_valid == 0.0)
            rnn_dones[1:] = torch.maximum(rnn_dones[1:], garbage[:-1].to(rnn_dones.dtype))
            batch_dict['dones'] = swap_and_flatten01(rnn_dones)
        states = []
        for k, mb_s in enumerate(self.mb_rnn_states):
            t_size = mb_s.size()[0] * mb_s.size()[2]
            h_size = mb_s.size()[3]
            flat = mb_s.permute(1, 2, 0, 3).reshape(-1, t_size, h_size)
            if self._rnn_state_store is None or len(self._rnn_state_store) <= k:
                self._rnn_state_store = (self._rnn_state_store or []) + [torch.empty_like(flat)]
            self._rnn_state_store[k].copy_(flat)      # same storage every epoch (HIP-graph replays)
            states.append(self._rnn_state_store[k])
        batch_dict['rnn_states'] = states
        return batch_dict

    # ================================================================== dataset
    def prepare_dataset(self, batch_dict):
        """a2c_common.py:1586-1660."""
        returns = batch_dict['returns']
        values = batch_dict['values']
        rnn_masks = batch_dict.get('rnn_masks', None)
        B = returns.shape[0]
        if self.value_size != 1:
            return self._prepare_dataset_general(batch_dict)
        fused = batch_dict.get('_fused')
        values_flat = values.reshape(-1)
        returns_flat = returns.reshape(-1)
        if not values_flat.is_contiguous():
            values_flat = values_flat.contiguous()
        if not returns_flat.is_contiguous():
            returns_flat = returns_flat.contiguous()
        mask = None
        if rnn_masks is not None:
            mask = rnn_masks.reshape(-1).float().contiguous()
        if fused is not None and mask is None:
            advantages, partials = fused['advantages'], fused['partials']
        else:
            advantages = returns_flat - values_flat                                     # :1598
            partials = ops.triple_moments(advantages, values_flat, returns_flat, mask)
        flags = 0
        value_stats = None
        if self.normalize_value:
            flags |= ops.PREP_NORM_VALUE
            if self.config.get('freeze_critic', False):
                flags |= ops.PREP_FREEZE_CRITIC
            m = self.value_mean_std
            value_stats = (m.running_mean, m.running_var, m.count)
        ema = None
        if self.normalize_advantage:
            if self.normalize_rms_advantage:
                flags |= ops.PREP_EMA_ADV
                ema = self.advantage_mean_std.kernel_state()
            else:
                flags |= ops.PREP_NORM_ADV
        eps = self.value_mean_std.epsilon if self.normalize_value else 1e-5
        if B > self._norm_values.shape[0]:
            raise ValueError('batch larger than the allocated dataset buffers')
        nv, nr, na = self._norm_values[:B], self._norm_returns[:B], self._norm_advantages[:B]
        if flags:
            ops.prepare_finalize(partials, B, flags, value_stats, eps, ema, self._prep_stats)
        else:
            self._prep_stats.zero_()
        ops.prepare_apply(values_flat, returns_flat, advantages, flags, self._prep_stats,
                          out=(nv.view(-1), nr.view(-1), na))
        if self.normalize_value:
            self.value_mean_std.eval()

        dataset_dict = {
            'old_values': nv, 'old_logp_actions': batch_dict['neglogpacs'], 'advantages': na,
            'returns': nr, 'actions': batch_dict['actions'], 'obs': batch_dict['obses'],
            'dones': batch_dict['dones'], 'rnn_states': batch_dict.get('rnn_states', None),
            'rnn_masks': rnn_masks,
        }
        if not self.is_discrete:
            dataset_dict['mu'], dataset_dict['sigma'] = batch_dict['mus'], batch_dict['sigmas']
        if self.use_action_masks:                                    # a2c_common.py:1346-1347
            dataset_dict['action_masks'] = batch_dict['action_masks']
        self.dataset.update_values_dict(dataset_dict)
        if self.has_central_value:                                   # a2c_common.py:1651-1660
            self.central_value_net.update_dataset({
                'old_values': nv, 'advantages': na, 'returns': nr, 'actions': batch_dict['actions'],
                'obs': batch_dict['states'], 'dones': batch_dict['dones'], 'rnn_masks': rnn_masks,
            })

    def _prepare_dataset_general(self, batch_dict):
        """prepare_dataset for value_size > 1 (returns / values [B, V]): the reference's operation sequence
        (a2c_common.py:1598-1634) with the normaliser modules and torch ops - the fused prepare kernels carry one value
        column.  advantages = returns - values BEFORE the value normalisation, summed over V behind it (:1622)."""
        from . import torch_fallback as tf
        returns, values = batch_dict['returns'], batch_dict['values']
        rnn_masks = batch_dict.get('rnn_masks', None)
        advantages = returns - values                                                   # :1598
        if self.normalize_value:
            vms = self.value_mean_std
            if self.config.get('freeze_critic', False):                                 # :1601-1604
                vms.eval()
                values, returns = vms(values), vms(returns)
            elif rnn_masks is not None:                                                 # :1605-1615
                valid = rnn_masks.reshape(-1).bool()
                vms.train()
                vms(values[valid])
                vms(returns[valid])
                vms.eval()
                values, returns = vms(values), vms(returns)
            else:                                                                       # :1616-1620
                vms.train()
                values = vms(values)
                returns = vms(returns)
                vms.eval()
        advantages = torch.sum(advantages, dim=1)                                       # :1622
        if self.normalize_advantage:
            if self.normalize_rms_advantage:
                advantages = self.advantage_mean_std(advantages, mask=rnn_masks)
            else:
                advantages = tf.normalize_advantages(advantages, rnn_masks)
        dataset_dict = {
            'old_values': values, 'old_logp_actions': batch_dict['neglogpacs'], 'advantages': advantages,
            'returns': returns, 'actions': batch_dict['actions'], 'obs': batch_dict['obses'],
            'dones': batch_dict['dones'], 'rnn_states': batch_dict.get('rnn_states', None), 'rnn_masks': rnn_masks,
            'mu': batch_dict['mus'], 'sigma': batch_dict['sigmas'],
        }
        self.dataset.update_values_dict(dataset_dict)
        if self.has_central_value:
            self.central_value_net.update_dataset({
                'old_values': values, 'advantages': advantages, 'returns': returns, 'actions': batch_dict['actions'],
                'obs': batch_dict['states'], 'dones': batch_dict['dones'], 'rnn_masks': rnn_masks,
            })

    def _forward_loss_backward_general(self, input_dict, row):
        """calc_gradients up to the gradients in the arena for value_size > 1: forward with autograd, the losses as torch
        ops (rl_games_amd/torch_fallback.py: a2c_continuous.py:97-134, :173-221), loss.backward() into the arena views.
        Leaves the same things behind as the fused path: the five scalars in `row`, the KL in the arena's tail slot (the
        device-side lr rule and the multi-GPU average read it there), mu / sigma written back into the dataset."""
        from . import torch_fallback as tf
        opt = self.optimizer
        batch = {'is_train': True, 'prev_actions': input_dict['actions'], 'obs': self._preproc_obs(input_dict['obs'])}
        if self.is_rnn:
            batch['rnn_states'] = input_dict['rnn_states']
            batch['seq_length'] = self.seq_length
            if self.zero_rnn_on_done:
                batch['dones'] = input_dict['dones']
        rnn_masks = input_dict.get('rnn_masks', None)
        mask = None if rnn_masks is None else rnn_masks.reshape(-1).float()
        opt.zero_grad()
        mu, logstd, values, _ = self.model.forward_heads(batch)
        mb = mu.shape[0]
        net = self.model.a2c_network
        kind = 0 if self.bounds_loss_coef is None else ops.BOUND_KINDS.get(self.bound_loss_type, 0)
        loss, scalars, sigma = tf.ppo_loss(
            mu, logstd, values.reshape(mb, -1), input_dict['actions'], input_dict['old_logp_actions'], input_dict['advantages'],
            input_dict['old_values'].reshape(mb, -1), input_dict['returns'].reshape(mb, -1), e_clip=self.e_clip,
            critic_coef=self.critic_coef if self.has_value_loss else 0.0, entropy_coef=self.entropy_coef,
            bounds_coef=self.bounds_loss_coef if self.bounds_loss_coef is not None else 0.0, bound_kind=kind,
            clip_value=self.clip_value, smooth=self.surrogate, mask=mask,
            sigma_fn=None if net.plain_sigma else net.sigma_and_logstd)
        loss.backward()
        with torch.no_grad():
            sig = sigma.detach().expand_as(mu)
            kl = tf.policy_kl(mu.detach(), sig, input_dict['mu'], input_dict['sigma'], mask)
            row[0], row[1], row[2], row[3], row[4] = (scalars['a_loss'], scalars['c_loss'], scalars['entropy'],
                                                      scalars['b_loss'], kl)
            opt.kl_slot.copy_(kl.reshape(1))
            input_dict['mu'].copy_(mu.detach())                                         # datasets.py:33-43
            input_dict['sigma'].copy_(sig)
        self._norm_ready = None

    # ================================================================== update
    def train_actor_critic(self, input_dict):
        self.set_train()
        self.calc_gradients(input_dict)
        return self.train_result

    def calc_gradients(self, input_dict):
        """a2c_continuous.py:136-234 - forward, fused loss + analytic backward + KL, optimiser."""
        row = self._mb_scalars[self._mb_index % self._mb_scalars.shape[0]]
        self._mb_index += 1
        self._forward_loss_backward(input_dict, row)
        self.trancate_gradients_and_step()
        # dataset.update_mu_sigma happened inside the loss kernel (write_back)
        c_loss = row[1] if self.has_value_loss else torch.zeros((), device=row.device)
        self.train_result = (row[0], c_loss, row[2], row[4], self._host_lr, 1.0,
                             input_dict['mu'], input_dict['sigma'], row[3])

    def _forward_loss_backward(self, input_dict, row):
        """Everything of calc_gradients up to (and including) the gradients in the arena.  No
        host-side scalars change between calls for a given minibatch slice, so this body is what
        gets captured into a HIP graph per minibatch index."""
        if self._general_forms:
            return self._forward_loss_backward_general(input_dict, row)
        opt = self.optimizer
        net = self.model.a2c_network
        obs_batch = self._preproc_obs(input_dict['obs'])
        rnn_masks = input_dict.get('rnn_masks', None)
        batch = {'is_train': True, 'prev_actions': input_dict['actions'], 'obs': obs_batch}
        if self.is_rnn:
            batch['rnn_states'] = input_dict['rnn_states']
            batch['seq_length'] = self.seq_length
            if self.zero_rnn_on_done:
                batch['dones'] = input_dict['dones']

        eng = self._engine
        if eng is None or not self._grads_overwritten:
            opt.zero_grad()
        if eng is not None:
            with torch.no_grad():
                if eng.chain is not None:
                    if not obs_batch.is_contiguous():
                        obs_batch = obs_batch.contiguous()
                    rms, fold = self._obs_rms(), None
                    if self.normalize_input and self.model.running_mean_std.training:
                        if self._fold_index is not None:
                            # statistics first (models.py:54-56), in the launch's prologue, from the
                            # epoch's precomputed minibatch moments
                            rms, fold = self.model.running_mean_std.fold_buffers(self._fold_index)
                        else:
                            self.model.running_mean_std.update(obs_batch)
                    # forward + loss + backward as ONE launch where the chain has that form (minibatches < 16,384 rows,
                    # the loss evaluated by the backward launch): the forward is handed to eng.backward() below
                    defer = (self.config.get('fused_step16', True) and self.config.get('fold_loss_finalize', True)
                             and self.config.get('loss_in_backward', True)
                             and not eng.chain.split_products(obs_batch.shape[0], 2))
                    heads = eng.forward_obs(obs_batch, rms, self._obs_eps(), rms_fold=fold, defer=defer)
                    obs_n = None
                elif self.is_rnn and eng.chain_rnn is not None and obs_batch.dtype == torch.float32:
                    # recurrent policy on the fused trunk: the statistics update stays its own (two) launches, the
                    # normalisation happens inside the trunk launch
                    if not obs_batch.is_contiguous():
                        obs_batch = obs_batch.contiguous()
                    if self.normalize_input and self.model.running_mean_std.training:
                        self.model.running_mean_std.update(obs_batch)
                    heads = eng.forward(obs_batch, rnn_states=batch['rnn_states'], dones=batch.get('dones'),
                                        seq_length=self.seq_length, raw_rms=self._obs_rms() or (), eps=self._obs_eps())
                    obs_n = None
                elif self.normalize_input:                              # updates the obs statistics
                    out = self._obs_norm_mb[:obs_batch.shape[0]] if self._obs_norm_mb is not None else None
                    obs_n = self.model.running_mean_std(obs_batch, out=out)
                else:
                    obs_n = obs_batch
                if eng.chain is not None or obs_n is None:
                    pass
                elif self.is_rnn:
                    heads = eng.forward(obs_n, rnn_states=batch['rnn_states'], dones=batch.get('dones'),
                                        seq_length=self.seq_length)
                else:
                    heads = eng.forward(obs_n)
            mu, values = eng.mu_view(heads), eng.values_view(heads)
            logstd = net.sigma
        else:
            mu, logstd, values, _ = self.model.forward_heads(batch)
        mb, A = mu.shape
        mask = mask_sum = None
        if rnn_masks is not None:
            mask = rnn_masks.reshape(-1).float().contiguous()
            mask_sum = mask.sum().reshape(1)
        coef_b = self.bounds_loss_coef if self.bounds_loss_coef is not None else 0.0
        kind = 0 if self.bounds_loss_coef is None else ops.BOUND_KINDS.get(self.bound_loss_type, 0)
        if eng is not None:
            d_heads = eng.d_heads[:mb]
            d_mu, d_val = eng.mu_view(d_heads), eng.values_view(d_heads)
            mu_bias_grad, value_bias_grad = eng.net.mu.bias.grad, eng.net.value.bias.grad
        else:
            d_mu, d_val = self._d_mu[:mb], self._d_val[:mb]
            mu_bias_grad = value_bias_grad = None
        # central value without `use_experimental_cv`: the actor's own value head is not trained
        # (c_loss = zeros, a2c_continuous.py:112-115) - a zero coefficient removes it from loss and grads
        coef_c = self.critic_coef if self.has_value_loss else 0.0
        with torch.no_grad():
            loss_args = (mu.detach(), logstd.detach(),
                         values.detach().reshape(mb, -1)[:, 0] if eng is not None else values.detach().reshape(-1),
                         input_dict['actions'], input_dict['old_logp_actions'], input_dict['advantages'],
                         input_dict['old_values'].reshape(-1), input_dict['returns'].reshape(-1),
                         input_dict['mu'], input_dict['sigma'], d_mu, d_val[:, 0] if eng is not None else d_val,
                         self._loss_partials, self.e_clip, coef_c, coef_b, self.clip_value,
                         self.surrogate, kind, True, mask, mask_sum)
            fold = eng is not None and self.config.get('fold_loss_finalize', True)
            # On the fused chain the backward launch evaluates the loss of its own row tiles first (no loss
            # launch; one partial row per backward workgroup).
            in_backward = fold and eng.chain is not None and self.config.get('loss_in_backward', True)
            if in_backward:
                loss_blocks = eng.chain.num_blocks(mb, 1)
                ppo = ops.ppo_loss_desc(*loss_args)
            else:
                loss_blocks = ops.ppo_loss_blocks(mb)
                ppo = None
                ops.ppo_loss_fused(*loss_args)
            fin = (self._loss_partials, loss_blocks, A, mb, mask is not None,
                   coef_c, self.entropy_coef, coef_b, row, net.sigma.grad,
                   opt.kl_slot, mu_bias_grad, value_bias_grad)
            if fold:
                # the loss partials are folded by the weight-gradient finalise launch (one launch less),
                # and - single GPU, every gradient of the arena written by that launch - the sums of
                # squares for clip_grad_norm_ with them (another one)
                norm = None
                if self._norm_in_finalize():
                    norm = (self._fin_norm_partials, 1.0, opt.step_counter)
                nb = eng.backward(d_heads, loss_finalize=ops.loss_finalize_desc(*fin), norm=norm, ppo_loss=ppo)
                self._norm_ready = (self._fin_norm_partials, nb) if nb else None
            else:
                ops.ppo_loss_finalize(*fin)
                if eng is not None:
                    eng.backward(d_heads)
        if eng is None:
            torch.autograd.backward([mu, values], [d_mu, d_val.view(mb, 1)])

    def _native_comm(self):
        """The in-graph gradient collective, created on first use: the hipIpc all-reduce kernel
        (csrc/ipc_allreduce.hip; `native_allreduce: True`, the default), RCCL through the C ABI as a stream launch
        (csrc/rccl_wrap.hip; `native_allreduce: 'rccl'`, or as the fallback of the former with `rccl_in_graph_fallback:
        True`), or None: `native_allreduce: False` / nothing of the above could be set up - the collective then runs
        through torch.distributed between two graph replays per step."""
        if self._ipc_comm is False:
            return None
        if self._ipc_comm is None:
            self._ipc_comm = False
            want = self.config.get('native_allreduce', True)
            if self.multi_gpu and want:
                n = self.optimizer.flat_grads.numel()
                if want != 'rccl':
                    try:
                        from .ipc_allreduce import IpcAllReduce
                        self._ipc_comm = IpcAllReduce(n, self.ppo_device,
                                                      timeout_s=self.config.get('native_allreduce_timeout_s'),
                                                      two_phase=self.config.get('native_allreduce_two_phase'))
                    except Exception as e:
                        print(f'rl_games_amd: native all-reduce unavailable ({type(e).__name__}: {e}); using RCCL')
                        self._ipc_comm = False
                if not self._ipc_comm and (want == 'rccl' or self.config.get('rccl_in_graph_fallback', False)):
                    try:
                        from .rccl_allreduce import RcclAllReduce
                        self._ipc_comm = RcclAllReduce(n, self.ppo_device)
                    except Exception as e:
                        print(f'rl_games_amd: in-graph RCCL unavailable ({type(e).__name__}: {e}); using torch.distributed')
                        self._ipc_comm = False
        return self._ipc_comm or None

    def _all_reduce_grads(self):
        """Gradients + KL slot, one collective (a2c_common.py:493-509, :1559-1560)."""
        comm = self._native_comm()
        if comm is not None:
            # a plain kernel launch: capturable.  The reduced gradients pass through its registers, so it
            # also leaves the sums of squares clip_grad_norm_ needs (no grad_sumsq launch behind it).
            comm.all_reduce_sum(self.optimizer.flat_grads, norm=self._all_reduce_norm_plan(comm))
        else:
            rdist.all_reduce_sum(self.optimizer.flat_grads)
        self.last_allreduce = getattr(comm, 'kind', 'ipc') if comm is not None else 'rccl'

    def _all_reduce_norm_plan(self, comm):
        """The `norm` argument of the native all-reduce launch; also tells the optimiser launch (eager or about
        to be captured) that the gradient norm partials will be there.  Launches nothing."""
        if comm is None or not getattr(comm, 'supports_norm', True) or not self.config.get('norm_in_allreduce', True):
            return None
        opt = self.optimizer
        if self._ar_norm_partials is None:
            self._ar_norm_partials = torch.zeros(comm.norm_blocks(), dtype=torch.float64, device=self.ppo_device)
        self._norm_ready = (self._ar_norm_partials, comm.norm_blocks())
        return (self._ar_norm_partials, opt.numel, 1.0 / self.world_size, opt.step_counter)

    def trancate_gradients_and_step(self):
        """a2c_common.py:493-514 (+ the per-minibatch lr control of :1557-1563)."""
        if self.multi_gpu:
            self._all_reduce_grads()
        self._optimizer_kernels()

    def _norm_in_finalize(self):
        """The weight-gradient finalise launch may leave the gradient-norm partials (and advance the Adam
        step counter) when it writes EVERY gradient of the arena and nothing touches them before Adam."""
        ok = self._fin_norm_ok
        if ok is None:
            eng = self._engine
            ok = (not self.multi_gpu and eng is not None and eng.chain is not None and not self.is_rnn
                  and self.config.get('norm_in_finalize', True)
                  and eng.gradient_elements() == self.optimizer.numel)
            if ok:
                # one entry per finalise workgroup: ~ one per 64 weights + the bias / loss blocks (MlpDwPlan.
                # finalize_blocks); a plan that still needs more falls back to grad_sumsq (mlp_engine._weight_grads)
                entries = max(1 << 15, eng.gradient_elements() // 64 + 8192)
                self._fin_norm_partials = torch.zeros(entries, dtype=torch.float64, device=self.ppo_device)
            self._fin_norm_ok = ok
        return ok

    def _step_arguments(self):
        scale = 1.0 / self.world_size if self.multi_gpu else 1.0
        schedule = None
        if self.is_adaptive_lr and self.schedule_type == 'per_minibatch':
            schedule = self.scheduler.device_rule()
        return dict(grad_scale=scale, max_norm=self.grad_norm if self.truncate_grads else None,
                    schedule=schedule, kl_scale=scale)

    def _optimizer_kernels(self):
        opt = self.optimizer
        lean = self._lean_chain()
        # behind the in-graph all-reduce the step takes the collective's error word: a step whose gradients
        # are invalid (a peer never arrived) changes nothing
        skip = self._ipc_comm.error_word if (self.multi_gpu and self._ipc_comm) else None
        # the lean 16-row kernels read the weights as fragments in each wave's consumption order: one pack launch behind the step writes them - inside
        # the captured graphs too: no launch of this agent ever packs on demand.  (Round 4 had the Adam launch write them
        # itself on one GPU, adam_frags_kernel; that kernel family left two ranks that SHARE a GPU one exp_avg_sq update
        # apart and is gone - profiles/r5_two_rank_sync.txt.)
        opt.step(norm_ready=self._norm_ready, skip_flag=skip, pack=self._adam_pack_chain(), **self._step_arguments())
        self._norm_ready = None
        if lean is not None:
            lean.pack_frags(opt.flat_params)
            lean.mark_frags(opt.weights_token())

    def _lean_chain(self):
        """The fused chains (the MLP's, or the trunk in front of a recurrent layer) whose launches run the lean 16-row
        kernels at one of this agent's sizes (csrc/mlp_chain_lean.hip: minibatches / rollouts of < 16,384 rows on exact
        products), as one object with pack_frags / mark_frags / ensure_frags - or None."""
        c = self._lean_pack
        if c is None:
            c = False
            eng = self._engine
            # (launch kinds: the update's training forward (2) and backward (1), the rollout's inference forward (0))
            kinds = ((self.minibatch_size, 2), (self.minibatch_size, 1), (self.num_actors * self.num_agents, 0))
            chains = [ch for ch in (getattr(eng, 'chain', None), getattr(eng, 'chain_rnn', None))
                      if ch is not None and any(ch.lean_used(r, d) for r, d in kinds)]
            if chains:
                c = _LeanChains(chains)
            self._lean_pack = c
        return c or None

    def _adam_pack_chain(self):
        """The fused chain whose bf16 weight planes the Adam launch writes itself (csrc/mlp_chain_bx.hip,
        adam_pack_kernel: one launch instead of Adam + pack per optimiser step, no pack in front of the rollout
        forwards), or None: no fused chain, or none of this agent's launch sizes runs on planes."""
        c = self._adam_pack
        if c is None:
            c = False
            eng = self._engine
            chain = getattr(eng, 'chain', None) if eng is not None else None
            if chain is not None and self.config.get('adam_writes_planes', True):
                kinds = ((self.minibatch_size, 2), (self.minibatch_size, 1), (self.num_actors * self.num_agents, 0))
                if any(chain.split_products(r, d) for r, d in kinds):
                    c = chain
            self._adam_pack = c
        return c or None

    def _planes_before_replay(self):
        """A captured graph of this mode contains no pack launch: the planes must belong to the weights as they are
        (they do behind the previous step's Adam launch; not behind a restore / broadcast / set_weights)."""
        chain = self._adam_pack_chain()
        if chain is not None:
            chain.ensure_planes(self.optimizer.flat_params)
        if self._lean_chain() is not None:
            self._lean_chain().ensure_frags(self.optimizer.flat_params)

    # ------------------------------------------------------------------ HIP graphs
    def _with_fold(self, mb_index, fn, *args):
        """Run fn with the minibatch index visible to _forward_loss_backward (the in-kernel statistics
        fold needs it); a no-op wrapper when the epoch has no precomputed minibatch moments."""
        self._fold_index = mb_index if self._fold_ready else None
        try:
            return fn(*args)
        finally:
            self._fold_index = None

    def _prepare_obs_fold(self):
        """Once per epoch, after prepare_dataset: column moments of every minibatch's observations in
        two launches (instead of a moments + a merge launch in each of the epoch's optimiser steps)."""
        self._fold_ready = False
        eng = self._engine
        if (eng is None or eng.chain is None or not self.normalize_input or self.is_rnn
                or not self.config.get('fold_obs_stats', True)):
            return
        obs = self.dataset.values_dict.get('obs')
        if not torch.is_tensor(obs) or obs.dim() != 2 or not obs.is_contiguous() or obs.dtype != torch.float32:
            return
        if obs.shape[0] != len(self.dataset) * self.minibatch_size:
            return
        self.model.running_mean_std.precompute_minibatch_moments(obs, self.minibatch_size)
        self._fold_ready = True

    def _graph_signature(self):
        """Everything a captured minibatch graph bakes in: dataset storage addresses and the
        scalar hyper-parameters passed by value to the kernels."""
        vd = self.dataset.values_dict
        ptrs = tuple(vd[k].data_ptr() for k in ('obs', 'actions', 'old_logp_actions', 'advantages',
                                               'old_values', 'returns', 'mu', 'sigma', 'dones'))
        if self.is_rnn:
            ptrs += tuple(s.data_ptr() for s in vd['rnn_states'])
        s = self.scheduler
        sched = (getattr(s, 'kl_threshold', None), getattr(s, 'min_lr', None), getattr(s, 'max_lr', None),
                 getattr(s, 'lr_multiplier', None))
        return (ptrs, self.e_clip, self.critic_coef, self.entropy_coef, self.bounds_loss_coef,
                self.bound_loss_type, self.clip_value, self.surrogate, self.grad_norm,
                self.truncate_grads, self.schedule_type, self.is_adaptive_lr, sched, self.world_size,
                self._fold_ready and self.model.running_mean_std._mb_table.data_ptr())

    def _graphs_usable(self):
        return (self._hip_graphs and self._engine is not None and self._eager_epochs >= 1
                and self.dataset.values_dict.get('rnn_masks') is None and not self._graph_failed)

    def _capture(self, body):
        """Capture `body` into a HIP graph.  Capturing executes nothing on the device; host mirrors
        that the body advances (the optimiser's step count) are restored afterwards, also when the
        capture fails (GraphCaptureError: the caller then continues on the eager path)."""
        if self._graph_pool is None:
            self._graph_pool = torch.cuda.graph_pool_handle()
        g = torch.cuda.CUDAGraph()
        count0, version0 = self.optimizer.step_count, self.optimizer.weights_version
        # ... and what the chains believe their planes / fragments hold: the body's pack launches are recorded, not run
        cached0 = self._chain_cache_states()
        # No automatic garbage collection while the stream is capturing: a cycle that owns device objects of an
        # earlier graph (torch.cuda.graph collects once on entry, but the body allocates hundreds of Python objects)
        # would be finalised in the middle of the capture, and releasing device resources there aborts the process.
        gc_was_enabled = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.graph(g, pool=self._graph_pool, capture_error_mode='thread_local'):
                body()
        except Exception as e:
            raise GraphCaptureError('HIP graph capture failed') from e
        finally:
            if gc_was_enabled:
                gc.enable()
            self.optimizer.step_count, self.optimizer.weights_version = count0, version0
            self._restore_chain_cache_states(cached0)
        return g

    def _chain_cache_states(self):
        eng = self._engine
        chains = [c for c in (getattr(eng, 'chain', None), getattr(eng, 'chain_rnn', None)) if c is not None]
        return [(c, c.cache_state()) for c in chains]

    @staticmethod
    def _restore_chain_cache_states(states):
        for c, state in states:
            c.restore_cache_state(state)

    def _graph_minibatch(self, i):
        """Replay (capturing on first use) the forward/loss/backward graph of minibatch i, run the
        gradient all-reduce eagerly, then replay the (norm, Adam, lr) graph."""
        sig = self._graph_signature()
        if sig != self._graph_sig:
            self._graphs.clear()
            self._graph_opt = None
            self._graph_epoch = None
            self._graph_sig = sig
        g = self._graphs.get(i)
        if g is None:
            item = self.dataset[i]
            g = self._graphs[i] = self._capture(lambda: self._with_fold(i, self._forward_loss_backward, item,
                                                                       self._graph_rows[i]))
            self._graph_norm_state = self._norm_ready      # what the captured launches will have produced
        self._planes_before_replay()
        if self._graph_opt is None:
            # Captured BEFORE anything of this minibatch runs (a failed capture then leaves minibatch i untouched
            # for the eager path), knowing which launch in front of it produces the gradient norm - the
            # weight-gradient finalise on one GPU, the native all-reduce on several - so that the graph carries no
            # grad_sumsq of its own.
            self._norm_ready = self._graph_norm_state
            if self.multi_gpu:
                self._all_reduce_norm_plan(self._native_comm())
            self._graph_opt = self._capture(self._optimizer_kernels)
        g.replay()
        if self.multi_gpu:
            self._all_reduce_grads()
        self._norm_ready = None
        self._graph_opt.replay()
        self.optimizer.step_done()
        if self._adam_pack_chain() is not None:
            self._adam_pack_chain().mark_planes(self.optimizer.weights_token())
        if self._lean_chain() is not None:
            self._lean_chain().mark_frags(self.optimizer.weights_token())

    def _graph_mini_epoch(self, nmb):
        """Single-GPU runs with nothing to do on the host between minibatches (device-side or
        per-epoch lr schedule): ALL minibatches of a mini-epoch - forward, loss, backward, clip, Adam,
        lr update, nmb times - are one HIP graph, replayed once per mini-epoch."""
        sig = self._graph_signature()
        if sig != self._graph_sig:
            self._graphs.clear()
            self._graph_opt = None
            self._graph_epoch = None
            self._graph_sig = sig
        if self._graph_epoch is None:
            def body():
                for i in range(nmb):
                    self._with_fold(i, self._forward_loss_backward, self.dataset[i], self._graph_rows[i])
                    if self.multi_gpu:
                        self._all_reduce_grads()          # native in-graph collective only (see train_epoch)
                    self._optimizer_kernels()
                if self._fold_ready:
                    self.model.running_mean_std.fold_sync(nmb)
            self._graph_epoch = self._capture(body)
        self._planes_before_replay()
        self._graph_epoch.replay()
        self.optimizer.step_count += nmb
        self.optimizer.weights_version += nmb
        if self._adam_pack_chain() is not None:
            self._adam_pack_chain().mark_planes(self.optimizer.weights_token())
        if self._lean_chain() is not None:
            self._lean_chain().mark_frags(self.optimizer.weights_token())

    def _host_schedule(self, kl_value):
        lr, self.entropy_coef = self.scheduler.update(self._host_lr, self.entropy_coef, self.epoch_num,
                                                      self.frame, kl_value)
        if lr != self._host_lr:
            self.update_lr(lr)

    def train_epoch(self):
        """a2c_common.py:1517-1584."""
        self.vec_env.set_train_info(self.frame, self)
        self.set_eval()
        play_time_start = time.perf_counter()
        with torch.no_grad():
            batch_dict = self.play_steps_rnn() if self.is_rnn else self.play_steps()
        play_time_end = time.perf_counter()
        update_time_start = time.perf_counter()
        self.set_train()
        self.curr_frames = batch_dict.pop('played_frames')
        self.prepare_dataset(batch_dict)
        self.algo_observer.after_steps()
        self._prepare_obs_fold()

        a_losses, c_losses, b_losses, entropies, kls = [], [], [], [], []
        if self.has_central_value:                                   # a2c_common.py:1536-1537
            self.train_central_value()
            self.set_train()
        self._mb_index = 0
        device_schedule = self.is_adaptive_lr and self.schedule_type == 'per_minibatch'
        last_lr, lr_mul = self.last_lr, 1.0
        use_graphs = self._graphs_usable() and (device_schedule or self.schedule_type != 'per_minibatch'
                                                or not self.is_adaptive_lr)
        nmb = len(self.dataset)
        for mini_ep in range(self.mini_epochs_num):
            first = self._mb_index
            done = 0                     # minibatches of this mini-epoch already stepped
            fold_synced = False
            if use_graphs:
                try:
                    self.set_train()
                    host_between = self.schedule_type == 'per_minibatch' and not device_schedule
                    in_graph_comm = (not self.multi_gpu) or self._native_comm() is not None
                    if in_graph_comm and not host_between and self.config.get('mini_epoch_graph', True):
                        self._graph_mini_epoch(nmb)      # (the statistics hand-back is part of the graph)
                        done = nmb
                        fold_synced = True
                    else:
                        for i in range(nmb):
                            self._graph_minibatch(i)
                            done = i + 1
                            if host_between:
                                self._host_schedule(None)
                except GraphCaptureError as e:
                    # A capture executes nothing and the host mirrors were rolled back, so the
                    # remaining minibatches of this mini-epoch (and all later ones) run eagerly.
                    print(f'rl_games_amd: HIP graph capture failed ({e.__cause__!r}); continuing eagerly')
                    self._graph_failed = True
                    self._graphs.clear()
                    self._graph_opt = self._graph_epoch = None
                    use_graphs = False
                self._mb_scalars[first:first + done].copy_(self._graph_rows[:done])
                self._mb_index += done
                for i in range(done):
                    row = self._mb_scalars[first + i]
                    a_losses.append(row[0])
                    c_losses.append(row[1])
                    entropies.append(row[2])
                    if self.bounds_loss_coef is not None:
                        b_losses.append(row[3])
            if done < nmb:
                if self.is_discrete and done == 0:    # a2c_common.py:1263 (no-op unless `permute`)
                    self.dataset.apply_permutation()
                for i in range(done, nmb):
                    res = self._with_fold(i, self.train_actor_critic, self.dataset[i])
                    a_loss, c_loss, entropy, kl, last_lr, lr_mul = res[:6]
                    b_loss = res[8] if len(res) > 8 else None
                    a_losses.append(a_loss)
                    c_losses.append(c_loss)
                    entropies.append(entropy)
                    if self.bounds_loss_coef is not None:
                        b_losses.append(b_loss)
                    if self.schedule_type == 'per_minibatch' and not device_schedule:
                        self._host_schedule(None if not self.is_adaptive_lr else float(kl.item()))
            if self._fold_ready and not fold_synced:
                self.model.running_mean_std.fold_sync(nmb)
            av_kls = self._mb_scalars[first:self._mb_index, 4].mean()
            if self.multi_gpu:
                rdist.all_reduce_sum(av_kls)
                av_kls /= self.world_size
            if self.schedule_type == 'standard':
                self._host_schedule(float(av_kls.item()))
            kls.append(av_kls)
            if self.normalize_input:
                self.model.running_mean_std.eval()
        if self.schedule_type == 'standard_epoch':
            self._host_schedule(float(torch.stack(kls).mean().item()))
        self._fold_ready = False
        self.sync_running_stats()
        self._check_ranks_in_sync()
        if self._ipc_comm:
            # one host read per epoch; the verdict is collective, so that every rank raises in the same epoch
            # instead of one rank leaving the others to hang in their next collective
            _, timed_out = self._ipc_comm.status()
            flag = torch.tensor([float(timed_out != 0)], device=self.ppo_device)
            import torch.distributed as dist
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if flag.item() != 0:
                raise RuntimeError(f'in-graph all-reduce: a launch gave up waiting for a peer rank (this rank: '
                                   f'launch {timed_out or "none"}); the optimiser steps behind it were skipped on the '
                                   f'rank that gave up - parameters may differ between ranks, aborting on all ranks. '
                                   f'Raise native_allreduce_timeout_s / RLG_IPC_TIMEOUT_S for legitimately long rank '
                                   f'skews, or set native_allreduce: False to use RCCL.')
        self._eager_epochs += 0 if use_graphs else 1
        self._check_split_range()
        if device_schedule:
            # one host read per epoch: [lr the last minibatch was stepped with, lr for the next one]
            used, nxt = self.optimizer.last_and_next_lr()
            last_lr = used                  # what the reference returns (train_result[4])
            self.last_lr = self._host_lr = nxt
        else:
            last_lr = self._host_lr
        update_time_end = time.perf_counter()
        play_time = play_time_end - play_time_start
        update_time = update_time_end - update_time_start
        total_time = update_time_end - play_time_start
        return (batch_dict['step_time'], play_time, update_time, total_time, a_losses, c_losses, b_losses,
                entropies, kls, last_lr, lr_mul)

    def _check_split_range(self):
        """Split-fp16 chain launches (csrc/bx_form.hpp) scale weights and hidden activations by FIXED powers of two: a
        hidden activation of 4,094 or a weight of 1,023 and more becomes Inf in its plane and NaN in everything computed
        from it - never a finite wrong value, but the reason deserves a sentence (a warning, once: the reference would train on
        with NaNs as well).  One host read per epoch."""
        chain = getattr(self._engine, 'chain', None) if self._engine is not None else None
        if chain is None or ops.chain_split_form()[1] != 'fp16' or self._mb_index == 0:
            return
        n = min(self._mb_index, self._mb_scalars.shape[0])
        if not getattr(self, '_split_range_warned', False) and \
                not bool(torch.isfinite(self._mb_scalars[:n, :5]).all().item()):
            self._split_range_warned = True
            import warnings
            warnings.warn(
                'rl_games_amd: non-finite losses in this epoch.  If the inputs are finite: the split-fp16 chain kernels hold '
                '|hidden activation| < 4094 and |weight| < 1023 (fixed operand scales, csrc/bx_form.hpp); a network '
                'beyond that runs on exact fp32 products with RLG_CHAIN_BX=0 RLG_DW_BF16=0.', RuntimeWarning)

    # ================================================================== multi-GPU stats
    def _stats_sync_modules(self):
        mods = []
        if self.normalize_input and hasattr(self.model, 'running_mean_std'):
            mods.append(self.model.running_mean_std)
        if self.normalize_value and getattr(self.model, 'value_mean_std', None) is not None:
            mods.append(self.model.value_mean_std)
        if self.has_central_value:                                   # a2c_common.py:759-765
            cv_model = self.central_value_net.model
            if getattr(cv_model, 'running_mean_std', None) is not None:
                mods.append(cv_model.running_mean_std)
            if getattr(cv_model, 'value_mean_std', None) is not None:
                mods.append(cv_model.value_mean_std)
        return mods

    def _stats_sync(self):
        """One StatsSync over every normaliser of the agent (actor + central value): an epoch's exchange is
        two launches and ONE collective (csrc/running_stats.hip, distributed.StatsSync)."""
        mods = self._stats_sync_modules()
        if not mods:
            return None
        sync = getattr(self, '_stats_sync_obj', None)
        if sync is None or not sync.covers(mods):
            sync = self._stats_sync_obj = rdist.StatsSync(mods)
        return sync

    def _seed_stats_sync_snapshots(self):
        """a2c_common.py:766-780: restored statistics are shared history, not fresh per-rank data."""
        if not self.multi_gpu or not self.multi_gpu_sync_stats or self.multi_gpu_sync_stats_mode == 'broadcast':
            return
        sync = self._stats_sync()
        if sync is not None:
            sync.seed()

    def _check_ranks_in_sync(self):
        """Guard against silent rank drift (round 5: two ranks ended epochs with `exp_avg_sq` one update apart in 16
        elements, profiles/r5_two_rank_sync.txt - found by bench.py's probe only): an exact checksum of parameters and
        Adam moments, one tiny all-reduce per epoch."""
        if not self.multi_gpu or not self.multi_gpu_param_check:
            return
        o = self.optimizer
        if rdist.ranks_in_sync([o.flat_params, o.exp_avg, o.exp_avg_sq]):
            return
        if self.multi_gpu_param_check == 'rebroadcast':
            print(f'rl_games_amd: rank {self.global_rank}: parameters / Adam moments differ between the ranks after epoch '
                  f'{self.epoch_num}; taking rank 0\'s', flush=True)
            import torch.distributed as dist
            for t in (o.flat_params, o.exp_avg, o.exp_avg_sq):
                dist.broadcast(t, 0)
            o.weights_changed()
            return
        raise RuntimeError(f'rank {self.global_rank}: parameters / Adam moments are no longer bit-identical across the ranks '
                           f'(epoch {self.epoch_num}); multi_gpu_param_check: \'rebroadcast\' re-aligns them instead of raising')

    def sync_running_stats(self):
        """a2c_common.py:782-808."""
        if not self.multi_gpu or not self.multi_gpu_sync_stats:
            return
        sync = self._stats_sync()
        if sync is None:
            return
        if self.multi_gpu_sync_stats_mode == 'broadcast':
            sync.adopt_rank0(rdist.broadcast_from_rank0)
        else:
            sync.merge(rdist.all_reduce_sum)

    # ================================================================== weights / checkpoints
    def get_stats_weights(self, model_stats=False):
        state = {}
        if self.normalize_rms_advantage:
            state['advantage_mean_std'] = self.advantage_mean_std.state_dict()
        if self.has_central_value:
            state['central_val_stats'] = self.central_value_net.get_stats_weights(model_stats)
        if model_stats:
            if self.normalize_input:
                state['running_mean_std'] = self.model.running_mean_std.state_dict()
            if self.normalize_value:
                state['reward_mean_std'] = self.model.value_mean_std.state_dict()
        return state

    def set_stats_weights(self, weights):
        if self.normalize_rms_advantage and 'advantage_mean_std' in weights:
            self.advantage_mean_std.load_state_dict(weights['advantage_mean_std'])
        if self.normalize_input and 'running_mean_std' in weights:
            self.model.running_mean_std.load_state_dict(weights['running_mean_std'])
        if self.normalize_value and 'reward_mean_std' in weights:
            self.model.value_mean_std.load_state_dict(weights['reward_mean_std'])

    def _plain_model(self):
        """The policy module itself.  The reference Runner re-assigns `agent.model = torch.compile(agent.model)` unless the
        config says `torch_compile: False` (torch_runner.py:283-307); nothing of this agent calls the wrapper's forward,
        and checkpoints are written / read in the plain key format whichever object `self.model` is."""
        return getattr(self.model, '_orig_mod', self.model)

    def get_weights(self):
        state = self.get_stats_weights()
        state['model'] = self._plain_model().state_dict()
        return state

    def set_weights(self, weights):
        model_state = {k.replace('_orig_mod.', ''): v for k, v in weights['model'].items()}
        self._plain_model().load_state_dict(model_state)      # copy_ into the arena views: params stay flat
        self.optimizer.weights_changed()
        self.set_stats_weights(weights)
        self._seed_stats_sync_snapshots()

    def get_full_state_weights(self):
        state = self.get_weights()
        state['epoch'] = self.epoch_num
        state['frame'] = self.frame
        state['optimizer'] = self.optimizer.state_dict()
        if self.has_central_value:
            state['assymetric_vf_nets'] = self.central_value_net.state_dict()
            state['assymetric_vf_optimizer'] = self.central_value_net.optimizer.state_dict()
        state['last_mean_rewards'] = self.last_mean_rewards
        if self.vec_env is not None:
            state['env_state'] = self.vec_env.get_env_state()
        manifest = self.config.get('capability_manifest')
        if manifest is not None:
            state['capability_manifest'] = manifest
        return state

    def set_full_state_weights(self, weights, set_epoch=True):
        self.set_weights(weights)
        if set_epoch:
            self.epoch_num = weights['epoch']
            self.frame = weights['frame']
        if self.has_central_value:
            self.set_central_value_function_weights(weights)
            if 'assymetric_vf_optimizer' in weights:
                self.central_value_net.optimizer.load_state_dict(weights['assymetric_vf_optimizer'])
        self.optimizer.load_state_dict(weights['optimizer'])
        self._host_lr = self.last_lr = self.optimizer.param_groups[0]['lr']
        self.last_mean_rewards = weights.get('last_mean_rewards', -float('inf'))
        if self.vec_env is not None:
            self.vec_env.set_env_state(weights.get('env_state', None))
        if 'capability_manifest' in weights and self.config.get('capability_manifest') is None:
            self.config['capability_manifest'] = weights['capability_manifest']
        self._seed_stats_sync_snapshots()

    def save(self, fn):
        state = self.get_full_state_weights()
        path = fn if fn.endswith('.pth') else fn + '.pth'
        torch.save(state, path)                              # torch_ext.save_checkpoint :73-92
        return path

    def restore(self, fn, set_epoch=True):
        checkpoint = torch.load(fn, map_location=self.ppo_device, weights_only=False)
        self.set_full_state_weights(checkpoint, set_epoch=set_epoch)

    def set_central_value_function_weights(self, weights):
        state = {k.replace('_orig_mod.', ''): v for k, v in weights['assymetric_vf_nets'].items()}
        self.central_value_net.load_state_dict(state)
        self._seed_stats_sync_snapshots()

    def restore_central_value_function(self, fn):
        checkpoint = torch.load(fn, map_location=self.ppo_device, weights_only=False)
        self.set_central_value_function_weights(checkpoint)

    def get_param(self, param_name):
        if param_name in ('grad_norm', 'critic_coef', 'bounds_loss_coef', 'entropy_coef', 'kl_threshold',
                          'gamma', 'tau', 'mini_epochs_num', 'e_clip'):
            return getattr(self, param_name)
        if param_name == 'learning_rate':
            return self._sync_lr_to_host()
        raise NotImplementedError(f"Can't get param {param_name}")

    def set_param(self, param_name, param_value):
        if param_name in ('grad_norm', 'critic_coef', 'bounds_loss_coef', 'entropy_coef', 'gamma', 'tau',
                          'mini_epochs_num', 'e_clip'):
            setattr(self, param_name, param_value)
        elif param_name == 'learning_rate':
            if self.is_adaptive_lr:
                raise NotImplementedError("Can't directly mutate LR on this schedule")
            self.update_lr(param_value)
        elif param_name == 'kl_threshold':
            if not self.is_adaptive_lr:
                raise NotImplementedError("Can't directly mutate kl threshold")
            self.kl_threshold = param_value
            self.scheduler.kl_threshold = param_value
        else:
            raise NotImplementedError(f'No param found for {param_value}')

    def broadcast_parameters(self):
        """Rank 0's parameters and normaliser state to every rank (a2c_common.py:1670-1680, C2) -
        one broadcast of the flat arena instead of a pickled state_dict."""
        if not self.multi_gpu:
            return
        import torch.distributed as dist
        dist.broadcast(self.optimizer.flat_params, 0)
        self.optimizer.weights_changed()
        if self.has_central_value:
            dist.broadcast(self.central_value_net.optimizer.flat_params, 0)
            # (c10d collectives write in place WITHOUT bumping the tensor's version counter: the chain's planes /
            #  fragments of the pre-broadcast weights have to be invalidated by hand, like the actor's above)
            self.central_value_net.optimizer.weights_changed()
        sync = self._stats_sync()
        if sync is not None:
            sync.adopt_rank0(rdist.broadcast_from_rank0)
        self._seed_stats_sync_snapshots()

    # ================================================================== training loop
    def train(self):
        """a2c_common.py:1662-1782.  Returns (last_mean_rewards, epoch_num)."""
        self._ensure_dirs()
        self.init_tensors()
        total_time = 0
        self.obs = self.env_reset()
        self.curr_frames = self.batch_size_envs
        self.broadcast_parameters()
        while True:
            epoch_num = self.update_epoch()
            res = self.train_epoch()
            if len(res) == 10:                     # discrete agents report no bound losses
                res = res[:6] + ([],) + res[6:]
            (step_time, play_time, update_time, sum_time, a_losses, c_losses, b_losses, entropies, kls,
             last_lr, lr_mul) = res
            total_time += sum_time
            curr_frames = self.curr_frames * self.world_size if self.multi_gpu else self.curr_frames
            self.frame += curr_frames
            frame = self.frame // self.num_agents
            self.dataset.update_values_dict(None)
            should_exit = False
            if self.global_rank == 0:
                if self.print_stats:
                    fps_step = curr_frames / max(step_time, 1e-9)
                    fps_inf = curr_frames / (self.num_agents * play_time)
                    fps_total = curr_frames / (self.num_agents * sum_time)
                    print(f'fps step: {fps_step:.0f} fps step and policy inference: {fps_inf:.0f} '
                          f'fps total: {fps_total:.0f} epoch: {epoch_num:.0f}/{self.max_epochs:.0f} '
                          f'frames: {frame:.0f}/{self.max_frames:.0f}')
                w = self.writer
                w.add_scalar('performance/step_inference_rl_update_fps', curr_frames / sum_time, frame)
                w.add_scalar('performance/step_inference_fps', curr_frames / play_time, frame)
                w.add_scalar('performance/rl_update_time', update_time, frame)
                w.add_scalar('performance/step_inference_time', play_time, frame)
                if not isinstance(w, NullWriter):
                    w.add_scalar('losses/a_loss', torch.stack(a_losses).mean().item(), frame)
                    w.add_scalar('losses/c_loss', torch.stack(c_losses).mean().item(), frame)
                    w.add_scalar('losses/entropy', torch.stack(entropies).mean().item(), frame)
                    w.add_scalar('info/kl', torch.stack(kls).mean().item(), frame)
                    if len(b_losses) > 0:
                        w.add_scalar('losses/bounds_loss', torch.stack(b_losses).mean().item(), frame)
                w.add_scalar('info/last_lr', last_lr * lr_mul, frame)
                w.add_scalar('info/e_clip', self.e_clip * lr_mul, frame)
                w.add_scalar('info/epochs', epoch_num, frame)
                self.algo_observer.after_print_stats(frame, epoch_num, total_time)
                mean_rewards = -np.inf
                if self.game_rewards.current_size > 0:
                    mean_rewards = self.game_rewards.get_mean()
                    mean_lengths = self.game_lengths.get_mean()
                    mean_rewards = np.atleast_1d(mean_rewards)
                    self.mean_rewards = mean_rewards[0]
                    w.add_scalar('rewards/step', mean_rewards[0], frame)
                    w.add_scalar('rewards/iter', mean_rewards[0], epoch_num)
                    w.add_scalar('episode_lengths/step', float(np.atleast_1d(mean_lengths)[0]), frame)
                    checkpoint_name = self.config['name'] + '_ep_' + str(epoch_num) + '_rew_' + str(mean_rewards[0])
                    if self.save_freq > 0 and epoch_num % self.save_freq == 0:
                        self.save(os.path.join(self.nn_dir, 'last_' + checkpoint_name))
                    if mean_rewards[0] > self.last_mean_rewards and epoch_num >= self.save_best_after:
                        self.last_mean_rewards = mean_rewards[0]
                        self.save(os.path.join(self.nn_dir, self.config['name']))
                        if 'score_to_win' in self.config and self.last_mean_rewards > self.config['score_to_win']:
                            self.save(os.path.join(self.nn_dir, checkpoint_name))
                            should_exit = True
                if epoch_num >= self.max_epochs and self.max_epochs != -1:
                    self.save(os.path.join(self.nn_dir, 'last_' + self.config['name'] + '_ep_' + str(epoch_num)))
                    should_exit = True
                if self.frame >= self.max_frames and self.max_frames != -1:
                    self.save(os.path.join(self.nn_dir, 'last_' + self.config['name'] + '_frame_' + str(self.frame)))
                    should_exit = True
                if not should_exit and self.stop_fn is not None and self.stop_fn(self):
                    self.save(os.path.join(self.nn_dir, 'last_' + self.config['name'] + '_custom_stop_ep_'
                                           + str(epoch_num)))
                    should_exit = True
            if self.multi_gpu:
                import torch.distributed as dist
                flag = torch.tensor(float(should_exit), device=self.ppo_device)
                dist.broadcast(flag, 0)                                                   # C8
                should_exit = bool(flag.item())
            if should_exit:
                return self.last_mean_rewards, epoch_num
