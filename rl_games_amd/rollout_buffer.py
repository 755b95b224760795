"""ExperienceBuffer on env-major HBM storage.

Same constructor, attributes and methods as the reference's
`rl_games.common.experience.ExperienceBuffer` (rl_games/common/experience.py:326-522):
`tensor_dict[name]` has the reference's logical shape `[horizon, num_envs*agents, ...]` and
dtypes (numpy->torch map incl. float64->float32, rl_games/algos_torch/torch_ext.py:11-23), and
`update_data(name, index, val)` / `get_transformed_list(op, names)` behave identically.

What differs is the *physical* layout: every field is stored `[num_envs, horizon, ...]`
(flat index env*H + t) and `tensor_dict` exposes a transposed view.  `swap_and_flatten01`
(rl_games/common/a2c_common.py:33-40) applied to such a view is a zero-copy reshape, which
removes the ~2.9 GB/epoch transpose copy of the reference at 65,536 x 32, and GAE reads each
env's trajectory as one contiguous row.  `store_step` writes all per-step fields with one
HIP launch (csrc/experience.hip) instead of one strided copy kernel per field.
"""
import numpy as np
import torch

from . import ops
from .spaces import Box

_NP_TO_TORCH = {
    np.dtype('bool'): torch.bool, np.dtype('uint8'): torch.uint8, np.dtype('int8'): torch.int8,
    np.dtype('int16'): torch.int16, np.dtype('int32'): torch.int32, np.dtype('int64'): torch.int64,
    np.dtype('float16'): torch.float16, np.dtype('float32'): torch.float32,
    np.dtype('float64'): torch.float32,   # sic: the reference stores float64 spaces as fp32
}


class ExperienceBuffer:
    def __init__(self, env_info, algo_info, device, aux_tensor_dict=None):
        self.env_info = env_info
        self.algo_info = algo_info
        self.device = device
        self.num_agents = env_info.get('agents', 1)
        self.action_space = env_info['action_space']
        self.num_actors = algo_info['num_actors']
        self.horizon_length = algo_info['horizon_length']
        self.has_central_value = algo_info['has_central_value']
        self.use_action_masks = algo_info.get('use_action_masks', False)
        self.is_discrete = self.is_multi_discrete = self.is_continuous = False
        self.obs_base_shape = (self.horizon_length, self.num_agents * self.num_actors)
        self.state_base_shape = (self.horizon_length, self.num_actors)

        kind = type(self.action_space).__name__
        if kind == 'Discrete':
            self.actions_shape = ()
            self.actions_num = self.action_space.n
            self.is_discrete = True
        elif kind == 'Tuple':
            self.actions_shape = (len(self.action_space),)
            self.actions_num = [a.n for a in self.action_space]
            self.is_multi_discrete = True
        elif kind == 'Box':
            self.actions_shape = (self.action_space.shape[0],)
            self.actions_num = self.action_space.shape[0]
            self.is_continuous = True

        self.storage = {}       # name -> physical [N, H, ...] tensor (or dict of them)
        self.tensor_dict = {}   # name -> [H, N, ...] view (the reference's API)
        self._init_from_env_info(env_info)
        self.aux_tensor_dict = aux_tensor_dict
        if aux_tensor_dict is not None:
            for k, shape in aux_tensor_dict.items():
                self._add(k, Box(0, 1, shape=tuple(shape), dtype=np.float32), self.obs_base_shape)

    # ------------------------------------------------------------------ allocation
    def _alloc(self, space, base_shape):
        if space is None:
            raise ValueError('Space is None while allocating tensors. Ensure env_info provides required spaces.')
        kind = type(space).__name__
        horizon, rows = base_shape
        if kind == 'Dict':
            phys, view = {}, {}
            for k, sub in space.spaces.items():
                phys[k], view[k] = self._alloc(sub, base_shape)
            return phys, view
        if kind == 'Box':
            tail = tuple(space.shape)
        elif kind == 'Discrete':
            tail = ()
        elif kind == 'Tuple':
            tail = (len(space),)
        else:
            raise ValueError(f'Unsupported space type: {kind} ({type(space)}). Expected Box, Discrete, Tuple, or Dict.')
        dtype = _NP_TO_TORCH[np.dtype(space.dtype)]
        phys = torch.zeros((rows, horizon) + tail, dtype=dtype, device=self.device)
        return phys, phys.transpose(0, 1)

    def _add(self, name, space, base_shape):
        self.storage[name], self.tensor_dict[name] = self._alloc(space, base_shape)

    def _init_from_env_info(self, env_info):
        obs_base, state_base = self.obs_base_shape, self.state_base_shape
        self._add('obses', env_info['observation_space'], obs_base)
        if self.has_central_value:
            state_space = env_info.get('state_space') or env_info.get('observation_space')
            self._add('states', state_space, state_base)
        val_space = Box(0, 1, shape=(env_info.get('value_size', 1),), dtype=np.float32)
        self._add('rewards', val_space, obs_base)
        self._add('values', val_space, obs_base)
        self._add('neglogpacs', Box(0, 1, shape=(), dtype=np.float32), obs_base)
        self._add('dones', Box(0, 1, shape=(), dtype=np.uint8), obs_base)
        if self.is_discrete or self.is_multi_discrete:
            self._add('actions', Box(0, 1, shape=self.actions_shape, dtype=np.int64), obs_base)
        if self.use_action_masks:
            self._add('action_masks', Box(0, 1, shape=(int(np.sum(self.actions_num)),), dtype=np.bool_), obs_base)
        if self.is_continuous:
            for k in ('actions', 'mus', 'sigmas'):
                self._add(k, Box(0, 1, shape=self.actions_shape, dtype=np.float32), obs_base)

    # ------------------------------------------------------------------ reference API
    def update_data(self, name, index, val):
        stored = self.tensor_dict[name]
        if isinstance(stored, dict):
            if not isinstance(val, dict):
                raise ValueError(f"Expected dict value for '{name}' but got {type(val)}")
            for k, v in val.items():
                stored[k][index, :] = v
        else:
            if isinstance(val, dict):
                raise ValueError(f"Expected tensor value for '{name}' but got dict")
            stored[index, :] = val

    def update_data_rnn(self, name, indices, play_mask, val):
        if not isinstance(val, (dict, torch.Tensor)):
            raise TypeError(f'Expected dict or tensor, got {type(val)}')
        stored = self.tensor_dict[name]
        if isinstance(stored, dict):
            if not isinstance(val, dict):
                raise ValueError(f"Expected dict value for '{name}' but got {type(val)}")
            for k, v in val.items():
                if k not in stored:
                    raise KeyError(f'Key {k} not found in tensor_dict[{name}]')
                stored[k][indices, play_mask] = v
        else:
            if isinstance(val, dict):
                raise ValueError(f"Expected tensor value for '{name}' but got dict")
            stored[indices, play_mask] = val

    def get_transformed(self, transform_op):
        return self.get_transformed_list(transform_op, list(self.tensor_dict))

    def get_transformed_list(self, transform_op, tensor_list):
        out = {}
        for k in tensor_list:
            v = self.tensor_dict.get(k)
            if v is None:
                continue
            out[k] = {kd: transform_op(vd) for kd, vd in v.items()} if isinstance(v, dict) else transform_op(v)
        return out

    # ------------------------------------------------------------------ fused fast path
    def store_step(self, index, fields):
        """`fields`: dict name -> value for step `index` (tensor, or dict for Dict obs).
        One launch for all fields; identical result to calling update_data per field."""
        pairs = []
        for name, val in fields.items():
            phys = self.storage[name]
            if isinstance(phys, dict):
                for k, v in val.items():
                    pairs.append(self._pair(v, phys[k], f'{name}.{k}'))
            else:
                pairs.append(self._pair(val, phys, name))
        # one launch per row count: the privileged states of a multi-agent env have one row per env, everything else one
        # per agent (experience.py:346,380-383)
        by_rows = {}
        for pair in pairs:
            by_rows.setdefault(pair[1].shape[0], []).append(pair)
        for rows, group in by_rows.items():
            ops.rollout_store_step(group, rows, self.horizon_length, index)

    @staticmethod
    def _pair(val, phys, name):
        if val.dtype != phys.dtype:
            val = val.to(phys.dtype)
        if val.dim() == phys.dim() - 2 and phys.dim() >= 3:
            val = val.unsqueeze(-1)         # e.g. values [N] into a [N, H, 1] field
        if not val.is_contiguous():
            val = val.contiguous()
        return val, phys

    def flat(self, name):
        """Env-major flat view [N*H, ...] of a field: what swap_and_flatten01 returns."""
        phys = self.storage[name]
        if isinstance(phys, dict):
            return {k: v.reshape((-1,) + tuple(v.shape[2:])) for k, v in phys.items()}
        return phys.reshape((-1,) + tuple(phys.shape[2:]))
