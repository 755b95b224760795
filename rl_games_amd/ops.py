"""Thin, typed wrappers over the C ABI (include/rlg_hip.h): one Python function per entry
point, taking torch CUDA tensors, launching on torch's current stream.  No arithmetic
happens here - shapes/dtypes are checked and raw pointers are handed to librlg_hip.so.
"""
import ctypes

import numpy as np
import os

import torch

from . import _lib

F32 = torch.float32
F64 = torch.float64
U8 = torch.uint8


def _need(t, dtype, name, contiguous=True):
    if t is None:
        raise ValueError(f'{name} is None')
    _lib.require_gpu(t, name)
    if t.dtype != dtype:
        raise ValueError(f'{name}: expected {dtype}, got {t.dtype}')
    if contiguous and not t.is_contiguous():
        raise ValueError(f'{name}: expected a contiguous tensor, got strides {t.stride()}')
    return t.data_ptr()


def _opt(t, dtype, name):
    return None if t is None else _need(t, dtype, name)


def _stream(t):
    return _lib.stream_handle(t.device)


# ------------------------------------------------------------------ rollout buffer

def rollout_store_step(pairs, num_envs, horizon, step):
    """pairs: list of (src [N, ...] contiguous, dst env-major storage [N, H, ...] contiguous).
    Equivalent to `dst_view[step, :] = src` for every pair (experience.py:433-456)."""
    lib = _lib.load()
    n = len(pairs)
    srcs = (ctypes.c_void_p * n)()
    dsts = (ctypes.c_void_p * n)()
    rows = (ctypes.c_int * n)()
    for k, (src, dst) in enumerate(pairs):
        _lib.require_gpu(src, 'rollout_store_step src')
        if not src.is_contiguous() or not dst.is_contiguous():
            raise ValueError('rollout_store_step needs contiguous src and env-major dst')
        if src.dtype != dst.dtype:
            raise ValueError(f'dtype mismatch {src.dtype} vs {dst.dtype}')
        row_bytes = (src.numel() // num_envs) * src.element_size()
        if src.shape[0] != num_envs or dst.numel() * dst.element_size() != num_envs * horizon * row_bytes:
            raise ValueError(f'shape mismatch: src {tuple(src.shape)} dst {tuple(dst.shape)}')
        srcs[k] = src.data_ptr()
        dsts[k] = dst.data_ptr()
        rows[k] = row_bytes
    _lib.check(lib.rlg_rollout_store_step(n, srcs, dsts, rows, num_envs, horizon, step,
                                          _stream(pairs[0][0])), 'rlg_rollout_store_step')


def post_step_num_blocks(num_envs):
    return _lib.load().rlg_rollout_post_step_num_blocks(int(num_envs))


def rollout_post_step(rewards, dones, time_outs, values, live_rows, rewards_buf, cur_rewards,
                      cur_shaped, cur_lengths, ep_partials, shaper, bootstrap, gamma, horizon, step,
                      num_agents=1):
    """shaper = (shift, scale, min_val, max_val[, log_val])."""
    lib = _lib.load()
    N, V = rewards.shape
    shift, scale, rmin, rmax = shaper[:4]
    clamp = 0 if (rmin == -np.inf and rmax == np.inf) else 1
    if len(shaper) > 4 and shaper[4]:
        clamp |= 2
    kind, to_ptr = 0, None
    if time_outs is not None:
        if time_outs.dtype == F32:
            kind, to_ptr = 2, _need(time_outs, F32, 'time_outs')
        elif time_outs.dtype in (U8, torch.bool):
            kind, to_ptr = 1, (time_outs.view(U8) if time_outs.dtype == torch.bool else time_outs).data_ptr()
            _lib.require_gpu(time_outs, 'time_outs')
        else:
            raise ValueError(f'time_outs dtype {time_outs.dtype} unsupported')
    _lib.check(lib.rlg_rollout_post_step(
        _need(rewards, F32, 'rewards'), _need(dones, U8, 'dones'), to_ptr, kind,
        _need(values, F32, 'values'), _opt(live_rows, F32, 'live_rows'),
        _need(rewards_buf, F32, 'rewards_buf'), _need(cur_rewards, F32, 'cur_rewards'),
        _need(cur_shaped, F32, 'cur_shaped'), _need(cur_lengths, F32, 'cur_lengths'),
        _need(ep_partials, F64, 'ep_partials'), float(np.float32(shift)), float(np.float32(scale)),
        float(np.float32(max(rmin, -3.0e38))), float(np.float32(min(rmax, 3.0e38))), clamp,
        1 if bootstrap else 0, float(np.float32(gamma)), N, horizon, V, step, int(num_agents),
        _stream(rewards)),
        'rlg_rollout_post_step')


def episode_meters_update(ep_partials, horizon, num_blocks, value_size, max_size, mean_rewards,
                          mean_shaped, mean_lengths, current_sizes, finished_total):
    lib = _lib.load()
    _lib.check(lib.rlg_episode_meters_update(
        _need(ep_partials, F64, 'ep_partials'), horizon, num_blocks, value_size, max_size,
        _need(mean_rewards, F32, 'mean_rewards'), _need(mean_shaped, F32, 'mean_shaped'),
        _need(mean_lengths, F32, 'mean_lengths'), _need(current_sizes, torch.int32, 'current_sizes'),
        _need(finished_total, torch.int64, 'finished_total'), _stream(ep_partials)),
        'rlg_episode_meters_update')


def rollout_policy_head(heads, logstd, noise, value_stats, eps, actions_out, values_out, storage, horizon,
                        step, env_actions=None):
    """storage: ExperienceBuffer.storage (env-major fields).  value_stats = (mean, var) or None.
    env_actions = (out [N, A], low [A], high [A]): also rescale_actions(low, high, clamp(actions, -1, 1))."""
    lib = _lib.load()
    N, A = noise.shape
    vm = vv = None
    if value_stats is not None:
        vm, vv = _need(value_stats[0], F64, 'value mean'), _need(value_stats[1], F64, 'value var')
    _lib.check(lib.rlg_rollout_policy_head(
        _need(heads, F32, 'heads'), heads.stride(0), _need(logstd, F32, 'logstd'), _need(noise, F32, 'noise'),
        vm, vv, float(np.float32(eps)), _need(actions_out, F32, 'actions_out'),
        _need(values_out, F32, 'values_out'), _need(storage['actions'], F32, 'actions'),
        _need(storage['mus'], F32, 'mus'), _need(storage['sigmas'], F32, 'sigmas'),
        _need(storage['neglogpacs'], F32, 'neglogpacs'), _need(storage['values'], F32, 'values'),
        None if env_actions is None else _need(env_actions[0], F32, 'env actions'),
        None if env_actions is None else _need(env_actions[1], F32, 'actions low'),
        None if env_actions is None else _need(env_actions[2], F32, 'actions high'),
        N, horizon, A, step, _stream(heads)), 'rlg_rollout_policy_head')


def rnn_zero_done_states(states, dones):
    lib = _lib.load()
    L, N, U = states.shape
    _lib.check(lib.rlg_rnn_zero_done_states(_need(states, F32, 'states'), _need(dones, U8, 'dones'),
                                            L, N, U, _stream(states)), 'rlg_rnn_zero_done_states')


# ------------------------------------------------------------------ running statistics

def column_moments_blocks(rows, cols):
    return _lib.load().rlg_column_moments_num_blocks(int(rows), int(cols))


def column_moments(x, mask=None, partials=None):
    """x [rows, C] fp32 -> partials [blocks, 2C+1] fp64."""
    lib = _lib.load()
    x2 = x.reshape(x.shape[0], -1)
    rows, C = x2.shape
    nb = column_moments_blocks(rows, C)
    if partials is None:
        partials = torch.empty((nb, 2 * C + 1), dtype=F64, device=x.device)
    _lib.check(lib.rlg_column_moments(_need(x2, F32, 'x'), _opt(mask, F32, 'mask'), rows, C,
                                      _need(partials, F64, 'partials'), nb, _stream(x)),
               'rlg_column_moments')
    return partials, nb


def column_moments_segments(x, rows_per_segment, table=None, scratch=None):
    """x [segments * rows_per_segment, C] fp32 contiguous -> table [segments, 2C+1] fp64 =
    {sum[C], sumsq[C], rows} per band of rows (the minibatches of an epoch); two launches.
    Returns (table, scratch) so that callers can keep both between epochs."""
    lib = _lib.load()
    x2 = x.reshape(x.shape[0], -1)
    total, C = x2.shape
    rps = int(rows_per_segment)
    if rps <= 0 or total % rps != 0:
        raise ValueError(f'{total} rows are not a whole number of {rps}-row segments')
    segs = total // rps
    nb = column_moments_blocks(rps, C)
    W = 2 * C + 1
    if table is None or table.numel() != segs * W or table.device != x.device:
        table = torch.empty((segs, W), dtype=F64, device=x.device)
    if scratch is None or scratch.numel() < segs * nb * W or scratch.device != x.device:
        scratch = torch.empty(segs * nb * W, dtype=F64, device=x.device)
    _lib.check(lib.rlg_column_moments_segments(_need(x2, F32, 'x'), rps, C, segs, _need(scratch, F64, 'scratch'),
                                               nb, _need(table, F64, 'table'), _stream(x)),
               'rlg_column_moments_segments')
    return table, scratch


_tickets = {}


def _ticket(device):
    """Zero-initialised arrival counter shared by all rms_update launches of a device (the
    launches are stream-ordered and the kernel resets it)."""
    key = str(device)
    if key not in _tickets:
        _tickets[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _tickets[key]


def rms_update(partials, num_blocks, cols, total_rows, mode, running_mean, running_var, count):
    lib = _lib.load()
    _lib.check(lib.rlg_rms_update(_need(partials, F64, 'partials'), num_blocks, cols, total_rows, mode,
                                  _need(running_mean, F64, 'running_mean'),
                                  _need(running_var, F64, 'running_var'),
                                  _need(count, torch.int64, 'count'),
                                  _ticket(partials.device).data_ptr(), _stream(partials)),
               'rlg_rms_update')


def rms_apply(x, running_mean, running_var, eps, mode=0, out=None):
    lib = _lib.load()
    x2 = x.reshape(x.shape[0], -1) if x.dim() > 1 else x.reshape(-1, 1)
    rows, C = x2.shape
    if out is None:
        out = torch.empty_like(x, memory_format=torch.contiguous_format)
    _lib.check(lib.rlg_rms_apply(_need(x2, F32, 'x'), _need(out, F32, 'out'), rows, C,
                                 _need(running_mean, F64, 'running_mean'),
                                 _need(running_var, F64, 'running_var'), float(np.float32(eps)), mode,
                                 _stream(x)), 'rlg_rms_apply')
    return out


class StatsSyncKernels:
    """Device side of distributed.StatsSync: the normalisers' state tensors as the pointer tables of
    rlg_stats_sync_pack / rlg_stats_sync_apply (include/rlg_hip.h)."""

    PACK_DELTAS, PACK_SEED, PACK_STATE = 0, 1, 2
    APPLY_MERGE, APPLY_STATE = 0, 2

    MAX_SEGMENTS = 8      # kStatsMaxSegments of csrc/running_stats.hip

    def __init__(self, modules):
        import ctypes
        n = len(modules)
        if n > self.MAX_SEGMENTS:
            raise ValueError(f'StatsSyncKernels: {n} normalisers, one launch carries at most {self.MAX_SEGMENTS} '
                             '(build one StatsSync per group of normalisers)')
        self.device = modules[0].running_mean.device
        self._keep = [(m.running_mean, m.running_var, m.count) for m in modules]
        self.dims = [int(m.running_mean.numel()) for m in modules]
        self._means = (ctypes.c_void_p * n)(*[_need(m.running_mean, F64, 'running_mean') for m in modules])
        self._vars = (ctypes.c_void_p * n)(*[_need(m.running_var, F64, 'running_var') for m in modules])
        self._counts = (ctypes.c_void_p * n)(*[_need(m.count, torch.int64, 'count') for m in modules])
        self._dims = (ctypes.c_int * n)(*self.dims)
        self._n = n
        self._ctypes = ctypes
        self.flat_size = int(_lib.load().rlg_stats_sync_flat_size(n, ctypes.cast(self._dims, ctypes.c_void_p)))
        self._ptrs = [(m.running_mean.data_ptr(), m.running_var.data_ptr(), m.count.data_ptr()) for m in modules]

    def bound_to(self, modules):
        """True while the pointer tables still address these modules' state tensors."""
        return len(modules) == self._n and all(
            (m.running_mean.data_ptr(), m.running_var.data_ptr(), m.count.data_ptr()) == p
            for m, p in zip(modules, self._ptrs))

    def _tables(self, has_snapshot):
        c = self._ctypes
        flags = (c.c_int * self._n)(*[int(bool(h)) for h in has_snapshot])
        return (self._n, c.cast(self._means, c.c_void_p), c.cast(self._vars, c.c_void_p),
                c.cast(self._counts, c.c_void_p), c.cast(self._dims, c.c_void_p), c.cast(flags, c.c_void_p)), flags

    def pack(self, has_snapshot, snapshot, out, mode):
        args, keep = self._tables(has_snapshot)
        _lib.check(_lib.load().rlg_stats_sync_pack(*args, _need(snapshot, F64, 'snapshot'),
                                                   _need(out, F64, 'out'), mode, _stream(out)),
                   'rlg_stats_sync_pack')

    def apply(self, has_snapshot, snapshot, reduced, mode):
        args, keep = self._tables(has_snapshot)
        _lib.check(_lib.load().rlg_stats_sync_apply(*args, _need(snapshot, F64, 'snapshot'),
                                                    _need(reduced, F64, 'reduced'), mode, _stream(reduced)),
                   'rlg_stats_sync_apply')


PREP_NORM_VALUE = 1
PREP_NORM_ADV = 2
PREP_FREEZE_CRITIC = 4
PREP_EMA_ADV = 8


def prepare_stats_buffer(device):
    n = _lib.load().rlg_prepare_stats_bytes()
    return torch.zeros(n // 4, dtype=F32, device=device)


def prepare_finalize(gae_partials, batch, flags, value_stats, eps, ema, stats_out):
    """value_stats = (running_mean, running_var, count) or None; ema = dict(mean, sqrs, step,
    decay, max, eps) or None."""
    lib = _lib.load()
    rm = rv = cnt = None
    if value_stats is not None:
        rm, rv, cnt = value_stats
    em = es = est = None
    decay = factor = 0.0
    emax, eeps = 1e5, 0.0
    if ema is not None:
        em, es, est = ema['mean'], ema['sqrs'], ema['step']
        decay = float(np.float32(ema['decay']))
        factor = float(np.float32(1 - ema['decay']))
        emax, eeps = float(ema['max']), float(ema['eps'])
    _lib.check(lib.rlg_prepare_finalize(
        _need(gae_partials, F64, 'gae_partials'), gae_partials.shape[0], gae_partials.shape[1], batch, flags,
        _opt(rm, F64, 'value running_mean'), _opt(rv, F64, 'value running_var'),
        _opt(cnt, torch.int64, 'value count'), float(np.float32(eps)), _opt(em, F32, 'ema mean'),
        _opt(es, F32, 'ema sqrs'), _opt(est, torch.int32, 'ema step'), decay, factor,
        float(np.float32(emax)), float(np.float32(eeps)), _need(stats_out, F32, 'stats_out'),
        _stream(gae_partials)), 'rlg_prepare_finalize')


def triple_moments(advantages, values, returns, mask=None):
    """[blocks, 6 or 7] fp64 partial sums in the GAE-partials format."""
    lib = _lib.load()
    B = advantages.numel()
    nb = lib.rlg_triple_moments_num_blocks(B)
    part = torch.empty((nb, 7 if mask is not None else 6), dtype=F64, device=advantages.device)
    _lib.check(lib.rlg_triple_moments(_need(advantages, F32, 'advantages'), _need(values, F32, 'values'),
                                      _need(returns, F32, 'returns'), _opt(mask, F32, 'mask'), B,
                                      part.data_ptr(), nb, _stream(advantages)), 'rlg_triple_moments')
    return part


def prepare_apply(values, returns, advantages, flags, stats, out=None):
    """out = (values_out, returns_out, advantages_out) or None for in place."""
    lib = _lib.load()
    B = advantages.numel()
    vo, ro, ao = (values, returns, advantages) if out is None else out
    _lib.check(lib.rlg_prepare_apply(_need(values, F32, 'values'), _need(returns, F32, 'returns'),
                                     _need(advantages, F32, 'advantages'), _need(vo, F32, 'values_out'),
                                     _need(ro, F32, 'returns_out'), _need(ao, F32, 'advantages_out'),
                                     B, flags, _need(stats, F32, 'stats'), _stream(values)),
               'rlg_prepare_apply')


# ------------------------------------------------------------------ PPO loss

BOUND_KINDS = {None: 0, 'none': 0, 'bound': 1, 'regularisation': 2}


def ppo_loss_blocks(minibatch):
    return _lib.load().rlg_ppo_loss_num_blocks(int(minibatch))


def ppo_loss_partials_per_block(actions):
    return _lib.load().rlg_ppo_loss_partials_per_block(int(actions))


def _rows_view(t, name):
    """(pointer, row stride) of a [rows, cols] / [rows] fp32 view whose inner stride is 1."""
    _lib.require_gpu(t, name)
    if t.dtype != F32:
        raise ValueError(f'{name}: expected fp32')
    if t.dim() == 2:
        if t.shape[1] != 1 and t.stride(1) != 1:
            raise ValueError(f'{name}: inner stride must be 1, got {t.stride()}')
        return t.data_ptr(), t.stride(0)
    if t.dim() == 1:
        return t.data_ptr(), t.stride(0)
    raise ValueError(f'{name}: expected 1-D or 2-D')


SURROGATE_CLIP, SURROGATE_SMOOTH, SURROGATE_NONE = 0, 1, 2


def _surrogate_kind(smooth):
    """The `smooth` argument of the loss launches: False / True (`use_smooth_clamp`) or one of SURROGATE_* -
    SURROGATE_NONE is `ppo: False`, the plain A2C actor loss neglogp * advantage (common_losses.py:59, 80)."""
    k = int(smooth)
    if k not in (SURROGATE_CLIP, SURROGATE_SMOOTH, SURROGATE_NONE):
        raise ValueError(f'surrogate kind {smooth!r}')
    return k


def ppo_loss_fused(mu, logstd, values, actions, old_neglogp, advantages, old_values, returns,
                   old_mu, old_sigma, d_mu, d_values, partials, e_clip, critic_coef, bounds_coef,
                   clip_value=True, smooth=False, bound_kind=1, write_back=True, mask=None,
                   mask_sum=None):
    """mu/d_mu [mb, A] and values/d_values [mb] may be strided row views (fused head buffer)."""
    lib = _lib.load()
    mb, A = mu.shape
    mu_p, ld_mu = _rows_view(mu, 'mu')
    val_p, ld_val = _rows_view(values, 'values')
    dmu_p, ld_dmu = _rows_view(d_mu, 'd_mu')
    dval_p, ld_dval = _rows_view(d_values, 'd_values')
    _lib.check(lib.rlg_ppo_loss_fused(
        mu_p, _need(logstd, F32, 'logstd'), val_p,
        _need(actions, F32, 'actions'), _need(old_neglogp, F32, 'old_neglogp'),
        _need(advantages, F32, 'advantages'), _need(old_values, F32, 'old_values'),
        _need(returns, F32, 'returns'), _need(old_mu, F32, 'old_mu'), _need(old_sigma, F32, 'old_sigma'),
        _opt(mask, F32, 'mask'), _opt(mask_sum, F32, 'mask_sum'), dmu_p,
        dval_p, _need(partials, F64, 'partials'), mb, A, ld_mu, ld_val, ld_dmu, ld_dval,
        float(np.float32(e_clip)), float(np.float32(critic_coef)), float(np.float32(bounds_coef)),
        1 if clip_value else 0, _surrogate_kind(smooth), bound_kind, 1 if write_back else 0, _stream(mu)),
        'rlg_ppo_loss_fused')


def ppo_loss_desc(mu, logstd, values, actions, old_neglogp, advantages, old_values, returns,
                  old_mu, old_sigma, d_mu, d_values, partials, e_clip, critic_coef, bounds_coef,
                  clip_value=True, smooth=False, bound_kind=1, write_back=True, mask=None, mask_sum=None):
    """The arguments of ppo_loss_fused as an rlg_ppo_loss_desc for MlpChain.backward(ppo_loss=...): the
    backward launch evaluates the loss of each row tile in front of its own work.  d_mu / d_values must be
    views of the d_heads tensor that backward then reads; partials needs MlpChain.num_blocks(rows, 1)
    rows.  The descriptor keeps no tensor alive - the caller does (they are the step's static buffers)."""
    mb, A = mu.shape
    mu_p, ld_mu = _rows_view(mu, 'mu')
    val_p, ld_val = _rows_view(values, 'values')
    dmu_p, ld_dmu = _rows_view(d_mu, 'd_mu')
    dval_p, ld_dval = _rows_view(d_values, 'd_values')
    return _lib.PpoLossDesc(
        mu_p, _need(logstd, F32, 'logstd'), val_p, _need(actions, F32, 'actions'),
        _need(old_neglogp, F32, 'old_neglogp'), _need(advantages, F32, 'advantages'),
        _need(old_values, F32, 'old_values'), _need(returns, F32, 'returns'), _need(old_mu, F32, 'old_mu'),
        _need(old_sigma, F32, 'old_sigma'), _opt(mask, F32, 'mask'), _opt(mask_sum, F32, 'mask_sum'), dmu_p, dval_p,
        _need(partials, F64, 'partials'), mb, A, ld_mu, ld_val, ld_dmu, ld_dval,
        float(np.float32(e_clip)), float(np.float32(critic_coef)), float(np.float32(bounds_coef)),
        1 if clip_value else 0, _surrogate_kind(smooth), bound_kind, 1 if write_back else 0)


def value_loss(values, old_values, returns, d_values, partials, e_clip, clip_value=True, mask=None, mask_sum=None):
    """Central-value critic loss + its gradient; partials [ceil(mb/256), 7] for ppo_loss_finalize."""
    lib = _lib.load()
    mb = values.shape[0]
    _lib.check(lib.rlg_value_loss(_need(values, F32, 'values'), _need(old_values, F32, 'old_values'),
                                  _need(returns, F32, 'returns'), _opt(mask, F32, 'mask'),
                                  _opt(mask_sum, F32, 'mask_sum'), _need(d_values, F32, 'd_values'),
                                  _need(partials, F64, 'partials'), mb, float(np.float32(e_clip)),
                                  1 if clip_value else 0, _stream(values)), 'rlg_value_loss')


def ppo_loss_discrete_blocks(minibatch):
    return _lib.load().rlg_ppo_loss_discrete_num_blocks(int(minibatch))


def ppo_loss_discrete(logits, values, actions, old_neglogp, advantages, old_values, returns, d_logits,
                      d_values, partials, e_clip, critic_coef, entropy_coef, clip_value=True,
                      smooth=False, mask=None, mask_sum=None, branch_sizes=None, action_masks=None):
    """Categorical PPO loss + gradients.  branch_sizes: widths of the multi-discrete heads
    (default: one head of logits.shape[1]); actions [mb] or [mb, branches] int64; action_masks
    [mb, n] bool/uint8 or None."""
    import ctypes
    lib = _lib.load()
    mb, n = logits.shape
    _lib.require_gpu(logits, 'logits')
    if logits.dtype != F32 or logits.stride(1) != 1:
        raise ValueError('logits: fp32 with unit inner stride expected')
    sizes = [n] if branch_sizes is None else [int(s) for s in branch_sizes]
    if sum(sizes) != n:
        raise ValueError(f'branch sizes {sizes} do not add up to {n} logits')
    if actions.numel() != mb * len(sizes):
        raise ValueError(f'actions must hold {len(sizes)} indices per row')
    am = None
    if action_masks is not None:
        if action_masks.dtype == torch.bool:
            action_masks = action_masks.view(torch.uint8)
        if tuple(action_masks.shape) != (mb, n):
            raise ValueError('action_masks must be [minibatch, sum(branch sizes)]')
        am = _need(action_masks, torch.uint8, 'action_masks')
    # values / d_values / d_logits may be columns of wider rows (the fused chain's [value | logits] head matrix)
    for name, t, shape in (('values', values, (mb,)), ('d_values', d_values, (mb,)), ('d_logits', d_logits, (mb, n))):
        _lib.require_gpu(t, name)
        if t.dtype != F32 or tuple(t.shape) != shape or (t.dim() == 2 and t.stride(1) != 1):
            raise ValueError(f'{name}: fp32 {shape} with unit inner stride expected')
    arr = (ctypes.c_int * len(sizes))(*sizes)
    _lib.check(lib.rlg_ppo_loss_discrete_strided(
        logits.data_ptr(), logits.stride(0), values.data_ptr(), max(values.stride(0), 1),
        _need(actions, torch.int64, 'actions'), am, arr, len(sizes), _need(old_neglogp, F32, 'old_neglogp'),
        _need(advantages, F32, 'advantages'), _need(old_values, F32, 'old_values'),
        _need(returns, F32, 'returns'), _opt(mask, F32, 'mask'), _opt(mask_sum, F32, 'mask_sum'),
        d_logits.data_ptr(), d_logits.stride(0), d_values.data_ptr(), max(d_values.stride(0), 1),
        _need(partials, F64, 'partials'),
        mb, float(np.float32(e_clip)), float(np.float32(critic_coef)), float(np.float32(entropy_coef)),
        1 if clip_value else 0, _surrogate_kind(smooth), _stream(logits)), 'rlg_ppo_loss_discrete_strided')


def ppo_loss_finalize(partials, num_blocks, actions_num, minibatch, masked, critic_coef,
                      entropy_coef, bounds_coef, scalars, d_logstd, kl_slot=None, d_mu_bias=None,
                      d_value_bias=None):
    lib = _lib.load()
    _lib.check(lib.rlg_ppo_loss_finalize(
        _need(partials, F64, 'partials'), num_blocks, actions_num, minibatch, 1 if masked else 0,
        float(np.float32(critic_coef)), float(np.float32(entropy_coef)), float(np.float32(bounds_coef)),
        _need(scalars, F32, 'scalars'), _need(d_logstd, F32, 'd_logstd'),
        _opt(kl_slot, F32, 'kl_slot'), _opt(d_mu_bias, F32, 'd_mu_bias'),
        _opt(d_value_bias, F32, 'd_value_bias'), _stream(partials)), 'rlg_ppo_loss_finalize')


def loss_finalize_desc(partials, num_blocks, actions_num, minibatch, masked, critic_coef, entropy_coef,
                       bounds_coef, scalars, d_logstd, kl_slot=None, d_mu_bias=None, d_value_bias=None):
    """The arguments of ppo_loss_finalize as an rlg_loss_finalize_desc: MlpDwPlan.launch(loss_finalize=...)
    folds the loss partials in its finalise launch; ppo_loss_finalize_from(desc) is the stand-alone launch."""
    return _lib.LossFinalizeDesc(
        _need(partials, F64, 'partials'), int(num_blocks), int(actions_num), int(minibatch), 1 if masked else 0,
        float(np.float32(critic_coef)), float(np.float32(entropy_coef)), float(np.float32(bounds_coef)),
        _need(scalars, F32, 'scalars'), _need(d_logstd, F32, 'd_logstd'), _opt(kl_slot, F32, 'kl_slot'),
        _opt(d_mu_bias, F32, 'd_mu_bias'), _opt(d_value_bias, F32, 'd_value_bias'))


def ppo_loss_finalize_from(desc, device):
    _lib.check(_lib.load().rlg_ppo_loss_finalize(
        desc.partials, desc.num_blocks, desc.actions_num, desc.minibatch, desc.masked, desc.critic_coef,
        desc.entropy_coef, desc.bounds_coef, desc.scalars8, desc.d_logstd, desc.kl_slot_or_null,
        desc.d_mu_bias_or_null, desc.d_value_bias_or_null, _lib.stream_handle(device)), 'rlg_ppo_loss_finalize')


# ------------------------------------------------------------------ manual MLP backward

ACT_KINDS = {'None': 0, None: 0, 'elu': 1, 'relu': 2, 'tanh': 3}


def act_bwd_blocks(rows, cols):
    return _lib.load().rlg_act_bwd_num_blocks(int(rows), int(cols))


def act_bwd_colsum(d_out, pre_act, d_pre, act_kind, partials, num_blocks):
    """d_pre = d_out * act'(pre_act) (d_pre may be d_out); partials[blocks, C] column sums."""
    lib = _lib.load()
    rows, C = d_out.shape
    _lib.check(lib.rlg_act_bwd_colsum(_need(d_out, F32, 'd_out'), _opt(pre_act, F32, 'pre_act'),
                                      _need(d_pre, F32, 'd_pre'), rows, C, d_out.stride(0), act_kind,
                                      _need(partials, F64, 'partials'), num_blocks, _stream(d_out)),
               'rlg_act_bwd_colsum')


def colsum_finalize(partials, num_blocks, cols, out, accumulate=False):
    lib = _lib.load()
    _lib.check(lib.rlg_colsum_finalize(_need(partials, F64, 'partials'), num_blocks, cols,
                                       _need(out, F32, 'out'), 1 if accumulate else 0,
                                       _stream(partials)), 'rlg_colsum_finalize')


# ------------------------------------------------------------------ optimiser

def grad_norm_blocks(n):
    return _lib.load().rlg_grad_norm_num_blocks(int(n))


def grad_sumsq(grads, grad_scale, partials, step_counter=None):
    """Also advances the device-resident Adam step counter when given."""
    lib = _lib.load()
    _lib.check(lib.rlg_grad_sumsq(_need(grads, F32, 'grads'), grads.numel(),
                                  float(np.float32(grad_scale)), _need(partials, F64, 'partials'),
                                  partials.numel(), _opt(step_counter, torch.int64, 'step_counter'),
                                  _stream(grads)), 'rlg_grad_sumsq')


def adam_step(params, grads, exp_avg, exp_avg_sq, norm_partials, grad_scale, max_norm, lr_slots,
              step_counter, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, schedule_kind=0,
              kl=None, kl_scale=1.0, kl_threshold=0.008, min_lr=1e-6, max_lr=1e-2,
              lr_multiplier=1.5, stats_out=None, skip_flag=None, pack=None):
    """skip_flag: device address (int) of the in-graph all-reduce's error word, or None.
    pack = MlpChain.adam_pack_target() (n, weights, in, out, planes address): the launch also leaves the chain's
    weight planes for the new weights (rlg_adam_step_pack, csrc/mlp_chain_bx.hip)."""
    lib = _lib.load()
    args = (
        _need(params, F32, 'params'), _need(grads, F32, 'grads'), _need(exp_avg, F32, 'exp_avg'),
        _need(exp_avg_sq, F32, 'exp_avg_sq'), params.numel(), _opt(norm_partials, F64, 'norm_partials'),
        0 if norm_partials is None else norm_partials.numel(), float(np.float32(grad_scale)),
        float(np.float32(max_norm)), _need(lr_slots, F64, 'lr_slots'),
        _need(step_counter, torch.int64, 'step_counter'),
        float(betas[0]), float(betas[1]), float(eps), float(weight_decay), schedule_kind,
        _opt(kl, F32, 'kl'), float(np.float32(kl_scale)), float(kl_threshold), float(min_lr),
        float(max_lr), float(lr_multiplier), _opt(stats_out, F32, 'stats_out'), skip_flag)
    if pack is not None:
        n, w, ins, outs, planes = pack
        _lib.check(lib.rlg_adam_step_pack(*args, n, w, ins, outs, planes, _stream(params)), 'rlg_adam_step_pack')
    else:
        _lib.check(lib.rlg_adam_step(*args, _stream(params)), 'rlg_adam_step')


# ------------------------------------------------------------------ MLP forward / dX (MFMA, fused epilogues)

def mlp_rowgemm_supported(reduction_dim, leading_dim):
    return bool(_lib.load().rlg_mlp_rowgemm_supported(int(reduction_dim), int(leading_dim)))


def mlp_linear_act_forward(x, w, bias, out, pre_act=None, act_kind=0):
    """out = act(x @ w.T + bias); pre_act (optional) receives x @ w.T + bias.  csrc/mlp_rowgemm.hip."""
    lib = _lib.load()
    rows, K = x.shape
    N = w.shape[0]
    if w.shape[1] != K or out.shape != (rows, N) or x.stride(1) != 1 or out.stride(1) != 1:
        raise ValueError('mlp_linear_act_forward: shape mismatch')
    if pre_act is not None and (pre_act.shape != out.shape or pre_act.stride() != out.stride()):
        raise ValueError('pre_act must have the layout of out')
    _lib.require_gpu(x, 'x')
    _lib.check(lib.rlg_mlp_linear_act_forward(
        x.data_ptr(), x.stride(0), _need(w, F32, 'w'), _opt(bias, F32, 'bias'),
        None if pre_act is None else pre_act.data_ptr(), out.data_ptr(), out.stride(0), rows, N, K,
        act_kind, _stream(x)), 'rlg_mlp_linear_act_forward')


def mlp_linear_act_backward(dz, w, z_prev, dz_prev, act_kind=0):
    """dz_prev = (dz @ w) * act'(z_prev)  (z_prev None: plain dz @ w).  csrc/mlp_rowgemm.hip."""
    lib = _lib.load()
    rows, No = dz.shape
    Mi = w.shape[1]
    if w.shape[0] != No or dz_prev.shape != (rows, Mi) or dz.stride(1) != 1 or dz_prev.stride(1) != 1:
        raise ValueError('mlp_linear_act_backward: shape mismatch')
    if z_prev is not None and (z_prev.shape != dz_prev.shape or z_prev.stride() != dz_prev.stride()):
        raise ValueError('z_prev must have the layout of dz_prev')
    _lib.require_gpu(dz, 'dz')
    _lib.check(lib.rlg_mlp_linear_act_backward(
        dz.data_ptr(), dz.stride(0), _need(w, F32, 'w'), None if z_prev is None else z_prev.data_ptr(),
        dz_prev.data_ptr(), dz_prev.stride(0), rows, No, Mi, act_kind, _stream(dz)),
        'rlg_mlp_linear_act_backward')


NARROW_MAX = 8


def narrow_dx(dz, w, dx):
    """dx [rows, in] = dz [rows, out] @ w [out, in] for out <= 8 (csrc/mlp_narrow.hip)."""
    rows, No = dz.shape
    if w.shape[0] != No or dx.shape != (rows, w.shape[1]) or dz.stride(1) != 1 or dx.stride(1) != 1:
        raise ValueError('narrow_dx: shape mismatch')
    _lib.require_gpu(dz, 'dz')
    _lib.check(_lib.load().rlg_narrow_dx(dz.data_ptr(), dz.stride(0), _need(w, F32, 'w'), dx.data_ptr(), dx.stride(0),
                                         rows, No, w.shape[1], _stream(dz)), 'rlg_narrow_dx')


_narrow_scratch = {}


def narrow_dw(dz, x, grad):
    """grad [out, in] = dz[rows, out].T @ x[rows, in] for in <= 8, out <= 256 (csrc/mlp_narrow.hip)."""
    rows, No = dz.shape
    Mi = x.shape[1]
    if x.shape[0] != rows or tuple(grad.shape) != (No, Mi) or dz.stride(1) != 1 or x.stride(1) != 1:
        raise ValueError('narrow_dw: shape mismatch')
    _lib.require_gpu(dz, 'dz')
    lib = _lib.load()
    need = lib.rlg_narrow_dw_blocks(rows) * No * Mi
    key = (str(dz.device), need)
    if key not in _narrow_scratch:          # (fixed address per shape: the launches are replayed from HIP graphs)
        _narrow_scratch[key] = torch.empty(need, dtype=F64, device=dz.device)
    _lib.check(lib.rlg_narrow_dw(dz.data_ptr(), dz.stride(0), x.data_ptr(), x.stride(0), _need(grad, F32, 'grad'),
                                 _narrow_scratch[key].data_ptr(), rows, No, Mi, _stream(dz)), 'rlg_narrow_dw')


# ------------------------------------------------------------------ fused MLP chain (MFMA, LDS-resident)

chain_timers = None     # bench.py: {'fwd': [...], 'bwd': [...]} of gae.HipEventPair, one per eager chain launch


# fixed operand scales of the split-fp16 form (csrc/bx_form.hpp kBxScaleH / kBxScaleObsNorm)
SPLIT_SCALE_HIDDEN = 16.0
SPLIT_SCALE_OBS_NORM = 4096.0


def chain_split_form():
    """(plane products per fp32 product, plane type) of this build's split-product chain kernels: (3, 'fp16') or (6, 'bf16')."""
    k = int(_lib.load().rlg_mlp_chain_split_products())
    return k, ('fp16' if k == 3 else 'bf16')


def _time_chain_launch(kind):
    if chain_timers is None or torch.cuda.is_current_stream_capturing():
        return
    from .gae import HipEventPair
    ev = HipEventPair()
    chain_timers.setdefault(kind, []).append(ev)
    _lib.check(_lib.load().rlg_mlp_chain_time_next(ev.start, ev.stop), 'rlg_mlp_chain_time_next')


def _untime_chain_launch(kind):
    """The launch that _time_chain_launch(kind) announced did not happen (the entry declined the shape)."""
    if chain_timers is None or torch.cuda.is_current_stream_capturing():
        return
    if chain_timers.get(kind):
        chain_timers[kind].pop()
    _lib.load().rlg_mlp_chain_time_next(None, None)


class MlpChain:
    """The whole MLP (hidden layers + the fused value|mu head as the last layer) as ONE forward and
    ONE backward launch (csrc/mlp_chain.hip).  `layers`: list of (weight [out, in], bias [out], act
    name) - tensors are referenced, not copied, so optimiser updates are seen.  Raises
    NotImplementedError if the network does not fit the LDS even with one 16-row group."""

    def __init__(self, layers, device, weights_version=None):
        import ctypes
        self.n = n = len(layers)
        self.layers = layers
        self.device = device
        # callable -> a value that changes whenever the referenced weights change (FlatArena.weights_token):
        # planes packed by a training forward are re-used by backward() only for the SAME weights
        self._weights_version = weights_version
        self.ins = [int(w.shape[1]) for w, _, _ in layers]
        self.outs = [int(w.shape[0]) for w, _, _ in layers]
        P, I, L = ctypes.c_void_p * n, ctypes.c_int * n, ctypes.c_longlong * n
        self._P, self._I, self._L = P, I, L
        self._w = P(*[_need(w, F32, 'weight') for w, _, _ in layers])
        self._b = P(*[_need(b, F32, 'bias') for _, b, _ in layers])
        self._in, self._out = I(*self.ins), I(*self.outs)
        self._act = I(*[ACT_KINDS[a] for _, _, a in layers])
        lib = _lib.load()
        _lib.check(lib.rlg_mlp_chain_prepare(), 'rlg_mlp_chain_prepare')
        self.max_groups = {}
        for direction in (0, 1):
            best = 0
            for g in (1, 2, 4):
                if lib.rlg_mlp_chain_lds_bytes(n, self._in, self._out, g, direction) >= 0:
                    best = g
            if best == 0 and (direction == 0 or n > 1):
                raise NotImplementedError('MLP does not fit the LDS of the fused chain kernels')
            self.max_groups[direction] = best
        # lean 16-row kernels (csrc/mlp_chain_lean.hip, mlp_chain_fwd_lean_kernel / mlp_chain_bwd_lean_kernel): the weights as
        # plane fragments in each wave's consumption order; RLG_CHAIN_LEAN=0 keeps the pipelined kernels
        self._lean = os.environ.get('RLG_CHAIN_LEAN', '1') != '0'
        self._frag_bytes = [int(lib.rlg_mlp_chain_frags_bytes(n, self._in, self._out, 0)),
                            int(lib.rlg_mlp_chain_frags_bytes(n, self._in, self._out, 1)) if n > 1 else -1]
        if self._frag_bytes[0] < 0:
            self._lean = False
        self._frags = None         # [forward fragments | backward fragments], fp32
        self._frags_for = None     # weights version the fragments hold
        self._planes = None        # bf16 plane fragments of the weights, both directions (csrc/mlp_chain_bx.hip)
        self._bwd_offset = 0
        self._planes_fresh = None  # (rows, weights version) of the training forward that packed the backward planes
        self._planes_for = None    # weights version for which BOTH directions' planes are valid (optimiser-written or packed)
        self._planes_packed_once = False
        # split-fp16 backward: per 64-row workgroup the largest magnitude of every dZ tensor (csrc/bx_form.hpp), for the
        # weight-gradient launch that follows it; rows of the last backward that left them
        self._grad_maxima = None
        self._maxima_bwd = None
        self.maxima_rows_per_entry = 64       # 64: left by the 64-row split kernels, 16: by the lean 16-row kernels

    def _grad_maxima_buffer(self, rows, per=64):
        need = max(1024, -(-int(rows) // per))
        if self._grad_maxima is None or self._grad_maxima.shape[1] < need:
            self._grad_maxima = torch.zeros(8, need, dtype=F32, device=self.device)
        return self._grad_maxima

    def _request_lean_maxima(self, rows):
        """The lean 16-row backward / one-launch step (fp16 form) leaves one gradient-maxima entry per 16-row workgroup."""
        if chain_split_form()[1] != 'fp16':
            return False
        buf = self._grad_maxima_buffer(rows, 16)
        _lib.load().rlg_mlp_chain_gradient_maxima(buf.data_ptr(), buf.shape[1])
        return True

    def gradient_maxima(self, rows):
        """[8, entries] fp32: row l = per 64-row workgroup the largest |dZ of layer l| of the backward of this step - when
        that backward (`rows` rows) ran the split-fp16 kernel, else None.  MlpDwPlan.launch(maxima=...) then runs its
        fp16 form."""
        if self._grad_maxima is not None and self._maxima_bwd == rows:
            return self._grad_maxima
        return None

    def invalidate_planes(self):
        """The weights changed behind this object's back: backward() re-packs its planes (and the lean kernels'
        fragments are packed again)."""
        self._planes_fresh = None
        self._planes_for = None
        self._frags_for = None

    def cache_state(self):
        """Host-side record of what the derived copies hold - for callers that issue launches WITHOUT running them
        (HIP-graph capture): taken in front of the capture, put back behind it (restore_cache_state), so that a pack
        launch that was only recorded never counts as done.  The persistent plane / fragment buffers are allocated HERE
        if they are not yet: a tensor first created inside torch.cuda.graph would live in the graph's private pool and
        dangle behind a capture that fails and is discarded."""
        if self._lean and self._frags is None and self._frag_bytes[0] >= 0:
            self._frag_buffer()
        if self._planes is None:
            try:
                self._plane_buffer()
            except ValueError:
                pass                      # (a network the split-product kernels do not take: they never ask for planes)
        return (self._planes_fresh, self._planes_for, self._frags_for, self._planes_packed_once)

    def restore_cache_state(self, state):
        self._planes_fresh, self._planes_for, self._frags_for, self._planes_packed_once = state

    # ---- fragments of the lean 16-row kernels (fp16 planes; the buffer is typed fp32 for its size only)
    def lean_used(self, rows, direction, requested=0):
        """True when the launch of this direction runs the lean 16-row kernel for `rows` rows (16-row workgroups, exact
        products; the backward additionally needs 16-byte aligned H / dZ rows, checked at its launch)."""
        if not self._lean or (direction == 1 and self._frag_bytes[1] < 0):
            return False
        return self.groups(rows, direction, requested) == 1 and not self.split_products(rows, direction, requested)

    def _frag_buffer(self):
        if self._frags is None:
            fb, bb = self._frag_bytes[0], max(self._frag_bytes[1], 0)
            self._frags = torch.empty((fb + bb) // 4, dtype=F32, device=self.device)
        return self._frags

    def _frags_ptr(self, direction):
        return self._frag_buffer().data_ptr() + (self._frag_bytes[0] if direction == 1 else 0)

    def pack_frags(self, stream_of):
        """Weights (+ biases) -> the fragments of both directions, one launch.  Behind every change of the weights:
        forward() / backward() do it themselves unless frags_current(); an agent issues it behind its optimiser step."""
        lib = _lib.load()
        _lib.check(lib.rlg_mlp_chain_pack_frags_both(self.n, self._w, self._b, self._in, self._out, self._frags_ptr(0),
                                                     self._frags_ptr(1) if self._frag_bytes[1] >= 0 else None,
                                                     _stream(stream_of)), 'rlg_mlp_chain_pack_frags_both')

    def frags_current(self):
        v = self._version()
        return v is not None and self._frags_for == v

    def mark_frags(self, version):
        self._frags_for = version

    def ensure_frags(self, stream_of):
        if not self.frags_current():
            self.pack_frags(stream_of)
            self._frags_for = self._version()

    def planes_current(self):
        """Both directions' planes belong to the weights as they are now (needs a weights-version source)."""
        v = self._version()
        return v is not None and self._planes_for == v

    def mark_planes(self, version):
        """The planes of both directions now hold the weights of `version` (FlatAdam.step(pack=...), or after a
        graph replay whose last launch was that step)."""
        self._planes_for = version

    def ensure_planes(self, stream_of):
        """Eager full pack unless the planes are current (in front of a graph replay that contains no pack launch)."""
        if not self.planes_current():
            self.pack_planes(2, stream_of)
            self._planes_for = self._version()

    def adam_pack_target(self):
        """(n, weights, in, out, planes address) for ops.adam_step(pack=...), or None when this network has no use for
        planes / no version source.  The first call packs the whole buffer once (the fragments' zero padding)."""
        if self._weights_version is None or self.n < 1:
            return None
        if not self._planes_packed_once:
            self.pack_planes(2, self.layers[0][0])
            self._planes_packed_once = True
            self._planes_for = self._version()
        return self.n, self._w, self._in, self._out, self._plane_buffer().data_ptr()

    def _version(self):
        return None if self._weights_version is None else self._weights_version()

    def split_products(self, rows, direction, requested=0):
        """True when the launch of this kind (0 inference forward, 1 backward, 2 training forward) runs the split-product
        kernel for `rows` rows."""
        return bool(_lib.load().rlg_mlp_chain_bx_supported(self.n, self._in, self._out, int(rows),
                                                           self.groups(rows, direction, requested), int(direction)))

    def pack_planes(self, direction, stream_of):
        """Weights -> bf16 plane fragments (one launch; direction 0 forward operand, 1 backward, 2 both).  Returns the
        view of that direction's fragments (direction 2: the whole buffer).  The planes must be re-packed after
        every change of the weights: forward() / backward() do it themselves, right in front of their launch."""
        buf = self._plane_buffer()
        lib = _lib.load()
        if direction == 2 and 2 * self.n - 1 > 8:                 # more jobs than one launch carries
            self.pack_planes(0, stream_of)
            self.pack_planes(1, stream_of)
            return buf
        view = (buf[:self._fwd_bytes], buf[self._bwd_offset:self._bwd_offset + self._bwd_bytes], buf)[direction]
        if view.numel() > 0:
            _lib.check(lib.rlg_mlp_chain_pack_planes(self.n, self._w, self._in, self._out, int(direction),
                                                     view.data_ptr(), _stream(stream_of)), 'rlg_mlp_chain_pack_planes')
        return view

    def _plane_buffer(self):
        """ONE buffer for both directions: forward fragments at 0, backward fragments at _bwd_offset."""
        if self._planes is None:
            lib = _lib.load()
            nbytes = lib.rlg_mlp_chain_planes_bytes(self.n, self._in, self._out, 2)
            off = lib.rlg_mlp_chain_planes_offset(self.n, self._in, self._out, 1)
            if nbytes < 0 or off < 0:
                raise ValueError('rlg_mlp_chain_planes_bytes: bad network')
            self._planes = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=self.device)
            self._bwd_offset = int(off)
            self._fwd_bytes = int(lib.rlg_mlp_chain_planes_bytes(self.n, self._in, self._out, 0))
            self._bwd_bytes = int(lib.rlg_mlp_chain_planes_bytes(self.n, self._in, self._out, 1))
        return self._planes

    def _planes_ptr(self, direction):
        return self._plane_buffer().data_ptr() + (self._bwd_offset if direction == 1 else 0)

    def groups(self, rows, direction, requested=0):
        """direction names the launch kind: 0 inference forward, 1 backward, 2 training forward (csrc/mlp_chain.hip
        chain_bx_min_rows: the training pair takes the 64-row split kernels from 8,192 rows, inference from 16,384)."""
        g = _lib.load().rlg_mlp_chain_groups(int(rows), int(requested), int(direction))
        return min(g, self.max_groups[1 if direction == 1 else 0])

    def num_blocks(self, rows, direction, requested=0):
        return _lib.load().rlg_mlp_chain_num_blocks(int(rows), self.groups(rows, direction, requested))

    def forward(self, x, heads, act_out=None, rms=None, eps=1e-5, xn_out=None, groups=0, rms_fold=None,
                split_products=None):
        """x [rows, in0] (row stride free), heads [rows, out_last] out.  act_out: per hidden layer a
        [rows, out_l] tensor or None.  rms = (running_mean, running_var) fp64 -> the observations are
        normalised on the way in (xn_out [rows, in0] optionally receives them).  rms_fold =
        (moments_row [2*in0+1] fp64, count int64, mean_out, var_out, count_out): training-mode
        RunningMeanStd - the minibatch's moments are folded into the state first, the new state is
        written to the *_out tensors (a second buffer set).
        split_products: None = the library's choice (split-bf16 products on 64-row tiles for minibatches of
        >= 16,384 rows, csrc/mlp_chain_bx_fwd.hip), False = exact f32 products."""
        rows = x.shape[0]
        n = self.n
        outs = list(act_out) if act_out is not None else [None] * (n - 1)
        outs = outs + [heads]
        ptrs = self._P(*[None if t is None else _need(t, F32, 'act_out', contiguous=False) for t in outs])
        lds = self._L(*[0 if t is None else t.stride(0) for t in outs])
        _lib.require_gpu(x, 'x')
        if x.dtype != F32 or (x.dim() == 2 and x.shape[1] != 1 and x.stride(1) != 1):
            raise ValueError('x: fp32 rows with unit inner stride expected')
        mean = var = None
        if rms is not None:
            mean, var = _need(rms[0], F64, 'running_mean'), _need(rms[1], F64, 'running_var')
        fold = [None] * 5
        if rms_fold is not None:
            if rms is None:
                raise ValueError('rms_fold needs rms')
            row, cnt, mean_o, var_o, cnt_o = rms_fold
            if row.numel() != 2 * self.ins[0] + 1:
                raise ValueError('rms_fold: moments row of 2*in0+1 doubles expected')
            fold = [_need(row, F64, 'moments row'), _need(cnt, torch.int64, 'count'), _need(mean_o, F64, 'mean out'),
                    _need(var_o, F64, 'var out'), _need(cnt_o, torch.int64, 'count out')]
        # A training forward also has the weights split for the backward launch that follows it: by the forward's own
        # pack launch when the forward runs on planes as well (one launch, both directions), else in extra workgroups
        # of the exact-product forward launch.  backward() uses those planes once; any other caller packs for itself.
        kind = 2 if (act_out is not None and self.n > 1 and any(t is not None for t in act_out)) else 0
        if split_products is not False and groups in (0, 1) and self.lean_used(rows, kind):
            self._planes_fresh = None
            self.ensure_frags(x)
            _time_chain_launch('fwd_train' if act_out is not None else 'fwd_infer')
            err = _lib.load().rlg_mlp_chain_forward_lean(
                n, self._b, self._in, self._out, self._act, ptrs, lds, x.data_ptr(), x.stride(0), mean, var,
                float(np.float32(eps)), _opt(xn_out, F32, 'xn_out'), *fold, rows, self._frags_ptr(0), _stream(x))
            if err != 801:                              # (hipErrorNotSupported: the pipelined / unit-structured launch)
                _lib.check(err, 'rlg_mlp_chain_forward_lean')
                return
            _untime_chain_launch('fwd_train' if act_out is not None else 'fwd_infer')
        train = act_out is not None and self.n > 1
        bwd_split = train and self.split_products(rows, 1)
        planes = fwd_planes = None
        self._planes_fresh = None
        packed_both = False
        if split_products is not False and self.split_products(rows, kind, groups):
            if not self.planes_current():
                self.pack_planes(2 if bwd_split else 0, x)
                packed_both = bwd_split
            fwd_planes = self._planes_ptr(0)
        elif bwd_split and not self.planes_current():
            planes = self._planes_ptr(1)
        _time_chain_launch('fwd_train' if act_out is not None else 'fwd_infer')
        _lib.check(_lib.load().rlg_mlp_chain_forward(
            n, self._w, self._b, self._in, self._out, self._act, ptrs, lds, x.data_ptr(), x.stride(0),
            mean, var, float(np.float32(eps)), _opt(xn_out, F32, 'xn_out'), *fold, rows,
            self.groups(rows, kind, groups), planes, fwd_planes, _stream(x)), 'rlg_mlp_chain_forward')
        if bwd_split:
            # only behind a launch that succeeded, and only for the weights as they are now
            self._planes_fresh = (rows, self._version())
        if packed_both and self._weights_version is not None:
            self._planes_for = self._version()

    def step(self, x, heads, act_out, d_heads, dz_out, bias_partials, ppo_loss, rms=None, eps=1e-5, xn_out=None,
             rms_fold=None):
        """forward(x, heads, act_out, ...) and backward(d_heads, act_out, dz_out, bias_partials, ppo_loss=...) as ONE
        launch (rlg_mlp_chain_step: 16-row workgroups, minibatches < 16,384 rows).  Returns False - having launched
        nothing - when the shape is outside that kernel's envelope; the caller then issues the two launches."""
        rows = x.shape[0]
        n = self.n
        if act_out is None or any(t is None for t in act_out) or ppo_loss is None or n < 2:
            return False
        if self.split_products(rows, 2) or self.split_products(rows, 1) or self.groups(rows, 2) != 1 or self.groups(rows, 1) != 1:
            return False
        lean = self.lean_used(rows, 2) and self.lean_used(rows, 1)
        outs = list(act_out) + [heads]
        ptrs = self._P(*[_need(t, F32, 'act_out', contiguous=False) for t in outs])
        lds = self._L(*[t.stride(0) for t in outs])
        _lib.require_gpu(x, 'x')
        if x.dtype != F32 or (x.dim() == 2 and x.shape[1] != 1 and x.stride(1) != 1):
            raise ValueError('x: fp32 rows with unit inner stride expected')
        mean = var = None
        if rms is not None:
            mean, var = _need(rms[0], F64, 'running_mean'), _need(rms[1], F64, 'running_var')
        fold = [None] * 5
        if rms_fold is not None:
            if rms is None:
                raise ValueError('rms_fold needs rms')
            row, cnt, mean_o, var_o, cnt_o = rms_fold
            if row.numel() != 2 * self.ins[0] + 1:
                raise ValueError('rms_fold: moments row of 2*in0+1 doubles expected')
            fold = [_need(row, F64, 'moments row'), _need(cnt, torch.int64, 'count'), _need(mean_o, F64, 'mean out'),
                    _need(var_o, F64, 'var out'), _need(cnt_o, torch.int64, 'count out')]
        dz = self._P(*([_need(t, F32, 'dz', contiguous=False) for t in dz_out] + [None]))
        dl = self._L(*([t.stride(0) for t in dz_out] + [0]))
        bp = None
        if bias_partials is not None:
            bp = self._P(*([_need(t, F64, 'bias partials') for t in bias_partials] + [None]))
        if lean:
            # the lean form of the same launch (fp32 weight fragments); outside its envelope: the lean forward and the lean
            # backward as two launches (the caller's fallback), which beat the pipelined one-launch step
            self.ensure_frags(x)
            self._maxima_bwd = None
            lean_maxima = self._request_lean_maxima(rows)
            _time_chain_launch('step16')
            err = _lib.load().rlg_mlp_chain_step_lean(
                n, self._b, self._in, self._out, self._act, ptrs, lds, x.data_ptr(), x.stride(0),
                mean, var, float(np.float32(eps)), _opt(xn_out, F32, 'xn_out'), *fold,
                _need(d_heads, F32, 'd_heads', contiguous=False), d_heads.stride(0), dz, dl, bp,
                ctypes.addressof(ppo_loss), rows, self._frags_ptr(0), self._frags_ptr(1), _stream(x))
            if err == 801:
                _untime_chain_launch('step16')
                return False
            _lib.check(err, 'rlg_mlp_chain_step_lean')
            self._planes_fresh = None
            if lean_maxima:
                self._maxima_bwd, self.maxima_rows_per_entry = rows, 16
            return True
        _time_chain_launch('step16')
        err = _lib.load().rlg_mlp_chain_step(
            n, self._w, self._b, self._in, self._out, self._act, ptrs, lds, x.data_ptr(), x.stride(0),
            mean, var, float(np.float32(eps)), _opt(xn_out, F32, 'xn_out'), *fold,
            _need(d_heads, F32, 'd_heads', contiguous=False), d_heads.stride(0), dz, dl, bp,
            ctypes.addressof(ppo_loss), rows, _stream(x))
        if err == 801:                                  # hipErrorNotSupported: outside the fused kernel's envelope
            _untime_chain_launch('step16')
            return False
        _lib.check(err, 'rlg_mlp_chain_step')
        self._planes_fresh = None
        return True

    def backward(self, d_heads, acts, dz_out, bias_partials=None, groups=0, ppo_loss=None, split_products=None):
        """d_heads [rows, out_last]; acts / dz_out: per hidden layer H_l (forward's act_out) and the
        dZ_l output; bias_partials: per hidden layer fp64 [num_blocks(rows, 1), out_l] or None.
        ppo_loss = ops.ppo_loss_desc(...): the launch first evaluates the PPO loss of its row tiles, i.e.
        it produces d_heads itself (and the loss partials, the mu/sigma write-back).
        split_products: None = the library's choice (split-bf16 products on 64-row tiles for minibatches of
        >= 16,384 rows, csrc/mlp_chain_bx.hip), False = exact f32 products."""
        rows = d_heads.shape[0]
        n = self.n
        h = self._P(*([_need(t, F32, 'H', contiguous=False) for t in acts] + [None]))
        hl = self._L(*([t.stride(0) for t in acts] + [0]))
        dz = self._P(*([_need(t, F32, 'dz', contiguous=False) for t in dz_out] + [None]))
        dl = self._L(*([t.stride(0) for t in dz_out] + [0]))
        bp = None
        if bias_partials is not None:
            bp = self._P(*([_need(t, F64, 'bias partials') for t in bias_partials] + [None]))
        _lib.require_gpu(d_heads, 'd_heads')
        if split_products is not False and groups in (0, 1) and self.lean_used(rows, 1):
            self._planes_fresh = None
            self._maxima_bwd = None
            self.ensure_frags(d_heads)
            lean_maxima = self._request_lean_maxima(rows)
            _time_chain_launch('bwd_loss' if ppo_loss is not None else 'bwd')
            err = _lib.load().rlg_mlp_chain_backward_lean(
                n, self._in, self._out, self._act, h, hl, d_heads.data_ptr(), d_heads.stride(0), dz, dl, bp,
                None if ppo_loss is None else ctypes.addressof(ppo_loss), rows, self._frags_ptr(1), _stream(d_heads))
            if err != 801:
                _lib.check(err, 'rlg_mlp_chain_backward_lean')
                if lean_maxima:
                    self._maxima_bwd, self.maxima_rows_per_entry = rows, 16
                return
            _untime_chain_launch('bwd_loss' if ppo_loss is not None else 'bwd')
        planes = None
        if split_products is not False and self.split_products(rows, 1, groups):
            if not self.planes_current() and (self._planes_fresh is None or self._planes_fresh != (rows, self._version())):
                self.pack_planes(1, d_heads)                    # (else: packed with the forward launch of this step)
            planes = self._planes_ptr(1)
        # the split-fp16 launch leaves the gradient maxima the weight-gradient launch scales by - claimed only where the
        # library is sure to take that launch (chain_bx_bwd_eligible: 16-byte aligned H / dZ rows of 4-float groups)
        left_maxima = False
        if planes is not None and chain_split_form()[1] == 'fp16' and all(
                t.data_ptr() % 16 == 0 and t.stride(0) % 4 == 0 and t.shape[1] % 4 == 0 for t in list(acts) + list(dz_out)):
            buf = self._grad_maxima_buffer(rows)
            _lib.load().rlg_mlp_chain_gradient_maxima(buf.data_ptr(), buf.shape[1])
            left_maxima = True
        self._maxima_bwd = rows if left_maxima else None
        if left_maxima:
            self.maxima_rows_per_entry = 64
        self._planes_fresh = None
        _time_chain_launch('bwd_loss' if ppo_loss is not None else 'bwd')
        _lib.check(_lib.load().rlg_mlp_chain_backward(
            n, self._w, self._in, self._out, self._act, h, hl, d_heads.data_ptr(), d_heads.stride(0), dz, dl,
            bp, None if ppo_loss is None else ctypes.addressof(ppo_loss), rows, self.groups(rows, 1, groups),
            planes, _stream(d_heads)), 'rlg_mlp_chain_backward')


# ------------------------------------------------------------------ MLP weight gradients (MFMA)

class MlpDwPlan:
    """All weight-gradient GEMMs of one MLP backward as one launch (csrc/mlp_dw.hip).

    shapes: [(No, Mi)] per weight matrix, for minibatches of `rows` rows.  The plan owns the split-K
    workspaces; `launch(jobs)` takes the (dz [rows, No], x [rows, Mi], grad [No, Mi]) tensors of the
    step.  Raises NotImplementedError when a shape is outside the kernel's envelope (callers use
    the library GEMMs then)."""

    def __init__(self, shapes, rows, device, target_blocks=None):
        import ctypes
        lib = _lib.load()
        n = len(shapes)
        if target_blocks is None:
            # 0: the library's default workgroup count; RLG_DW_BLOCKS: A/B measurements of it (tools only)
            target_blocks = max(int(os.environ.get('RLG_DW_BLOCKS') or 0), 0)
        self.rows, self.n, self.shapes = int(rows), n, [tuple(s) for s in shapes]
        self._plans = (ctypes.c_int * (4 * n))()
        self.workspaces = []
        for k, (No, Mi) in enumerate(self.shapes):
            p4 = (ctypes.c_int * 4)()
            need = lib.rlg_mlp_dw_plan(self.rows, No, Mi, target_blocks, p4)
            if need < 0:
                raise NotImplementedError(f'dW shape [{No} x {Mi}] is not supported by the MFMA path')
            self._plans[4 * k:4 * k + 4] = list(p4)
            self.workspaces.append(torch.empty(need, dtype=F32, device=device))
        P = ctypes.c_void_p * n
        self._dz, self._x, self._grad = P(), P(), P()
        self._ws = P(*[w.data_ptr() for w in self.workspaces])
        self._no = (ctypes.c_int * n)(*[s[0] for s in self.shapes])
        self._mi = (ctypes.c_int * n)(*[s[1] for s in self.shapes])
        self._device = device

    def plan(self, k):
        return tuple(self._plans[4 * k:4 * k + 4])

    def finalize_blocks(self, colsums=(), loss_finalize=None):
        """Workgroups of the finalise launch (= entries of the optional norm partials)."""
        n = sum((No * Mi // 4 + 15) // 16 for No, Mi in self.shapes)
        n += sum((int(c[2]) + 15) // 16 for c in colsums)
        if loss_finalize is not None:
            n += 1 + (2 * loss_finalize.actions_num + 7) // 8
        return n

    def launch(self, jobs, colsums=(), loss_finalize=None, norm=None, maxima=None):
        """jobs: (dz, x, grad) per planned layer.  maxima = (MlpChain.gradient_maxima() [8, entries], layer of each job's
        dz, fixed scale of each job's x - SPLIT_SCALE_HIDDEN / SPLIT_SCALE_OBS_NORM): the launch runs on three fp16 plane
        products per fp32 product (csrc/split_f16.hpp), dz scaled by the power of two the largest entry over a wave's
        rows asks for, x by the scale the forward splits the same tensor with; without them: six bf16 plane products.  colsums: optional (partials fp64 [blocks*cols],
        blocks, cols, out fp32 [cols]) items - bias gradients finished in the same finalise launch.
        loss_finalize: ops.loss_finalize_desc(...) - the PPO loss partials are folded there as well.
        norm = (partials fp64 [>= finalise blocks], grad_scale, step_counter int64 [1]): the finalise
        launch also leaves per-block sums of (g * grad_scale)^2 and advances the Adam step counter (what
        grad_sumsq does) - only meaningful when this launch writes every gradient of the arena.
        Returns the number of finalise blocks (= valid entries of the norm partials)."""
        import ctypes
        if len(jobs) != self.n:
            raise ValueError('job count does not match the plan')
        nc = len(colsums)
        if nc:
            P = ctypes.c_void_p * nc
            cs_part = P(*[_need(c[0], F64, 'colsum partials') for c in colsums])
            cs_out = P(*[_need(c[3], F32, 'colsum out') for c in colsums])
            cs_blocks = (ctypes.c_int * nc)(*[int(c[1]) for c in colsums])
            cs_cols = (ctypes.c_int * nc)(*[int(c[2]) for c in colsums])
        else:
            cs_part = cs_out = cs_blocks = cs_cols = None
        nfin = ctypes.c_int(0)
        if norm is not None:
            need = self.finalize_blocks(colsums, loss_finalize)
            if norm[0].numel() < need:
                raise ValueError(f'norm partials: {need} entries needed, {norm[0].numel()} given')
        for k, (dz, x, grad) in enumerate(jobs):
            No, Mi = self.shapes[k]
            if tuple(dz.shape) != (self.rows, No) or tuple(x.shape) != (self.rows, Mi) or \
                    tuple(grad.shape) != (No, Mi):
                raise ValueError(f'layer {k}: dz {tuple(dz.shape)} / x {tuple(x.shape)} / grad {tuple(grad.shape)} '
                                 f'do not match the planned [{self.rows}] x [{No} x {Mi}]')
            self._dz[k] = _need(dz, F32, 'dz')
            self._x[k] = _need(x, F32, 'x')
            self._grad[k] = _need(grad, F32, 'grad')
        if maxima is not None:
            entries, dzs, xscales = maxima[:3]
            per = int(maxima[3]) if len(maxima) > 3 else 64           # rows one entry covers: 64, or 16 (the lean kernels)
            if len(dzs) != self.n or len(xscales) != self.n or entries.dim() != 2 or entries.shape[0] != 8:
                raise ValueError('maxima: [8, entries] fp32, one dz layer and one x scale per job')
            _lib.check(_lib.load().rlg_mlp_dw_gradient_maxima(
                _need(entries, F32, 'gradient maxima'), int(entries.shape[1]), per, (ctypes.c_int * self.n)(*[int(v) for v in dzs]),
                (ctypes.c_float * self.n)(*[float(v) for v in xscales]), self.n), 'rlg_mlp_dw_gradient_maxima')
        _lib.check(_lib.load().rlg_mlp_dw_launch(self.n, self._dz, self._x, self._ws, self._grad, self._no,
                                                 self._mi, self._plans, self.rows, nc, cs_part, cs_blocks,
                                                 cs_cols, cs_out,
                                                 None if loss_finalize is None else ctypes.addressof(loss_finalize),
                                                 None if norm is None else _need(norm[0], F64, 'norm partials'),
                                                 1.0 if norm is None else float(np.float32(norm[1])),
                                                 None if norm is None else _need(norm[2], torch.int64, 'step_counter'),
                                                 ctypes.byref(nfin), _lib.stream_handle(self._device)),
                   'rlg_mlp_dw_launch')
        return nfin.value


# ------------------------------------------------------------------ recurrent policy (LSTM)

def lstm_supported(hidden):
    return bool(_lib.load().rlg_lstm_supported(int(hidden)))


def lstm_seq_forward(gates, w_hh, h0, c0, dones, out, c_all=None, hprev=None, h_final=None, c_final=None,
                     seq_len=1):
    """gates [S*T, 4H]: input projection (+ both biases) in, activated gates out.  See csrc/lstm.hip."""
    lib = _lib.load()
    B, G = gates.shape
    H = G // 4
    S = B // seq_len
    if S * seq_len != B:
        raise ValueError(f'rows ({B}) must be a multiple of seq_len ({seq_len})')
    _lib.check(lib.rlg_lstm_seq_forward(
        _need(gates, F32, 'gates'), _need(w_hh, F32, 'w_hh'), _need(h0, F32, 'h0'), _need(c0, F32, 'c0'),
        _opt(dones, torch.uint8, 'dones'), _need(out, F32, 'out'), _opt(c_all, F32, 'c_all'),
        _opt(hprev, F32, 'hprev'), _opt(h_final, F32, 'h_final'), _opt(c_final, F32, 'c_final'),
        S, seq_len, H, _stream(gates)), 'rlg_lstm_seq_forward')


def lstm_seq_backward(gates, c_all, c0, dones, w_hh, d_out, d_gates, seq_len):
    lib = _lib.load()
    B, G = gates.shape
    _lib.check(lib.rlg_lstm_seq_backward(
        _need(gates, F32, 'gates'), _need(c_all, F32, 'c_all'), _need(c0, F32, 'c0'),
        _opt(dones, torch.uint8, 'dones'), _need(w_hh, F32, 'w_hh'), _need(d_out, F32, 'd_out'),
        _need(d_gates, F32, 'd_gates'), B // seq_len, seq_len, G // 4, _stream(gates)),
        'rlg_lstm_seq_backward')
