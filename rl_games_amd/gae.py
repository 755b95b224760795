"""GAE on MI355X: the `compute_gae` function seam and the fused env-major pass.

Mirrors the signature of the reference's
`rl_games.triton_kernels.compute_gae(mb_rewards, mb_values, mb_dones, last_values,
last_dones, gamma, tau)` (rl_games/triton_kernels/gae_kernel.py:125-147), which
`A2CBase.discount_values` calls (rl_games/common/a2c_common.py:729-734).  Unlike the
reference there is no backend switch: tensors must live on the GPU and the hand-written
HIP kernels in csrc/gae.hip do the work, or a RuntimeError is raised.
"""
import ctypes

import numpy as np
import torch

from . import _lib


def _f32_scalars(gamma, tau):
    # The reference multiplies gamma*tau as Python doubles and PyTorch then rounds the
    # product once to fp32 when it meets the fp32 tensor (gae_kernel.py:78-79).
    return float(np.float32(gamma)), float(np.float32(float(gamma) * float(tau)))


def _as_done_tensor(d, want_float):
    if want_float:
        return d if d.dtype == torch.float32 else d.to(torch.float32)
    if d.dtype == torch.bool:
        return d.view(torch.uint8)
    return d


def _is_envmajor(x, H, N):
    """x[t, e, 0] stored at e*H + t  (the rollout buffer's physical [N, H] layout)."""
    if x.dim() == 3:
        if x.shape[2] != 1:
            return False
        st = x.stride()[:2]
    else:
        st = x.stride()
    return (N == 1 or st[1] == H) and (H == 1 or st[0] == 1)


def compute_gae(mb_rewards, mb_values, mb_dones, last_values, last_dones, gamma, tau):
    """Advantages `A_t` of shape [horizon, num_envs, value_size] (same maths, op order and
    done semantics as gae_kernel.py:63-80; dones[0] is never read)."""
    lib = _lib.load()
    _lib.require_gpu(mb_rewards, 'compute_gae')
    if mb_rewards.dim() != 3:
        raise ValueError(f'mb_rewards must be [horizon, num_envs, value_size], got {tuple(mb_rewards.shape)}')
    H, N, V = mb_rewards.shape
    if mb_rewards.dtype != torch.float32:
        mb_rewards = mb_rewards.float()
    if mb_values.dtype != torch.float32:
        mb_values = mb_values.float()
    if last_values.dtype != torch.float32:
        last_values = last_values.float()
    if last_values.dim() == 1:
        last_values = last_values.unsqueeze(1)
    byte_like = (torch.uint8, torch.bool)
    want_float = not (mb_dones.dtype in byte_like and last_dones.dtype in byte_like)
    mb_dones = _as_done_tensor(mb_dones, want_float)
    last_dones = _as_done_tensor(last_dones, want_float)
    g, gt = _f32_scalars(gamma, tau)
    stream = _lib.stream_handle(mb_rewards.device)

    fast = (V == 1 and not want_float and lib.rlg_gae_envmajor_supported(H)
            and _is_envmajor(mb_rewards, H, N) and _is_envmajor(mb_values, H, N)
            and _is_envmajor(mb_dones, H, N) and last_values.is_contiguous()
            and last_dones.is_contiguous()
            and all(t.data_ptr() % 16 == 0 for t in (mb_rewards, mb_values, mb_dones)))
    if fast:
        out = torch.empty((N, H), dtype=torch.float32, device=mb_rewards.device)
        _lib.check(lib.rlg_gae_envmajor_raw(
            mb_rewards.data_ptr(), mb_values.data_ptr(), mb_dones.data_ptr(),
            last_values.data_ptr(), last_dones.data_ptr(), out.data_ptr(), N, H, g, gt, stream),
            'rlg_gae_envmajor_raw')
        return out.t().unsqueeze(2)

    advs = torch.empty_like(mb_rewards)
    strides = (*mb_rewards.stride(), *mb_values.stride(), *mb_dones.stride(),
               *last_values.stride(), *last_dones.stride(), *advs.stride(), 0, 0, 0)
    arr = (ctypes.c_longlong * 17)(*strides)
    _lib.check(lib.rlg_gae_strided(
        mb_rewards.data_ptr(), mb_values.data_ptr(), mb_dones.data_ptr(),
        last_values.data_ptr(), last_dones.data_ptr(), advs.data_ptr(), None,
        H, N, V, arr, 1 if want_float else 0, g, gt, stream), 'rlg_gae_strided')
    return advs


def num_moment_partials(num_envs):
    return _lib.load().rlg_gae_envmajor_num_partials(int(num_envs))


class HipEventPair:
    """Two raw hipEvents bracketing one launch (bench.py's roofline timing)."""

    def __init__(self):
        lib = _lib.load()
        self.start, self.stop = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(lib.rlg_event_create(ctypes.byref(self.start)), 'rlg_event_create')
        _lib.check(lib.rlg_event_create(ctypes.byref(self.stop)), 'rlg_event_create')

    def elapsed_us(self):
        out = ctypes.c_float()
        _lib.check(_lib.load().rlg_event_elapsed_us(self.start, self.stop, ctypes.byref(out)),
                   'rlg_event_elapsed_us')
        return float(out.value)

    def __del__(self):
        try:
            lib = _lib.load()
            lib.rlg_event_destroy(self.start)
            lib.rlg_event_destroy(self.stop)
        except Exception:
            pass


def gae_returns_advantages(rewards, values, dones, last_values, last_dones, gamma, tau,
                           out_returns=None, out_advantages=None, moment_partials=None, events=None):
    """Fused rollout epilogue on the buffer's physical layout.

    rewards/values: [N, H] fp32 contiguous, dones: [N, H] uint8, last_values: [N] or [N,1],
    last_dones: [N] uint8.  Returns (returns [N,H], advantages [N,H], partials) where
    `returns = A + values` (a2c_common.py:1060), `advantages = returns - values`
    (a2c_common.py:1598) and partials [ceil(N/64), 6] fp64 hold per-tile sums
    {adv, adv^2, v, v^2, ret, ret^2}.
    """
    lib = _lib.load()
    _lib.require_gpu(rewards, 'gae_returns_advantages')
    N, H = rewards.shape
    if not lib.rlg_gae_envmajor_supported(H):
        raise ValueError(f'horizon {H} is not supported by the fused env-major GAE kernel '
                         f'(multiple of 4, 4..64)')
    for name, t, dt in (('rewards', rewards, torch.float32), ('values', values, torch.float32),
                        ('dones', dones, torch.uint8), ('last_dones', last_dones, torch.uint8),
                        ('last_values', last_values, torch.float32)):
        if t.dtype != dt or not t.is_contiguous():
            raise ValueError(f'{name} must be contiguous {dt}, got {t.dtype} strides {t.stride()}')
    dev = rewards.device
    if out_returns is None:
        out_returns = torch.empty((N, H), dtype=torch.float32, device=dev)
    if out_advantages is None:
        out_advantages = torch.empty((N, H), dtype=torch.float32, device=dev)
    if moment_partials is None:
        moment_partials = torch.empty((num_moment_partials(N), 6), dtype=torch.float64, device=dev)
    g, gt = _f32_scalars(gamma, tau)
    args = (rewards.data_ptr(), values.data_ptr(), dones.data_ptr(), last_values.data_ptr(),
            last_dones.data_ptr(), out_returns.data_ptr(), out_advantages.data_ptr(),
            moment_partials.data_ptr(), N, H, g, gt, _lib.stream_handle(dev))
    if events is None:
        _lib.check(lib.rlg_gae_envmajor_fused(*args), 'rlg_gae_envmajor_fused')
    else:
        _lib.check(lib.rlg_gae_envmajor_fused_timed(*args, events.start, events.stop),
                   'rlg_gae_envmajor_fused_timed')
    return out_returns, out_advantages, moment_partials
