"""ctypes binding of librlg_hip.so (the C ABI declared in include/rlg_hip.h).

The product path has no CPU or eager-PyTorch fallback: if the HIP library is missing,
fails to load, or a launch fails, the caller gets a RuntimeError.  `torch` is imported
first on purpose - the library must bind to the HIP runtime PyTorch already loaded, so
that raw `data_ptr()` addresses and `torch.cuda.current_stream()` handles are valid.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede the dlopen below)

_HERE = os.path.dirname(os.path.abspath(__file__))
# RLG_HIP_LIB: another build of the same ABI (A/B measurements of kernel variants, tools only)
LIB_PATH = os.environ.get('RLG_HIP_LIB') or os.path.join(_HERE, 'librlg_hip.so')

_c_void_p = ctypes.c_void_p
_c_int = ctypes.c_int
_c_float = ctypes.c_float
_c_double = ctypes.c_double
_c_ll = ctypes.c_longlong
_P = _c_void_p  # every device pointer crosses the boundary as an address

# name -> argtypes.  Order/meaning: include/rlg_hip.h.  All functions return int.
_PROTOTYPES = {
    'rlg_gae_strided': [_P, _P, _P, _P, _P, _P, _P, _c_int, _c_int, _c_int,
                        ctypes.POINTER(_c_ll), _c_int, _c_float, _c_float, _P],
    'rlg_gae_envmajor_supported': [_c_int],
    'rlg_gae_envmajor_num_partials': [_c_int],
    'rlg_gae_envmajor_fused': [_P, _P, _P, _P, _P, _P, _P, _P, _c_int, _c_int, _c_float,
                               _c_float, _P],
    'rlg_gae_envmajor_raw': [_P, _P, _P, _P, _P, _P, _c_int, _c_int, _c_float, _c_float, _P],
    'rlg_event_create': [ctypes.POINTER(_P)],
    'rlg_event_destroy': [_P],
    'rlg_event_elapsed_us': [_P, _P, ctypes.POINTER(_c_float)],
    'rlg_gae_envmajor_fused_timed': [_P, _P, _P, _P, _P, _P, _P, _P, _c_int, _c_int, _c_float,
                                     _c_float, _P, _P, _P],
    # experience.hip
    'rlg_rollout_store_step': [_c_int, ctypes.POINTER(_P), ctypes.POINTER(_P),
                               ctypes.POINTER(_c_int), _c_int, _c_int, _c_int, _P],
    'rlg_rollout_store_streaming': [_c_int],
    'rlg_rollout_post_step_num_blocks': [_c_int],
    'rlg_rollout_post_step': [_P, _P, _P, _c_int, _P, _P, _P, _P, _P, _P, _P, _c_float, _c_float,
                              _c_float, _c_float, _c_int, _c_int, _c_float, _c_int, _c_int, _c_int,
                              _c_int, _c_int, _P],
    'rlg_episode_meters_update': [_P, _c_int, _c_int, _c_int, _c_int, _P, _P, _P, _P, _P, _P],
    'rlg_rnn_zero_done_states': [_P, _P, _c_int, _c_int, _c_int, _P],
    'rlg_rollout_policy_head': [_P, _c_int, _P, _P, _P, _P, _c_float, _P, _P, _P, _P, _P, _P, _P,
                                _P, _P, _P, _c_int, _c_int, _c_int, _c_int, _P],
    # running_stats.hip
    'rlg_column_moments_num_blocks': [_c_ll, _c_int],
    'rlg_column_moments': [_P, _P, _c_ll, _c_int, _P, _c_int, _P],
    'rlg_column_moments_segments': [_P, _c_ll, _c_int, _c_int, _P, _c_int, _P, _P],
    'rlg_rms_update': [_P, _c_int, _c_int, _c_ll, _c_int, _P, _P, _P, _P, _P],
    'rlg_rms_apply': [_P, _P, _c_ll, _c_int, _P, _P, _c_float, _c_int, _P],
    'rlg_stats_sync_flat_size': [_c_int, _P],
    'rlg_stats_sync_pack': [_c_int, _P, _P, _P, _P, _P, _P, _P, _c_int, _P],
    'rlg_stats_sync_apply': [_c_int, _P, _P, _P, _P, _P, _P, _P, _c_int, _P],
    'rlg_prepare_stats_bytes': [],
    'rlg_triple_moments_num_blocks': [_c_ll],
    'rlg_triple_moments': [_P, _P, _P, _P, _c_ll, _P, _c_int, _P],
    'rlg_prepare_finalize': [_P, _c_int, _c_int, _c_ll, _c_int, _P, _P, _P, _c_float, _P, _P, _P,
                             _c_float, _c_float, _c_float, _c_float, _P, _P],
    'rlg_prepare_apply': [_P, _P, _P, _P, _P, _P, _c_ll, _c_int, _P, _P],
    # ppo_loss.hip
    'rlg_ppo_loss_num_blocks': [_c_int],
    'rlg_ppo_loss_partials_per_block': [_c_int],
    'rlg_ppo_loss_fused': [_P] * 15 + [_c_int] * 6 + [_c_float, _c_float, _c_float, _c_int, _c_int,
                                                      _c_int, _c_int, _P],
    'rlg_mlp_rowgemm_supported': [_c_int, _c_ll],
    'rlg_mlp_linear_act_forward': [_P, _c_ll, _P, _P, _P, _P, _c_ll, _c_int, _c_int, _c_int, _c_int, _P],
    'rlg_mlp_linear_act_backward': [_P, _c_ll, _P, _P, _P, _c_ll, _c_int, _c_int, _c_int, _c_int, _P],
    'rlg_mlp_dw_plan': [_c_int, _c_int, _c_int, _c_int, _P],
    'rlg_mlp_dw_launch': [_c_int, _P, _P, _P, _P, _P, _P, _P, _c_int, _c_int, _P, _P, _P, _P, _P, _P, _c_float, _P,
                          _P, _P],
    'rlg_narrow_dx': [_P, _c_ll, _P, _P, _c_ll, _c_ll, _c_int, _c_int, _P],
    'rlg_narrow_dw_blocks': [_c_ll],
    'rlg_narrow_dw': [_P, _c_ll, _P, _c_ll, _P, _P, _c_ll, _c_int, _c_int, _P],
    # mlp_chain.hip
    'rlg_mlp_chain_prepare': [],
    'rlg_mlp_chain_groups': [_c_ll, _c_int, _c_int],
    'rlg_mlp_chain_num_blocks': [_c_ll, _c_int],
    'rlg_mlp_chain_lds_bytes': [_c_int, _P, _P, _c_int, _c_int],
    'rlg_mlp_chain_debug_stamps': [_P],
    'rlg_mlp_chain_time_next': [_P, _P],
    'rlg_mlp_chain_gradient_maxima': [_P, _c_int],
    'rlg_mlp_chain_split_products': [],
    'rlg_mlp_dw_gradient_maxima': [_P, _c_int, _c_int, _P, _P, _c_int],
    'rlg_mlp_chain_forward': [_c_int, _P, _P, _P, _P, _P, _P, _P, _P, _c_ll, _P, _P, _c_float, _P,
                              _P, _P, _P, _P, _P, _c_ll, _c_int, _P, _P, _P],
    'rlg_mlp_chain_step': [_c_int, _P, _P, _P, _P, _P, _P, _P, _P, _c_ll, _P, _P, _c_float, _P,
                           _P, _P, _P, _P, _P, _P, _c_ll, _P, _P, _P, _P, _c_ll, _P],
    'rlg_mlp_chain_backward': [_c_int, _P, _P, _P, _P, _P, _P, _P, _c_ll, _P, _P, _P, _P, _c_ll, _c_int, _P, _P],
    # lean 16-row forward (csrc/mlp_chain_lean.hip, mlp_chain_fwd_lean_kernel)
    'rlg_mlp_chain_frags_bytes': [_c_int, _P, _P, _c_int],
    'rlg_mlp_chain_pack_frags': [_c_int, _P, _P, _P, _P, _c_int, _P, _P],
    'rlg_mlp_chain_pack_frags_both': [_c_int, _P, _P, _P, _P, _P, _P, _P],
    'rlg_mlp_chain_step_lean': [_c_int, _P, _P, _P, _P, _P, _P, _P, _c_ll, _P, _P, _c_float, _P,
                                _P, _P, _P, _P, _P, _P, _c_ll, _P, _P, _P, _P, _c_ll, _P, _P, _P],
    'rlg_mlp_chain_backward_lean': [_c_int, _P, _P, _P, _P, _P, _P, _c_ll, _P, _P, _P, _P, _c_ll, _P, _P],
    'rlg_mlp_chain_forward_lean': [_c_int, _P, _P, _P, _P, _P, _P, _P, _c_ll, _P, _P, _c_float, _P,
                                   _P, _P, _P, _P, _P, _c_ll, _P, _P],
    # mlp_chain_bx.hip
    'rlg_mlp_chain_planes_bytes': [_c_int, _P, _P, _c_int],
    'rlg_mlp_chain_planes_offset': [_c_int, _P, _P, _c_int],
    'rlg_mlp_chain_pack_planes': [_c_int, _P, _P, _P, _c_int, _P, _P],
    'rlg_mlp_chain_bx_supported': [_c_int, _P, _P, _c_ll, _c_int, _c_int],
    'rlg_lstm_supported': [_c_int],
    'rlg_lstm_seq_forward': [_P] * 10 + [_c_int, _c_int, _c_int, _P],
    'rlg_lstm_seq_backward': [_P] * 7 + [_c_int, _c_int, _c_int, _P],
    'rlg_value_loss': [_P, _P, _P, _P, _P, _P, _P, _c_int, _c_float, _c_int, _P],
    'rlg_ppo_loss_discrete_num_blocks': [_c_int],
    'rlg_ppo_loss_discrete': [_P, _c_ll, _P, _P, _P, _P, _c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _c_int,
                              _c_float, _c_float, _c_float, _c_int, _c_int, _P],
    'rlg_ppo_loss_discrete_strided': [_P, _c_ll, _P, _c_ll, _P, _P, _P, _c_int, _P, _P, _P, _P, _P, _P, _P, _c_ll, _P,
                                      _c_ll, _P, _c_int, _c_float, _c_float, _c_float, _c_int, _c_int, _P],
    'rlg_ppo_loss_finalize': [_P, _c_int, _c_int, _c_int, _c_int, _c_float, _c_float, _c_float, _P,
                              _P, _P, _P, _P, _P],
    # mlp_fused.hip
    'rlg_act_bwd_num_blocks': [_c_ll, _c_int],
    'rlg_act_bwd_colsum': [_P, _P, _P, _c_ll, _c_int, _c_ll, _c_int, _P, _c_int, _P],
    'rlg_colsum_finalize': [_P, _c_int, _c_int, _P, _c_int, _P],
    # ipc_allreduce.hip
    'rlg_ipc_handle_bytes': [],
    'rlg_ipc_comm_create': [_c_int, _c_int, _c_ll, ctypes.POINTER(_P), _P],
    'rlg_ipc_comm_connect': [_P, ctypes.c_char_p],
    'rlg_ipc_comm_fine_grained': [_P],
    'rlg_ipc_comm_set_timeout': [_P, _c_double],
    'rlg_ipc_comm_set_variant': [_P, _c_int],
    'rlg_ipc_comm_get_config': [_P, ctypes.POINTER(_c_int), ctypes.POINTER(_c_double)],
    'rlg_ipc_comm_error_word': [_P, ctypes.POINTER(_P)],
    'rlg_ipc_allreduce_sum': [_P, _P, _c_ll, _P],
    'rlg_ipc_allreduce_norm_blocks': [],
    'rlg_ipc_allreduce_sum_norm': [_P, _P, _c_ll, _P, _c_ll, _c_float, _P, _P],
    'rlg_ipc_comm_status': [_P, ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint)],
    'rlg_ipc_comm_destroy': [_P],
    # optim.hip
    'rlg_grad_norm_num_blocks': [_c_ll],
    'rlg_grad_sumsq': [_P, _c_ll, _c_float, _P, _c_int, _P, _P],
    'rlg_adam_step': [_P, _P, _P, _P, _c_ll, _P, _c_int, _c_float, _c_float, _P, _P,
                      _c_double, _c_double, _c_double, _c_double, _c_int, _P, _c_float, _c_double,
                      _c_double, _c_double, _c_double, _P, _P, _P],
    'rlg_rccl_available': [],
    'rlg_rccl_unique_id_bytes': [],
    'rlg_rccl_get_unique_id': [ctypes.c_char_p],
    'rlg_rccl_comm_create': [ctypes.c_char_p, _c_int, _c_int, ctypes.POINTER(_P)],
    'rlg_rccl_allreduce_sum': [_P, _P, _c_ll, _P],
    'rlg_rccl_allreduce_sum_f64': [_P, _P, _c_ll, _P],
    'rlg_rccl_comm_destroy': [_P],
    'rlg_adam_step_pack': [_P, _P, _P, _P, _c_ll, _P, _c_int, _c_float, _c_float, _P, _P,
                           _c_double, _c_double, _c_double, _c_double, _c_int, _P, _c_float, _c_double,
                           _c_double, _c_double, _c_double, _P, _P, _c_int, _P, _P, _P, _P, _P],
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def exported_prototypes():
    return dict(_PROTOTYPES)


_RETURNS_LONG_LONG = {'rlg_mlp_dw_plan', 'rlg_stats_sync_flat_size', 'rlg_mlp_chain_planes_bytes', 'rlg_mlp_chain_planes_offset',
                      'rlg_mlp_chain_frags_bytes'}


def load():
    """dlopen librlg_hip.so once and attach argtypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f'{LIB_PATH} is not built. Run `python -c "import __graft_entry__ as g; g.build()"` '
            f'or `make -C rl_games_amd/csrc`. rl_games_amd has no CPU fallback.')
    try:
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    except OSError as e:  # pragma: no cover - depends on the box
        raise HipLibraryError(f'cannot load {LIB_PATH}: {e}') from e
    for name, argtypes in _PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f'{LIB_PATH} does not export {name}; rebuild it') from e
        fn.argtypes = argtypes
        fn.restype = _c_ll if name in _RETURNS_LONG_LONG else _c_int
    _lib = lib
    return lib


class PpoLossDesc(ctypes.Structure):
    """rlg_ppo_loss_desc (include/rlg_hip.h): the PPO-loss arguments of a fused backward launch."""
    _fields_ = ([(n, ctypes.c_void_p) for n in (
        'mu', 'logstd', 'values', 'actions', 'old_neglogp', 'advantages', 'old_values', 'returns', 'old_mu',
        'old_sigma', 'mask_or_null', 'mask_sum_or_null', 'd_mu', 'd_values', 'partials')]
        + [(n, ctypes.c_int) for n in ('minibatch', 'actions_num', 'ld_mu', 'ld_values', 'ld_d_mu', 'ld_d_values')]
        + [(n, ctypes.c_float) for n in ('e_clip', 'critic_coef', 'bounds_coef')]
        + [(n, ctypes.c_int) for n in ('clip_value', 'use_smooth_clamp', 'bound_kind', 'write_back')])


class LossFinalizeDesc(ctypes.Structure):
    """rlg_loss_finalize_desc (include/rlg_hip.h): the arguments of rlg_ppo_loss_finalize, for launches
    that fold the loss partials alongside their own work (rlg_mlp_dw_launch)."""
    _fields_ = [('partials', ctypes.c_void_p), ('num_blocks', ctypes.c_int), ('actions_num', ctypes.c_int),
                ('minibatch', ctypes.c_int), ('masked', ctypes.c_int), ('critic_coef', ctypes.c_float),
                ('entropy_coef', ctypes.c_float), ('bounds_coef', ctypes.c_float), ('scalars8', ctypes.c_void_p),
                ('d_logstd', ctypes.c_void_p), ('kl_slot_or_null', ctypes.c_void_p),
                ('d_mu_bias_or_null', ctypes.c_void_p), ('d_value_bias_or_null', ctypes.c_void_p)]


def check(err, what):
    if err != 0:
        raise HipLibraryError(f'{what} failed with hipError_t {err}')


def ptr(t):
    """Device address of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_handle(device=None):
    """hipStream_t of torch's current stream, as an integer address."""
    return torch.cuda.current_stream(device).cuda_stream


def require_gpu(t, what):
    if not t.is_cuda:
        raise HipLibraryError(
            f'{what}: tensor lives on {t.device}; the rl_games_amd hot path only runs on an '
            f'MI355X (HIP) device and has no CPU fallback')
