"""RCCL gradient all-reduce as a plain stream launch through the C ABI (csrc/rccl_wrap.hip) - capturable into the
mini-epoch HIP graph, unlike a collective issued through torch.distributed.

Same host interface as `ipc_allreduce.IpcAllReduce` (the agent treats the two alike): replaces dist.all_reduce of the
flat gradient arena (rl_games/common/a2c_common.py:493-509, KL slot :1559-1560).  Opt-in (`native_allreduce: 'rccl'`):
the default in-graph collective is the hand-written hipIpc kernel, whose rank-ordered sum also leaves the gradient-norm
partials; this one is the in-graph fallback for a node where that kernel is unavailable.  The ncclComm_t is created from
a unique id that rank 0 generates and the existing torch.distributed group (any backend) carries to the others."""
import ctypes

import torch
import torch.distributed as dist

from . import _lib


class RcclAllReduce:
    kind = 'rccl-in-graph'
    supports_norm = False          # (the reduced values never pass through registers of this library)
    two_phase = False
    error_word = None              # no fail-safe word: RCCL's own watchdog / the process group timeout bound a hang

    def __init__(self, numel, device, rank=None, world=None, group=None):
        lib = _lib.load()
        if not lib.rlg_rccl_available():
            raise _lib.HipLibraryError('librccl could not be loaded (rlg_rccl_available() == 0)')
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.numel = int(numel)
        self.device = torch.device(device)
        self._comm = None
        self._launches = 0
        nb = lib.rlg_rccl_unique_id_bytes()
        uid = ctypes.create_string_buffer(nb)
        err = 0
        if self.rank == 0:
            err = lib.rlg_rccl_get_unique_id(uid)
        if self.world > 1:
            box = [bytes(uid.raw) if err == 0 else None]
            dist.broadcast_object_list(box, src=0, group=group)
            if box[0] is None:
                raise _lib.HipLibraryError('rlg_rccl_get_unique_id failed on rank 0')
            uid = ctypes.create_string_buffer(box[0], nb)
        elif err != 0:
            raise _lib.HipLibraryError(f'rlg_rccl_get_unique_id failed ({err})')
        comm = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            err = lib.rlg_rccl_comm_create(uid.raw, self.rank, self.world, ctypes.byref(comm))      # collective
        if self.world > 1:
            oks = [None] * self.world
            dist.all_gather_object(oks, err, group=group)
        else:
            oks = [err]
        if err == 0:
            self._comm = comm
        if any(oks):
            self.close()
            raise _lib.HipLibraryError(f'rlg_rccl_comm_create failed on a rank (per rank: {oks})')

    def all_reduce_sum(self, t, norm=None):
        """In place, on torch's current stream (capturable).  `t`: contiguous fp32 / fp64 CUDA tensor."""
        if norm is not None:
            raise ValueError('RcclAllReduce does not produce gradient-norm partials (supports_norm is False)')
        if not t.is_contiguous() or t.dtype not in (torch.float32, torch.float64):
            raise ValueError('RcclAllReduce: contiguous fp32 / fp64 tensor expected')
        _lib.require_gpu(t, 'all_reduce_sum')
        lib = _lib.load()
        fn = lib.rlg_rccl_allreduce_sum if t.dtype == torch.float32 else lib.rlg_rccl_allreduce_sum_f64
        _lib.check(fn(self._comm, t.data_ptr(), t.numel(), _lib.stream_handle(t.device)), 'rlg_rccl_allreduce_sum')
        self._launches += 1
        return t

    @staticmethod
    def norm_blocks():
        return 0

    def status(self):
        """(launches issued from the host, 0): same shape as IpcAllReduce.status()."""
        torch.cuda.synchronize(self.device)
        return self._launches, 0

    def close(self):
        if getattr(self, '_comm', None) is not None:
            _lib.load().rlg_rccl_comm_destroy(self._comm)
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
