"""In-graph gradient all-reduce over peer-mapped device memory (csrc/ipc_allreduce.hip).

One process per GPU; every rank allocates its staging buffers through the C ABI, the ranks exchange
the 64-byte hipIpcMemHandle_t of those buffers once through the existing torch.distributed group
(RCCL or gloo - only used for this hand-shake), map each other's memory and from then on call
`all_reduce_sum(flat_tensor)`: ONE kernel launch on the current stream, no host synchronisation, safe
to capture in a HIP graph.  Replaces dist.all_reduce of the flat gradient arena
(rl_games/common/a2c_common.py:493-509, KL slot :1559-1560) for the <= 1 MB arenas of the
BASELINE configs; the RCCL path (rl_games_amd.distributed.all_reduce_sum) stays available behind the
same agent call (`native_allreduce: False`).
"""
import ctypes

import torch
import torch.distributed as dist

from . import _lib


class IpcAllReduce:
    def __init__(self, numel, device, rank=None, world=None, group=None, timeout_s=None, two_phase=None):
        """timeout_s: bound of a launch's wait for its peers (None: RLG_IPC_TIMEOUT_S or 600 s; <= 0: unbounded).
        two_phase: reduce-scatter + all-gather variant (None: RLG_IPC_TWO_PHASE or the one-shot variant)."""
        lib = _lib.load()
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.numel = int(numel)
        self.device = torch.device(device)
        nb = lib.rlg_ipc_handle_bytes()
        handle = (ctypes.c_char * nb)()
        comm = ctypes.c_void_p()
        self._comm = None
        # Every step below is collective: a rank that fails still takes part in the exchange, and the
        # ranks agree on the outcome, so either all of them end up with a working communicator or all raise.
        with torch.cuda.device(self.device):
            err = lib.rlg_ipc_comm_create(self.rank, self.world, self.numel, ctypes.byref(comm), handle)
            handles = [None] * self.world
            dist.all_gather_object(handles, bytes(handle) if err == 0 else None, group=group)
            if err == 0:
                self._comm = comm
            if any(h is None for h in handles):
                self.close()
                raise _lib.HipLibraryError(f'rlg_ipc_comm_create failed on a rank (local hipError_t {err})')
            err = lib.rlg_ipc_comm_connect(self._comm, b''.join(handles))
            oks = [None] * self.world
            dist.all_gather_object(oks, err, group=group)     # also: every rank has mapped every buffer
            if any(oks):
                self.close()
                raise _lib.HipLibraryError(f'rlg_ipc_comm_connect failed on a rank (hipError_t per rank: {oks})')
        self.fine_grained = bool(lib.rlg_ipc_comm_fine_grained(self._comm))     # (creation fails without it)
        if timeout_s is not None:
            lib.rlg_ipc_comm_set_timeout(self._comm, float(timeout_s))
        if two_phase is not None:
            lib.rlg_ipc_comm_set_variant(self._comm, int(bool(two_phase)))
        # what the library will actually run (RLG_IPC_TWO_PHASE / RLG_IPC_TIMEOUT_S are per-process defaults): the
        # variant is a collective choice, and ranks with different bounds would give up at different times
        variant, bound = ctypes.c_int(), ctypes.c_double()
        lib.rlg_ipc_comm_get_config(self._comm, ctypes.byref(variant), ctypes.byref(bound))
        self.two_phase = bool(variant.value)
        self.timeout_s = float(bound.value)
        settings = [None] * self.world
        dist.all_gather_object(settings, (self.two_phase, self.timeout_s), group=group)
        if any(s != settings[0] for s in settings):
            self.close()
            raise _lib.HipLibraryError(f'in-graph all-reduce: ranks disagree on (two_phase, timeout_s): {settings}')
        word = ctypes.c_void_p()
        lib.rlg_ipc_comm_error_word(self._comm, ctypes.byref(word))
        self.error_word = int(word.value)      # device address: FlatAdam.step(skip_flag=...)
        self._self_test(group)

    def _self_test(self, group):
        """Two known-answer all-reduces (one per staging slot) before the communicator is handed out:
        mapping peers' memory can succeed on a topology where the flags or the data do not become visible
        across devices in time.  Collective like the rest of the set-up - all ranks keep the communicator
        or all raise (the agent then uses RCCL)."""
        n = min(self.numel, 4096)
        ok = True
        try:
            for it in range(2):
                t = torch.arange(n, dtype=torch.float32, device=self.device) * 0.5 + float(self.rank + 1 + it)
                self.all_reduce_sum(t)
                want = torch.arange(n, dtype=torch.float32, device=self.device) * (0.5 * self.world) + float(
                    self.world * (self.world + 1) // 2 + it * self.world)
                ok = ok and bool(torch.equal(t, want))
            _, timed_out = self.status()
            ok = ok and timed_out == 0
        except Exception:
            ok = False
        oks = [None] * self.world
        dist.all_gather_object(oks, ok, group=group)
        if not all(oks):
            self.close()
            raise _lib.HipLibraryError(f'in-graph all-reduce self-test failed (per rank: {oks})')

    def all_reduce_sum(self, t, norm=None):
        """In place, on torch's current stream.  `t`: contiguous fp32 CUDA tensor of <= numel elements.
        norm = (partials fp64 [>= norm_blocks()], n_grads, grad_scale, step_counter int64 [1]): the launch
        also leaves the per-block sums of (reduced x * grad_scale)^2 over the first n_grads elements and
        advances the Adam step counter (FlatAdam.step(norm_ready=(partials, norm_blocks())))."""
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() > self.numel:
            raise ValueError('IpcAllReduce: contiguous fp32 tensor of at most the planned size expected')
        _lib.require_gpu(t, 'all_reduce_sum')
        lib = _lib.load()
        if norm is None:
            _lib.check(lib.rlg_ipc_allreduce_sum(self._comm, t.data_ptr(), t.numel(), _lib.stream_handle(t.device)),
                       'rlg_ipc_allreduce_sum')
        else:
            partials, n_grads, scale, counter = norm
            if partials.dtype != torch.float64 or partials.numel() < self.norm_blocks() or counter.dtype != torch.int64:
                raise ValueError('IpcAllReduce: norm partials fp64 [norm_blocks] and an int64 step counter expected')
            _lib.check(lib.rlg_ipc_allreduce_sum_norm(self._comm, t.data_ptr(), t.numel(), partials.data_ptr(),
                                                      int(n_grads), float(scale), counter.data_ptr(),
                                                      _lib.stream_handle(t.device)), 'rlg_ipc_allreduce_sum_norm')
        return t

    @staticmethod
    def norm_blocks():
        return _lib.load().rlg_ipc_allreduce_norm_blocks()

    def status(self):
        """(launches completed, ordinal of a launch that gave up waiting for a peer or 0).  Synchronises."""
        torch.cuda.synchronize(self.device)
        a, b = ctypes.c_uint(), ctypes.c_uint()
        _lib.check(_lib.load().rlg_ipc_comm_status(self._comm, ctypes.byref(a), ctypes.byref(b)), 'rlg_ipc_comm_status')
        return int(a.value), int(b.value)

    def close(self):
        """Unmaps the peers' buffers and frees this rank's.  Collective in spirit: every rank must be past its last
        launch on this communicator.  Do not create ANOTHER communicator in the same process afterwards and expect the
        first one's memory to be gone from the peers' view: creating and destroying two communicators in front of the
        agent's own made one 2-rank run in eight end with parameters that differed between the ranks in the 8th digit
        (profiles/r4_two_rank_sync.txt).  The agent holds one communicator for its lifetime."""
        if getattr(self, '_comm', None) is not None:
            _lib.load().rlg_ipc_comm_destroy(self._comm)
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
