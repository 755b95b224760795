"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl" is
RCCL on ROCm) / gloo for the CPU functional tests.

The reference's collectives and what replaces them:
  C3 all_reduce of torch.cat(grads) + copy back  (a2c_common.py:493-509)
  C4 all_reduce of the minibatch KL               (:1560)
  C6 broadcast of [lr, entropy_coef] + 2x .item() (:564-576)
      -> ONE in-place all_reduce(SUM) of the flat gradient arena whose tail slot carries the
         KL; every rank then derives the same learning rate on device (csrc/optim.hip).
  C7 merge_rank_stats / broadcast_rank_stats     (:61-93, :124-141)
      -> StatsSync below: every normaliser of the agent is a segment of ONE flat fp64 buffer;
         pack launch, ONE collective, apply launch per epoch (the reference: three collectives and a
         dozen small fp64 torch ops per normaliser).  Kernels: csrc/running_stats.hip.
  C2 parameter broadcast, C8 exit flag            -> broadcast of the flat arena / a scalar.
Payloads are <= 1 MB: latency bound, so fewer collectives matter more than bandwidth.
"""
import os

import torch
import torch.distributed as dist

STATS_SYNC_MODES = ('pooled', 'broadcast')


def env_ranks():
    return (int(os.getenv('LOCAL_RANK', '0')), int(os.getenv('RANK', '0')),
            int(os.getenv('WORLD_SIZE', '1')))


def single_gpu_test_mode():
    """RLG_TEST_SINGLE_GPU=1 (tests only): every rank uses GPU 0 and the collectives go through
    gloo, so the whole multi-rank agent path can be exercised on a one-GPU box.  RCCL refuses two
    ranks on one device, which is why the production backend cannot be used for that."""
    return os.environ.get('RLG_TEST_SINGLE_GPU', '0') not in ('0', '')


def backend_for(device_is_gpu=True):
    return 'nccl' if (device_is_gpu and not single_gpu_test_mode()) else 'gloo'


def local_device_index(local_rank):
    return 0 if single_gpu_test_mode() else int(local_rank)


def init_process_group(device_is_gpu=True):
    if dist.is_initialized():
        return
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    _, rank, world = env_ranks()
    dist.init_process_group(backend_for(device_is_gpu), rank=rank, world_size=world)


def all_reduce_sum(t):
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def broadcast_from_rank0(t):
    dist.broadcast(t, src=0)
    return t


def ranks_in_sync(tensors):
    """True when `tensors` (fp32 arenas: parameters, Adam moments) hold the same BITS on every rank.  Each tensor is
    reduced to an exact checksum - the int64 sum of its bit patterns, order-independent - and the checksums are
    compared by ONE all-reduce (MAX of [c, -c] = [max c, -min c]).  The reference keeps its ranks aligned by
    construction and never checks (a2c_common.py:493-514); here the ranks run a hand-written in-graph collective and
    rank-ordered sums, so the agent asks once per epoch (`multi_gpu_param_check`) - a desynchronisation that is not a
    crash would otherwise only show as a slowly diverging policy."""
    sums = torch.stack([t.detach().contiguous().view(torch.int32).sum(dtype=torch.int64) for t in tensors])
    both = torch.cat([sums, -sums])
    dist.all_reduce(both, op=dist.ReduceOp.MAX)
    n = sums.numel()
    return bool((both[:n] == -both[n:]).all().item())


def resolve_stats_sync_mode(mode):
    if mode in STATS_SYNC_MODES:
        return mode
    raise ValueError(f"multi_gpu_sync_stats_mode must be one of {STATS_SYNC_MODES}, got '{mode}'")


class StatsSync:
    """Cross-rank synchronisation of a set of RunningMeanStd modules (row a19; behaviour of
    rl_games/common/a2c_common.py:43-141, tested like tests/test_multigpu_stats_sync.py).

    pooled mode (`merge`): every rank contributes the moment totals it accumulated since the last
    merge - count, sum x, sum x^2 of each normaliser, i.e. totals(state) minus the totals right
    after the previous merge (the "snapshot"; absent = the whole history is rank-local) - the
    summed deltas are added to the shared snapshot and turned back into mean / var.  History that
    all ranks already share (previous merges, a loaded checkpoint -> `seed`) is therefore counted
    once, not world_size times.
    broadcast mode (`adopt_rank0`): stateless, every rank takes rank 0's state.

    Layout: normaliser s with D_s features owns 1 + 2*D_s consecutive doubles of the flat buffers
    `deltas` / `snapshot`: [count | D_s first moments | D_s second moments].  One collective moves
    all of them.  `kernels` does the arithmetic on the device (ops.StatsSyncKernels: no host
    fallback); the CPU functional tests inject the oracle's restatement of the same layout."""

    def __init__(self, modules, kernels=None):
        self.modules = list(modules)
        if not self.modules:
            raise ValueError('StatsSync needs at least one normaliser')
        if kernels is None:
            from . import ops
            kernels = ops.StatsSyncKernels(self.modules)
        self.kernels = kernels
        ref = self.modules[0].running_mean
        size = kernels.flat_size
        self.snapshot = torch.zeros(size, dtype=torch.float64, device=ref.device)
        self.deltas = torch.zeros(size, dtype=torch.float64, device=ref.device)
        self.has_snapshot = [False] * len(self.modules)

    def covers(self, modules):
        """Same normalisers AND still the same state tensors: the kernels hold raw device pointers, and a module's
        buffers can be rebound behind its identity (load_state_dict(assign=True), .to(), a manual swap)."""
        modules = list(modules)
        if len(modules) != len(self.modules) or not all(a is b for a, b in zip(modules, self.modules)):
            return False
        bound = getattr(self.kernels, 'bound_to', None)
        return True if bound is None else bound(modules)

    def seed(self):
        """The current state is shared history (a checkpoint was loaded on every rank)."""
        k = self.kernels
        k.pack(self.has_snapshot, self.snapshot, self.deltas, k.PACK_SEED)
        self.has_snapshot = [True] * len(self.modules)

    def merge(self, all_reduce):
        """`all_reduce(t)` must SUM the fp64 tensor t in place across ranks."""
        k = self.kernels
        k.pack(self.has_snapshot, self.snapshot, self.deltas, k.PACK_DELTAS)
        all_reduce(self.deltas)
        k.apply(self.has_snapshot, self.snapshot, self.deltas, k.APPLY_MERGE)
        self.has_snapshot = [True] * len(self.modules)

    def adopt_rank0(self, broadcast):
        """`broadcast(t)` must overwrite the fp64 tensor t with rank 0's in place."""
        k = self.kernels
        k.pack(self.has_snapshot, self.snapshot, self.deltas, k.PACK_STATE)
        broadcast(self.deltas)
        k.apply(self.has_snapshot, self.snapshot, self.deltas, k.APPLY_STATE)


def _sync_of(m):
    """The single-module StatsSync behind the reference's per-module function seams."""
    sync = getattr(m, '_stats_sync', None)
    if sync is None or not sync.covers([m]):
        sync = StatsSync([m])
        m._stats_sync = sync
    return sync


def seed_stats_sync_snapshot(m):
    """Function seam of a2c_common.py:50-58 for one normaliser."""
    _sync_of(m).seed()


def merge_rank_stats(m, all_reduce):
    """Function seam of a2c_common.py:61-93 for one normaliser (one collective instead of three)."""
    _sync_of(m).merge(all_reduce)


def broadcast_rank_stats(m, broadcast):
    """Function seam of a2c_common.py:124-141 for one normaliser (one collective instead of three)."""
    _sync_of(m).adopt_rank0(broadcast)
