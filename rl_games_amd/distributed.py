"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl" is
RCCL on ROCm) / gloo for the CPU functional tests.

The reference's collectives and what replaces them:
  C3 all_reduce of torch.cat(grads) + copy back  (a2c_common.py:493-509)
  C4 all_reduce of the minibatch KL               (:1560)
  C6 broadcast of [lr, entropy_coef] + 2x .item() (:564-576)
      -> ONE in-place all_reduce(SUM) of the flat gradient arena whose tail slot carries the
         KL; every rank then derives the same learning rate on device (csrc/optim.hip).
  C7 merge_rank_stats / broadcast_rank_stats     (:61-93, :124-141)   -> same maths below
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


def stats_totals(m):
    """(count, sum x, sum x^2) equivalent of a RunningMeanStd state (a2c_common.py:43-47)."""
    return (m.count.clone(), m.running_mean * m.count, (m.running_var + m.running_mean ** 2) * m.count)


def seed_stats_sync_snapshot(m):
    """After loading stats from a checkpoint: what is there is shared history
    (a2c_common.py:50-58)."""
    m._stats_sync_snapshot = tuple(t.clone() for t in stats_totals(m))


def merge_rank_stats(m, all_reduce):
    """Pooled cross-rank merge of one RunningMeanStd via summed per-epoch moment DELTAS against
    the last merged snapshot (a2c_common.py:61-93).  `all_reduce(t)` must SUM t in place."""
    cur = stats_totals(m)
    prev = getattr(m, '_stats_sync_snapshot', None)
    if prev is None:
        deltas = [c.clone() for c in cur]
        base = [torch.zeros_like(c) for c in cur]
    else:
        deltas = [c - p for c, p in zip(cur, prev)]
        base = prev
    for t in deltas:
        all_reduce(t)
    n = base[0] + deltas[0]
    s1 = base[1] + deltas[1]
    s2 = base[2] + deltas[2]
    m.count.copy_(n)
    m.running_mean.copy_(s1 / n)
    m.running_var.copy_((s2 / n - m.running_mean ** 2).clamp_(min=1e-8))
    m._stats_sync_snapshot = (n.clone(), s1.clone(), s2.clone())


def broadcast_rank_stats(m, broadcast):
    """Every rank adopts rank 0's statistics (a2c_common.py:124-141)."""
    broadcast(m.count)
    broadcast(m.running_mean)
    broadcast(m.running_var)


def resolve_stats_sync_mode(mode):
    if mode not in STATS_SYNC_MODES:
        raise ValueError(f"multi_gpu_sync_stats_mode must be one of {STATS_SYNC_MODES}, got '{mode}'")
    return mode
