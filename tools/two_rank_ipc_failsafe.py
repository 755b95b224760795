"""Launched by torch.distributed.run with RLG_TEST_SINGLE_GPU=1 (2 ranks on ONE GPU): the in-graph all-reduce when a
peer never arrives.  Rank 1 skips one collective; rank 0's launch must give up after its (short, test-only) bound and be
FAIL-SAFE: zeros instead of a sum of stale staging data, the sticky error word set, the Adam launch behind it skipping
its step (parameters, moments, learning-rate slot untouched), every later launch failing fast."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
from rl_games_amd import distributed as rdist, ops
from rl_games_amd.ipc_allreduce import IpcAllReduce

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
rdist.init_process_group(True)
dev = 'cuda:0'
N = 10_000
comm = IpcAllReduce(N, dev, timeout_s=2.0, two_phase=os.environ.get('RLG_IPC_CHECK_TWO_PHASE', '0') != '0')
ok = True
t = torch.full((N,), float(rank + 1), device=dev)
comm.all_reduce_sum(t)                                   # a healthy one first
ok &= bool((t == 3.0).all())
if rank == 0:
    params = torch.randn(N, device=dev)
    grads = torch.full((N,), 123.0, device=dev)
    m, v = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
    lr_slots = torch.tensor([3e-4, -1.0], dtype=torch.float64, device=dev)
    counter = torch.tensor([1], dtype=torch.int64, device=dev)
    before = params.clone()
    t0 = time.time()
    comm.all_reduce_sum(grads)                           # rank 1 never joins this one
    ops.adam_step(params, grads, m, v, None, 0.5, 0.0, lr_slots, counter, skip_flag=comm.error_word)
    launches, timed_out = comm.status()                  # (synchronises)
    waited = time.time() - t0
    ok &= timed_out != 0 and 1.0 < waited < 30.0
    ok &= bool((grads == 0).all())                       # zeros, not stale staging data
    ok &= bool(torch.equal(params, before)) and bool((m == 0).all()) and bool((v == 0).all())
    ok &= lr_slots[1].item() == 3e-4                     # the learning rate is carried over
    t0 = time.time()
    comm.all_reduce_sum(grads)                           # sticky: fails fast, still zeros
    _, again = comm.status()
    ok &= again != 0 and time.time() - t0 < 1.0 and bool((grads == 0).all())
    print('IPC_FAILSAFE_CHECK', 'ok' if ok else f'FAILED timed_out {timed_out} waited {waited:.1f}', flush=True)
else:
    time.sleep(6.0)                                      # "a rank that stalled": rank 0 has given up by now
flag = torch.tensor([1.0 if ok else 0.0])
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
comm.close()
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if flag.item() == 1.0 else 1)
