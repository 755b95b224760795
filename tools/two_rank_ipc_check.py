"""Launched by torch.distributed.run with RLG_TEST_SINGLE_GPU=1 (N ranks on ONE GPU; gloo only for the
hand-shake): exercises the in-graph IPC all-reduce (csrc/ipc_allreduce.hip) - eager launches, launches
captured in a HIP graph and replayed, odd sizes - and checks every rank against the rank-ordered fp32
sum computed locally from all ranks' seeded inputs, bit for bit."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
from rl_games_amd import distributed as rdist
from rl_games_amd.ipc_allreduce import IpcAllReduce

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
rdist.init_process_group(True)
dev = 'cuda:0'
N = 229_937                       # humanoid arena size class, not a multiple of 4
two_phase = os.environ.get('RLG_IPC_CHECK_TWO_PHASE', '0') != '0'      # the reduce-scatter + all-gather kernel
comm = IpcAllReduce(N, dev, two_phase=two_phase)
launches0, _ = comm.status()       # the communicator's known-answer self-test (2 launches)


def contribution(r, it, n):
    g = torch.Generator().manual_seed(1000 * it + r)
    return torch.randn(n, generator=g)


def expected(it, n):
    s = contribution(0, it, n).clone()
    for r in range(1, world):
        s += contribution(r, it, n)            # rank order, fp32: what the kernel does
    return s


ok = True
t = torch.empty(N, device=dev)
for it in range(40):
    n = N if it % 3 else N - 5 * (it + 1)
    t[:n].copy_(contribution(rank, it, n))
    comm.all_reduce_sum(t[:n])
    ok &= bool(torch.equal(t[:n].cpu(), expected(it, n)))
# captured: 4 all-reduces per graph, replayed 3 times on fresh inputs
srcs = [torch.empty(N, device=dev) for _ in range(4)]
outs = [torch.empty(N, device=dev) for _ in range(4)]
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode='thread_local'):
    for k in range(4):
        t.copy_(srcs[k])
        comm.all_reduce_sum(t)
        outs[k].copy_(t)
for rep in range(3):
    for k in range(4):
        srcs[k].copy_(contribution(rank, 100 + 10 * rep + k, N))
    g.replay()
    torch.cuda.synchronize()
    for k in range(4):
        ok &= bool(torch.equal(outs[k].cpu(), expected(100 + 10 * rep + k, N)))
# by-product: per-block sums of (reduced x * scale)^2 over the leading n_grads elements + the step counter
partials = torch.full((comm.norm_blocks() + 2,), float('nan'), dtype=torch.float64, device=dev)
counter = torch.tensor([7], dtype=torch.int64, device=dev)
n_grads = N - 4                                         # the arena's tail slots do not count
for it in range(200, 203):
    t.copy_(contribution(rank, it, N))
    comm.all_reduce_sum(t, norm=(partials, n_grads, 1.0 / world, counter))
    want = expected(it, N)
    ok &= bool(torch.equal(t.cpu(), want))
    sq = ((want[:n_grads] * (1.0 / world)).double() ** 2).sum()
    ok &= bool(torch.allclose(partials[:comm.norm_blocks()].sum().cpu(), sq, rtol=1e-12))
    ok &= bool(torch.isnan(partials[comm.norm_blocks():]).all())
ok &= counter.item() == 10
launches, timed_out = comm.status()
ok &= (timed_out == 0) and launches0 == 2 and launches == launches0 + 40 + 12 + 3
flag = torch.tensor([1.0 if ok else 0.0])
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
if rank == 0:
    print('IPC_ALLREDUCE_CHECK', 'ok' if flag.item() == 1.0 else 'FAILED', f'world {world} launches {launches} '
          f'timed_out {timed_out} fine_grained {comm.fine_grained} two_phase {comm.two_phase}', flush=True)
comm.close()
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if flag.item() == 1.0 else 1)
