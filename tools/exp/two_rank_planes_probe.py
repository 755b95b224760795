"""2 ranks on one GPU (RLG_TEST_SINGLE_GPU=1, launched by torch.distributed.run), bench.py's humanoid job at world 2: after every
epoch compare parameters, Adam moments and the chain's bf16 weight planes across the ranks - which of them diverges first?"""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from rl_games_amd import distributed as rdist  # noqa: E402,F401
from rl_games_amd.agent import A2CAgent  # noqa: E402

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
device = torch.device('cuda:0')
torch.cuda.set_device(device)
dist.init_process_group('gloo')
w = bench.WORKLOADS['humanoid'] if hasattr(bench, 'WORKLOADS') else None
params = bench.make_params('humanoid', 65536 // world, 32768 // world, device, multi_gpu=True)
params['config']['env_config']['seed'] = 1234 + rank
params['config'].update(json.loads(os.environ.get('RLG_BENCH_CONFIG', '{}')))
torch.manual_seed(42 + rank)
agent = A2CAgent('probe', params)
agent.init_tensors()
agent.obs = agent.env_reset()
agent.broadcast_parameters()


def same(t):
    """bit-equality of a tensor across the ranks (byte checksums in int64 + sum of squares in fp64)"""
    b = t.contiguous().view(torch.uint8)
    pad = (-b.numel()) % 8
    if pad:
        b = torch.cat([b, torch.zeros(pad, dtype=torch.uint8, device=b.device)])
    q = b.view(torch.int64)
    probe = torch.stack([q.sum().double(), (q ^ (q >> 17)).sum().double(), (q % 1000003).sum().double()]).cpu()
    lo, hi = probe.clone(), probe.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return bool(torch.equal(lo, hi))


epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
every = os.environ.get('PROBE_SYNC', '1') != '0'          # 0: compare behind the last epoch only (no host sync in between)
timers = os.environ.get('PROBE_TIMERS', '0') == '1'       # 1: bench.py's kernel timers (HIP events on launches) in the last epoch
params['config']['gemm_tuning_online'] = True
for ep in range(epochs):
    if timers and ep == epochs - 1:
        agent.kernel_timers = {}
    agent.update_epoch()
    agent.train_epoch()
    if not every and ep < epochs - 1:
        continue
    torch.cuda.synchronize()
    opt = agent.optimizer
    chain = agent._engine.chain
    res = dict(params=same(opt.flat_params), exp_avg=same(opt.exp_avg), exp_avg_sq=same(opt.exp_avg_sq),
               planes=same(chain._plane_buffer()), grads=same(opt.grads))
    if rank == 0:
        print('epoch', ep, res, flush=True)
    if not all(res.values()):
        # where?  rank 1's arrays to rank 0
        for name, t in (('exp_avg_sq', opt.exp_avg_sq), ('params', opt.flat_params)):
            mine = t.detach().cpu()
            other = mine.clone()
            if rank == 0:
                dist.recv(other, src=1)
                d = (mine != other).nonzero().flatten()
                print(f'   {name}: {d.numel()} of {mine.numel()} elements differ; index range {int(d.min()) if d.numel() else None} .. '
                      f'{int(d.max()) if d.numel() else None}; first {d[:8].tolist()}; rel diff max '
                      f'{float(((mine - other).abs() / other.abs().clamp_min(1e-30)).max()):.2e}', flush=True)
                if d.numel():
                    segs = []
                    off = 0
                    for nme, prm in agent.model.named_parameters():
                        cnt = prm.numel()
                        k = int(((d >= off) & (d < off + cnt)).sum())
                        if k:
                            segs.append((nme, k, cnt))
                        off += cnt
                    print('   by parameter (arena order assumed = named_parameters order):', segs, flush=True)
            else:
                dist.send(mine, dst=0)
        break
dist.barrier()
dist.destroy_process_group()
