"""Which quantity of which optimiser step differs first between ranks that share one GPU?  (round 5; the diagnosis of
profiles/r4_two_rank_sync.txt.)  Needs the instrumented library: tools/exp/build_trace_libs.sh, RLG_HIP_LIB=tools/exp/_build/
trace/lib.so.  Launch:  RLG_TEST_SINGLE_GPU=1 python -m torch.distributed.run --nproc-per-node W ... adam_trace_probe.py EPOCHS
(bench.py runs the same trace on its own job with RLG_BENCH_ADAM_TRACE=1)."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools', 'exp'))
import adam_trace  # noqa: E402
import bench  # noqa: E402
from rl_games_amd import distributed as rdist  # noqa: E402,F401
from rl_games_amd.agent import A2CAgent  # noqa: E402

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
device = torch.device('cuda:0')
torch.cuda.set_device(device)
dist.init_process_group('gloo')
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
flags = int(os.environ.get('PROBE_FLAGS', '1'))
envs = int(os.environ.get('PROBE_ENVS', '65536'))
mb = int(os.environ.get('PROBE_MB', '32768'))
params = bench.make_params('humanoid', envs // world, mb // world, device, multi_gpu=True)
params['config']['env_config']['seed'] = 1234 + rank
params['config'].update(json.loads(os.environ.get('RLG_BENCH_CONFIG', '{}')))
torch.manual_seed(42 + rank)
notrace = os.environ.get('PROBE_NOTRACE') == '1'          # the product library: in-sync verdict only
rows = None if notrace else adam_trace.install(device, 5 * (envs * 32 // mb) * epochs + 64, flags)

agent = A2CAgent('probe', params)
agent.init_tensors()
agent.obs = agent.env_reset()
agent.broadcast_parameters()
for ep in range(epochs):
    agent.update_epoch()
    agent.train_epoch()
torch.cuda.synchronize()
opt = agent.optimizer


def same(t):
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


res = dict(params=same(opt.flat_params), exp_avg=same(opt.exp_avg), exp_avg_sq=same(opt.exp_avg_sq), grads=same(opt.grads))
kind = 'adam_pack' if agent._adam_pack_chain() is not None else 'adam_step'
label = f'launch {kind} in_sync {res}'
if notrace:
    if rank == 0:
        print(f'RESULT world {world} steps {int(opt.step_counter.item())} {label} all {all(res.values())}', flush=True)
    if not all(res.values()):
        adam_trace.diff_report(agent, rank, world, out=lambda s: print(s, flush=True))
else:
    adam_trace.report(rows, int(opt.step_counter.item()), rank, world, flags, label=label, out=lambda s: print(s, flush=True))
dist.barrier()
dist.destroy_process_group()
