"""Launched by torch.distributed.run with RLG_TEST_SINGLE_GPU=1 (2 ranks on one GPU, gloo for the hand-shakes): the
multi_gpu optimiser steps of the agent against the CPU oracle's data-parallel restatement of
A2CBase.trancate_gradients_and_step (rl_games/common/a2c_common.py:493-514; oracle/ppo_epoch_oracle.py,
data_parallel_minibatch_step - itself pinned to the reference method under gloo, tests/test_distributed_cpu.py).

Every rank plays its own env shard, the ranks exchange rollouts + model state (CPU copies), every rank builds BOTH oracle
ranks and steps them together; the device agent steps through the same minibatches with its in-graph gradient all-reduce
(or the torch.distributed fallback: RLG_TWO_RANK_CONFIG).  Compared on each rank, per optimiser step of the first mini-epoch:
  * this rank's losses / KL against its oracle rank (rtol 1e-5 + the floors of tests/test_headline_gpu.py),
  * the gradient arena behind the step - the SUM over the ranks / world_size, clipped - against the oracle's averaged,
    clipped gradients, tensor by tensor to 1e-5 of the tensor's scale,
  * the parameters behind the step against the oracle's (Adam on those gradients), and the learning rate (exact),
and at the end that both device ranks hold the same parameters bit for bit."""
import copy
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist  # noqa: E402
from oracle.ppo_epoch_oracle import OracleAgent, data_parallel_minibatch_step  # noqa: E402
from rl_games_amd import configs  # noqa: E402
from rl_games_amd.agent import A2CAgent  # noqa: E402
from rl_games_amd.synthetic_env import SyntheticTensorEnv  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else 'tiny'
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.manual_seed(100 + rank)
if shape == 'humanoid':
    # a rank of 8's minibatch of BASELINE config #4: 4,096 rows through the lean 16-row kernels
    N, H, OBS, ACT = 512, 32, 108, 21
    params = configs.humanoid_65536(num_actors=N, minibatch_size=4096, multi_gpu=True, hip_graphs=False)
    params['config']['mini_epochs'] = 1
else:
    N, H, OBS, ACT = 128, 8, 12, 3
    params = configs.tiny(num_actors=N, horizon=H, multi_gpu=True, hip_graphs=False)
params['config']['env_config']['seed'] = 10 + rank          # different data per rank
params['config'].update(json.loads(os.environ.get('RLG_TWO_RANK_CONFIG', '{}')))
params['config'].setdefault('native_allreduce_timeout_s', 60.0)      # (a peer that died: give up, do not hang the box)
agent = A2CAgent('dp', copy.deepcopy(params))
agent.init_tensors()
agent.obs = agent.env_reset()
agent.broadcast_parameters()
agent.set_eval()
with torch.no_grad():
    batch = agent.play_steps()
state = {k: v.detach().cpu().clone() for k, v in agent._plain_model().state_dict().items()}
cpu_batch = {k: v.detach().cpu().clone() for k, v in batch.items() if isinstance(v, torch.Tensor)}
gathered = [None] * world
dist.all_gather_object(gathered, (state, cpu_batch))

oracles = []
for r, (st, cb) in enumerate(gathered):
    cpu_params = copy.deepcopy(params)
    cpu_params['config']['device'] = 'cpu'
    o = OracleAgent(cpu_params, SyntheticTensorEnv(N, OBS, ACT, device='cpu', seed=1))
    o.model.load_full_state_dict(st)
    o.prepare_dataset(cb)
    oracles.append(o)
for p, q in zip(oracles[0].model.a2c_network.parameters(), oracles[1].model.a2c_network.parameters()):
    assert torch.equal(p, q), 'broadcast_parameters left the ranks with different weights'

ATOL = {'a_loss': 2e-6, 'c_loss': 2e-6, 'entropy': 2e-6, 'b_loss': 2e-7, 'kl': 2e-6}
agent.set_train()
agent.prepare_dataset(batch)
problems = []
solid, lr_sum = {}, 0.0
nmb = len(agent.dataset)
names = [n for n, _ in agent._plain_model().named_parameters()]
for i in range(nmb):
    a, c, e, kl, lr, lr_mul, mu, sigma, b = agent.train_actor_critic(agent.dataset[i])
    want = data_parallel_minibatch_step(oracles, i)
    mine = want[rank]
    for got, key in ((a, 'a_loss'), (c, 'c_loss'), (e, 'entropy'), (b, 'b_loss')):
        if not np.isclose(got.item(), mine[key].item(), rtol=1e-5, atol=ATOL[key]):
            problems.append((i, key, got.item(), mine[key].item()))
    # the KL slot of the arena went through the collective with the gradients: what the lr rule saw is the mean
    if not np.isclose(kl.item(), mine['kl'].item(), rtol=1e-4, atol=ATOL['kl']):
        problems.append((i, 'kl', kl.item(), mine['kl'].item()))
    ref = dict(oracles[rank].model.a2c_network.named_parameters())
    lr_sum += mine['lr']
    for name, p in agent._plain_model().named_parameters():
        q = ref[name.replace('a2c_network.', '')]
        g, g_ref = p.grad.detach().cpu(), q.grad
        scale = g_ref.abs().max().item()
        if (g - g_ref).abs().max().item() > 1e-5 * scale + 1e-9:
            problems.append((i, 'grad ' + name, (g - g_ref).abs().max().item(), scale))
        # parameters behind i + 1 Adam steps.  An element's normalised step m / (sqrt(v) + eps) is determined as well as its
        # gradients are: where every gradient so far was solid (>= 1 % of its tensor's scale, i.e. known to 1e-3) the two
        # trajectories agree to a few 1e-3 of the accumulated step; elsewhere (gradients at rounding-noise level: the very
        # first step is lr * sign(g)) only the bound of Adam itself holds, |delta| <= ~lr per step on both sides.
        solid[name] = solid.get(name, torch.ones_like(g_ref, dtype=torch.bool)) & (g_ref.abs() > 1e-2 * scale)
        diff = (p.detach().cpu() - q.detach()).abs()
        if solid[name].any() and diff[solid[name]].max().item() > 5e-3 * lr_sum + 1e-7:
            problems.append((i, 'param ' + name, diff[solid[name]].max().item(), lr_sum))
        if diff.max().item() > 2.5 * lr_sum:
            problems.append((i, 'param bound ' + name, diff.max().item(), lr_sum))
    used, nxt = agent.optimizer.last_and_next_lr()
    if nxt != oracles[rank].lr:
        problems.append((i, 'lr', nxt, oracles[rank].lr))

n_solid = sum(int(m.sum()) for m in solid.values())
n_all = sum(m.numel() for m in solid.values())
if n_solid < 0.2 * n_all:
    problems.append(('solid elements', n_solid, n_all))         # (the parameter comparison must not be vacuous)
flat = agent.optimizer.flat_params
lo, hi = flat.clone(), flat.clone()
dist.all_reduce(lo, op=dist.ReduceOp.MIN)
dist.all_reduce(hi, op=dist.ReduceOp.MAX)
in_sync = bool(torch.equal(lo, hi))
ok = in_sync and not problems
oks = [None] * world
dist.all_gather_object(oks, ok)
# (one write per line, the verdict behind a barrier: two ranks share the pipe, and print() hands its arguments over piece
#  by piece - a test once read "TWO_RANK_ORACLE_CHECK \nhumanoid ok" with the other rank's newline in the middle)
sys.stdout.write(f'TWO_RANK_ORACLE rank {rank} {shape} steps {nmb} allreduce {agent.last_allreduce} in_sync {in_sync} '
                 f'problems {problems[:6]}\n')
sys.stdout.flush()
dist.barrier()
if rank == 0:
    sys.stdout.write(f"TWO_RANK_ORACLE_CHECK {shape} {'ok' if all(oks) else 'FAILED'}\n")
    sys.stdout.flush()
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if all(oks) else 1)
