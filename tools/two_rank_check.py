"""Launched by torch.distributed.run with RLG_TEST_SINGLE_GPU=1 (2 ranks on one GPU, gloo): trains a
small agent of the requested kind with multi_gpu=True for a few epochs and checks that every rank
ends with bit-identical parameters, normaliser statistics and learning rate."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
from rl_games_amd import configs, distributed as rdist

kind = sys.argv[1]
rank = int(os.environ['RANK'])
torch.manual_seed(100 + rank)
if kind == 'lstm':
    from rl_games_amd.agent import A2CAgent as Agent
    params = configs.pendulum_lstm_4096(num_actors=64, multi_gpu=True)
elif kind == 'discrete':
    from rl_games_amd.discrete_agent import DiscreteA2CAgent as Agent
    params = configs.cartpole_discrete(num_actors=16, multi_gpu=True, normalize_input=True, normalize_value=True,
                                       lr_schedule='adaptive')
elif kind == 'central_value':
    from rl_games_amd.agent import A2CAgent as Agent
    params = configs.tiny(num_actors=64, horizon=8, multi_gpu=True)
    params['config']['central_value_config'] = {
        'minibatch_size': 128, 'mini_epochs': 2, 'learning_rate': 5e-4, 'clip_value': True, 'normalize_input': True,
        'truncate_grads': True, 'grad_norm': 1.0,
        'network': {'name': 'actor_critic', 'central_value': True,
                    'mlp': {'units': [32, 16], 'activation': 'elu', 'initializer': {'name': 'default'}}}}
    params['config']['env_config']['state_dim'] = 9
else:
    from rl_games_amd.agent import A2CAgent as Agent
    params = configs.tiny(num_actors=64, horizon=8, multi_gpu=True)
params['config']['env_config']['seed'] = 10 + rank          # different data per rank
# RLG_TWO_RANK_CONFIG='{"native_allreduce": false}': the torch.distributed fallback (RCCL in production, gloo in this
# one-GPU test mode) through the same agent code; '{"native_allreduce_two_phase": true}': the reduce-scatter +
# all-gather variant of the in-graph kernel
import json
params['config'].update(json.loads(os.environ.get('RLG_TWO_RANK_CONFIG', '{}')))
agent = Agent('mr', params)
agent.init_tensors()
agent.obs = agent.env_reset()
agent.broadcast_parameters()
for _ in range(4):
    agent.update_epoch()
    agent.train_epoch()
probes = [agent.optimizer.flat_params.double().sum(), agent.optimizer.flat_params.double().abs().sum(),
          torch.tensor(float(agent.optimizer.last_and_next_lr()[1]), dtype=torch.float64, device=agent.ppo_device)]
for m in agent._stats_sync_modules():
    probes += [m.running_mean.double().sum(), m.running_var.double().sum(), m.count.double()]
if agent.has_central_value:
    probes.append(agent.central_value_net.optimizer.flat_params.double().sum())
p = torch.stack([x.reshape(()) for x in probes])
lo, hi = p.clone(), p.clone()
dist.all_reduce(lo, op=dist.ReduceOp.MIN)
dist.all_reduce(hi, op=dist.ReduceOp.MAX)
ok = bool(torch.equal(lo, hi)) and bool(torch.isfinite(p).all())
if rank == 0:
    phase = 'two_phase' if (agent._ipc_comm and agent._ipc_comm.two_phase) else 'one_shot'
    verdict = 'in_sync' if ok else f'OUT_OF_SYNC {lo.tolist()} {hi.tolist()}'
    sys.stdout.write(f'TWO_RANK_ALLREDUCE {agent.last_allreduce} {phase}\nTWO_RANK_CHECK {kind} {verdict}\n')     # (one write)
    sys.stdout.flush()
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
