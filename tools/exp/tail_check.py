"""Fused step tail (finalise + Adam in one launch) against the launch pair at a shape whose finalise grid exceeds the
persistent grid (several finalise blocks per workgroup), and the 3-epoch drift of config #2 against the oracle with
either form.   python tools/exp/tail_check.py"""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from oracle.ppo_epoch_oracle import OracleAgent
from rl_games_amd import configs
from rl_games_amd.agent import A2CAgent
from rl_games_amd.synthetic_env import SyntheticTensorEnv

def run(fused, graphs, epochs, cfg=configs.ant_4096, **kw):
    params = cfg(hip_graphs=graphs, fused_step_tail=fused, **kw)
    torch.manual_seed(9)
    agent = A2CAgent('t', copy.deepcopy(params))
    agent.init_tensors(); agent.obs = agent.env_reset()
    caps = []
    orig = agent.play_steps
    def play():
        b = orig()
        caps.append({'batch': {k: v.detach().cpu().clone() for k, v in b.items() if isinstance(v, torch.Tensor)},
                     'state': {k: v.detach().cpu().clone() for k, v in agent.model.state_dict().items()}})
        return b
    agent.play_steps = play
    for _ in range(epochs):
        agent.update_epoch(); agent.train_epoch()
    return agent, caps, params

for graphs in (False, True):
    a, _, _ = run(True, graphs, 2)
    b, _, _ = run(False, graphs, 2)
    for name in ('flat_params', 'exp_avg', 'exp_avg_sq', 'grads', 'stats'):
        x, y = getattr(a.optimizer, name), getattr(b.optimizer, name)
        d = (x - y).abs()
        print(f'graphs {graphs} fused vs pair {name:12s} max|d| {d.max().item():.3e}  mean|d| {d.mean().item():.3e}  differing {int((d > 0).sum())}/{d.numel()}')
    print('  lr', a.optimizer.last_and_next_lr(), b.optimizer.last_and_next_lr(), 'steps', a.optimizer.step_count, b.optimizer.step_count)

for fused in (True, False):
    agent, caps, params = run(fused, True, 3)
    cpu = copy.deepcopy(params); cpu['config']['device'] = 'cpu'
    torch.set_num_threads(16)
    oracle = OracleAgent(cpu, SyntheticTensorEnv(4096, 60, 8, device='cpu', seed=1))
    oracle.model.load_full_state_dict(caps[0]['state'])
    for e in range(3):
        oracle.update(caps[e]['batch'])
    final, want = agent.model.state_dict(), oracle.model.full_state_dict()
    print(f'drift vs oracle after 3 epochs, fused_step_tail={fused}: lr agent {agent.optimizer.last_and_next_lr()[1]} oracle {oracle.lr}')
    for name, v in want.items():
        if not v.is_floating_point():
            continue
        got = final[name].cpu().to(v.dtype)
        rel = ((got - v).abs().mean() / v.abs().mean().clamp_min(1e-12)).item()
        print(f'   {name:42s} mean|d|/mean|x| {rel:.2e} max|d| {(got - v).abs().max().item():.2e}')
