"""How far the agent's 320 optimiser steps of one epoch drift from the oracle's, per mini-epoch (round 4: the
bounds the full-epoch parity tests in tests/test_headline_gpu.py assert were read off this script's output).
    python tools/exp/parity_drift.py rank|full [threads]"""
import copy, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from oracle.ppo_epoch_oracle import OracleAgent
from rl_games_amd import configs
from rl_games_amd.agent import A2CAgent
from rl_games_amd.synthetic_env import SyntheticTensorEnv

which = sys.argv[1] if len(sys.argv) > 1 else 'rank'
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
N, MB = (8192, 4096) if which == 'rank' else (65536, 32768)
params = configs.humanoid_65536(num_actors=N, minibatch_size=MB, hip_graphs=True)
torch.manual_seed(5)
agent = A2CAgent('drift', copy.deepcopy(params))
agent.init_tensors(); agent.obs = agent.env_reset()
caps = []
orig = agent.play_steps
def play():
    b = orig()
    caps.append({'batch': {k: v.detach().cpu().clone() for k, v in b.items() if isinstance(v, torch.Tensor)},
                 'state': {k: v.detach().cpu().clone() for k, v in agent.model.state_dict().items()}})
    return b
agent.play_steps = play
agent._eager_epochs = 1
agent.update_epoch()
res = agent.train_epoch()
NMB = len(agent.dataset); ME = agent.mini_epochs_num
rows = agent._mb_scalars[:ME * NMB].cpu()
torch.set_num_threads(max(1, min(threads, os.cpu_count() or 1)))
cpu_params = copy.deepcopy(params); cpu_params['config']['device'] = 'cpu'
oracle = OracleAgent(cpu_params, SyntheticTensorEnv(N, 108, 21, device='cpu', seed=1))
oracle.model.load_full_state_dict(caps[0]['state'])
t0 = time.perf_counter()
ref = oracle.update(caps[0]['batch'])
print(f'{which}: N {N} minibatch {MB}: oracle update {time.perf_counter() - t0:.1f} s with {torch.get_num_threads()} threads, {len(ref)} steps')
# the same reference algorithm on inputs that differ in the LAST BIT: every observation moved one ulp up or down at random
# (what any two fp32 implementations of the forward differ by); and on the same inputs with the rows inside every minibatch
# reversed (other summation order only)
def last_bit(batch, seed=1):
    out = dict(batch)
    x = batch['obses']
    g = torch.Generator().manual_seed(seed)
    up = torch.rand(x.shape, generator=g) < 0.5
    out['obses'] = torch.where(up, torch.nextafter(x, torch.full_like(x, float('inf'))), torch.nextafter(x, torch.full_like(x, float('-inf'))))
    return out
def reversed_rows(batch, mb):
    out = {}
    for k, v in batch.items():
        if isinstance(v, torch.Tensor) and v.dim() >= 1 and v.shape[0] % mb == 0 and v.shape[0] >= mb:
            out[k] = v.reshape(v.shape[0] // mb, mb, *v.shape[1:]).flip(1).reshape(v.shape).clone()
        else:
            out[k] = v
    return out
twins = {}
for label, b in (('last-bit twin', last_bit(caps[0]['batch'])), ('reversed-rows twin', reversed_rows(caps[0]['batch'], MB))):
    tw = OracleAgent(cpu_params, SyntheticTensorEnv(N, 108, 21, device='cpu', seed=1))
    tw.model.load_full_state_dict(caps[0]['state'])
    twins[label] = (tw, tw.update(b))
names = ('a_loss', 'c_loss', 'entropy', 'b_loss', 'kl')
lr_agent_traj = None
for m in range(ME):
    sl = slice(m * NMB, (m + 1) * NMB)
    line = [f'mini-epoch {m + 1}:']
    for col, key in enumerate(names):
        want = torch.stack([r[key].reshape(()) for r in ref[sl]])
        got = rows[sl, col]
        d = (got - want).abs()
        rel = (d / want.abs().clamp_min(1e-30)).max().item()
        line.append(f'{key} max|d| {d.max().item():.2e} maxrel {rel:.2e} (|x|~{want.abs().mean().item():.2e})')
    print('  '.join(line))
    for label, (tw, r2) in twins.items():
        line = [f'   {label:18s}:']
        for col, key in enumerate(names):
            want = torch.stack([r[key].reshape(()) for r in ref[sl]])
            other = torch.stack([r[key].reshape(()) for r in r2[sl]])
            line.append(f'{key} max|d| {(other - want).abs().max().item():.2e}')
        print('  '.join(line))
lrs = [r['lr'] for r in ref]
print('oracle lr changes at steps', [k for k in range(1, len(lrs)) if lrs[k] != lrs[k - 1]][:20], 'final', oracle.lr)
print('agent  lr (last used, next)', agent.optimizer.last_and_next_lr())
kth = params['config']['kl_threshold']
kls = torch.tensor([float(r['kl']) for r in ref])
margin = torch.minimum((kls / (2 * kth) - 1).abs(), (kls / (0.5 * kth) - 1).abs())
print(f'closest oracle KL to a threshold: relative margin {margin.min().item():.2e} at step {int(margin.argmin())}')
final, want = agent.model.state_dict(), oracle.model.full_state_dict()
for name, v in want.items():
    if not v.is_floating_point() or v.numel() < 16:
        continue
    got = final[name].cpu().to(v.dtype)
    rel = ((got - v).abs().mean() / v.abs().mean().clamp_min(1e-12)).item()
    mx = (got - v).abs().max().item()
    tws = '  '.join(f'{label}: {((tw.model.full_state_dict()[name] - v).abs().mean() / v.abs().mean().clamp_min(1e-12)).item():.2e}'
                    for label, (tw, _) in twins.items())
    print(f'  param {name:45s} mean|d|/mean|x| {rel:.2e}  max|d| {mx:.2e}   {tws}')
