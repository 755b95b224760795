"""Is the agent's rollout (mus, values stored in the buffer) what the oracle's network computes from the same
observations and the model state the rollout was played with?  Per epoch: max |mu_rollout - mu_oracle| etc., and the
3-epoch parameter drift.   python tools/exp/rollout_consistency.py   (variants by environment, e.g. RLG_CHAIN_PIPE1=0)"""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from oracle.ppo_epoch_oracle import OracleAgent
from rl_games_amd import configs
from rl_games_amd.agent import A2CAgent
from rl_games_amd.synthetic_env import SyntheticTensorEnv

params = configs.ant_4096(hip_graphs=True)
torch.manual_seed(9)
agent = A2CAgent('t', copy.deepcopy(params))
agent.init_tensors(); agent.obs = agent.env_reset()
caps = []
orig = agent.play_steps
def play():
    state = {k: v.detach().cpu().clone() for k, v in agent.model.state_dict().items()}     # BEFORE the rollout
    b = orig()
    caps.append({'batch': {k: v.detach().cpu().clone() for k, v in b.items() if isinstance(v, torch.Tensor)}, 'state': state,
                 'state_after': {k: v.detach().cpu().clone() for k, v in agent.model.state_dict().items()}})
    return b
agent.play_steps = play
cpu = copy.deepcopy(params); cpu['config']['device'] = 'cpu'
torch.set_num_threads(16)
oracle = None
print('env', {k: os.environ.get(k) for k in ('RLG_CHAIN_PIPE1', 'RLG_PIPE1_WAVES')})
for epoch in range(3):
    agent.update_epoch(); res = agent.train_epoch()
    cap = caps[epoch]
    probe = OracleAgent(cpu, SyntheticTensorEnv(4096, 60, 8, device='cpu', seed=1))
    probe.model.load_full_state_dict(cap['state'])
    b = cap['batch']
    with torch.no_grad():
        probe.model.obs_stats_training = False
        mu, logstd, values = probe.model.a2c_network(probe.model.norm_obs(b['obses']))
    d_mu = (mu - b['mus']).abs()
    rows_bad = (d_mu.max(dim=1).values > 1e-4).nonzero().reshape(-1)
    print(f'epoch {epoch}: max |mu_rollout - mu_oracle| {d_mu.max().item():.3e} (mean {d_mu.mean().item():.3e}, |mu| max {b["mus"].abs().max().item():.2f}); '
          f'rows off by > 1e-4: {rows_bad.numel()} of {mu.shape[0]} {rows_bad[:8].tolist()}')
    same_state = all(torch.equal(cap['state'][k], cap['state_after'][k]) for k in cap['state'])
    print(f'         model state unchanged by the rollout: {same_state}')
    if oracle is None:
        oracle = OracleAgent(cpu, SyntheticTensorEnv(4096, 60, 8, device='cpu', seed=1))
        oracle.model.load_full_state_dict(cap['state_after'])
    # the oracle's update step by step, with a look at how close rows sit to the kinks of the objective before each step:
    # the ratio clip (1 +- e_clip, common_losses.py:64-82), the value clip (|v - v_old| = e_clip, :16-29), the max() ties
    from oracle import ppo_oracle as O
    oracle.prepare_dataset(b)
    ref, nmb = [], oracle.B // oracle.mb
    e_clip = oracle.hp['e_clip']
    for me in range(oracle.mini_epochs):
        for i in range(nmb):
            ds = oracle.dataset
            lo, hi = i * oracle.mb, (i + 1) * oracle.mb
            with torch.no_grad():
                saved_stats = copy.deepcopy(oracle.model.obs_stats)       # (training-mode normalisation, as the step itself sees it)
                oracle.model.obs_stats_training = True
                mu_, logstd_, v_ = oracle.model.a2c_network(oracle.model.norm_obs(ds['obs'][lo:hi]))
                oracle.model.obs_stats = saved_stats
                sig_ = torch.exp(logstd_)
                nlp_ = torch.squeeze(O.neglogp(ds['actions'][lo:hi], mu_, sig_, logstd_))
                ratio = torch.exp(ds['old_logp_actions'][lo:hi] - nlp_)
                d_ratio = torch.minimum((ratio - (1 + e_clip)).abs(), (ratio - (1 - e_clip)).abs()).min().item()
                dv = (v_.reshape(-1) - ds['old_values'][lo:hi].reshape(-1)).abs()
                d_value = (dv - e_clip).abs().min().item()
            ref.append(oracle.minibatch_step(i))
            ref[-1]['near'] = (d_ratio, d_value)
    steps = len(res[4])
    rows = agent._mb_scalars[:steps].cpu()
    for k, r in enumerate(ref):
        print(f'   step {k}: a_loss {rows[k,0].item():+.6e} vs {float(r["a_loss"]):+.6e}  kl {rows[k,4].item():.6e} vs {float(r["kl"]):.6e}  lr(oracle) {r["lr"]:.3e}'
              f'  closest row to the ratio clip {r["near"][0]:.1e}, to the value clip {r["near"][1]:.1e}')
    final, want = agent.model.state_dict(), oracle.model.full_state_dict()
    name = 'a2c_network.actor_mlp.0.weight'
    rel = ((final[name].cpu() - want[name]).abs().mean() / want[name].abs().mean()).item()
    print(f'   after epoch {epoch}: {name} mean|d|/mean|x| {rel:.2e}; lr agent {agent.optimizer.last_and_next_lr()[1]} oracle {oracle.lr}')
