import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_games_amd import configs
from rl_games_amd.agent import A2CAgent
which = sys.argv[1] if len(sys.argv) > 1 else 'tiny'
params = getattr(configs, which)()
agent = A2CAgent('smoke', params)
agent.init_tensors(); agent.obs = agent.env_reset()
for ep in range(3):
    agent.update_epoch()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = agent.train_epoch()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    a = torch.stack(out[4]).mean().item(); c = torch.stack(out[5]).mean().item(); kl = torch.stack(out[8]).mean().item()
    print(f'epoch {ep}: {dt*1e3:.1f} ms  {agent.batch_size/dt/1e6:.2f} M env-steps/s  a_loss {a:.5f} c_loss {c:.5f} kl {kl:.5f} lr {out[9]:.2e} '
          f'play {out[1]*1e3:.1f} ms update {out[2]*1e3:.1f} ms')
print('count', agent.model.running_mean_std.count.item(), 'value count', agent.model.value_mean_std.count.item())
print('meters', agent.game_rewards.current_size, agent.game_rewards.get_mean(), agent.game_lengths.get_mean())
