"""Soak: many epochs of the bench workload; checks that losses, parameters and statistics stay finite."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_games_amd import configs
from rl_games_amd.agent import A2CAgent
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 60
agent = A2CAgent('soak', configs.humanoid_65536())
agent.init_tensors(); agent.obs = agent.env_reset()
t0 = time.time()
for e in range(epochs):
    agent.update_epoch()
    out = agent.train_epoch()
    if e % 10 == 9 or e == epochs - 1:
        a = torch.stack(out[4]).mean().item(); c = torch.stack(out[5]).mean().item(); kl = torch.stack(out[8]).mean().item()
        ok = torch.isfinite(agent.optimizer.flat_params).all().item()
        print(f'epoch {e+1}: a_loss {a:.4e} c_loss {c:.4e} kl {kl:.4e} lr {out[9]:.3e} params finite {ok} '
              f'obs count {agent.model.running_mean_std.count.item()}', flush=True)
        assert ok and all(map(lambda x: x == x, (a, c, kl)))
        w = max(p.abs().max().item() for p in agent.model.a2c_network.parameters())
        print(f'          largest |parameter| {w:.3f} (split-fp16 weight scale: < 1023)', flush=True)
torch.cuda.synchronize()
print(f'{epochs} epochs in {time.time()-t0:.1f} s; peak memory {torch.cuda.max_memory_allocated()/2**30:.2f} GiB')
