"""Single-GPU emulation of ONE rank's work at world sizes 1/2/4/8 (no collective): upper bound
on strong scaling = T(world=1) / T_rank(world)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_games_amd import configs
from rl_games_amd.agent import A2CAgent
base = None
for w in (1, 2, 4, 8):
    agent = A2CAgent('r', configs.humanoid_65536(num_actors=65536 // w, minibatch_size=32768 // w))
    agent.init_tensors(); agent.obs = agent.env_reset()
    for _ in range(2):
        agent.update_epoch(); agent.train_epoch()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        agent.update_epoch(); out = agent.train_epoch()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    base = base or dt
    print(f'world {w}: rank epoch {dt*1e3:.1f} ms (play {out[1]*1e3:.1f} update {out[2]*1e3:.1f})  ideal-collective speedup {base/dt:.2f}x')
    del agent
