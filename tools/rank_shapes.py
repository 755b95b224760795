"""Single-GPU emulation of ONE rank's work at world sizes 1/2/4/8 (no collective): upper bound
on strong scaling = T(world=1) / T_rank(world).  Optional A/B over agent config flags:
    python tools/rank_shapes.py mini_epoch_graph=0,1 worlds=1,8"""
import sys, os, time, itertools, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_games_amd import configs
from rl_games_amd.agent import A2CAgent

flags, worlds = {}, (1, 2, 4, 8)
for arg in sys.argv[1:]:
    k, v = arg.split('=')
    if k == 'worlds':
        worlds = tuple(int(x) for x in v.split(','))
    else:
        flags[k] = [int(x) for x in v.split(',')]
keys = list(flags)
for combo in itertools.product(*[flags[k] for k in keys]) if keys else [()]:
    over = dict(zip(keys, combo))
    base = None
    for w in worlds:
        agent = A2CAgent('r', configs.humanoid_65536(num_actors=65536 // w, minibatch_size=32768 // w, **over))
        agent.init_tensors(); agent.obs = agent.env_reset()
        for _ in range(2):
            agent.update_epoch(); agent.train_epoch()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            agent.update_epoch(); out = agent.train_epoch()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        base = base or dt
        print(f'{over} world {w}: rank epoch {dt*1e3:.1f} ms  ideal-collective speedup {base/dt:.2f}x', flush=True)
        del agent
