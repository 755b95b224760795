"""Run on an MI355X: tunes the MLP GEMM shapes of the BASELINE configs at world sizes 1/2/4/8
(per-rank shapes) and writes one merged TunableOp CSV to gpurun_out/tunableop_gfx950.csv."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rl_games_amd import configs, gemm_tuning
from rl_games_amd.agent import A2CAgent
out = os.path.join(ROOT, 'gpurun_out', 'tunableop_gfx950.csv')
os.makedirs(os.path.dirname(out), exist_ok=True)
gemm_tuning.enable(tuned_file=None, allow_tuning=True, max_tuning_ms=50)
jobs = [('humanoid_65536', dict(num_actors=65536 // w, minibatch_size=32768 // w)) for w in (1, 2, 4, 8)]
jobs += [('ant_4096', {}), ('pendulum_lstm_4096', {})]
for name, kw in jobs:
    t0 = time.time()
    try:
        agent = A2CAgent('tune', getattr(configs, name)(**kw))
        agent.init_tensors(); agent.obs = agent.env_reset()
        for _ in range(2):
            agent.update_epoch(); agent.train_epoch()
        torch.cuda.synchronize()
        print(name, kw, f'{time.time()-t0:.1f}s')
    except Exception as e:
        print('FAILED', name, kw, repr(e)[:300])
    del agent
gemm_tuning.write_results(out)
print(open(out).read()[:300]); print(sum(1 for _ in open(out)), 'lines')
