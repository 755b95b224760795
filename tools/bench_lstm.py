"""Epoch time of BASELINE config #5 (LSTM policy, 4,096 envs x seq_len 16) on one MI355X."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_games_amd import configs
from rl_games_amd.agent import A2CAgent
over = {}
for a in sys.argv[1:]:
    k, v = a.split('='); over[k] = int(v)
agent = A2CAgent('lstm', configs.pendulum_lstm_4096(**over))
agent.init_tensors(); agent.obs = agent.env_reset()
for _ in range(2):
    agent.update_epoch(); agent.train_epoch()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 5
for _ in range(n):
    agent.update_epoch(); out = agent.train_epoch()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f'cfg#5 lstm {over}: epoch {dt*1e3:.1f} ms -> {agent.batch_size/dt/1e3:.0f} k env-steps/s '
      f'(play {out[1]*1e3:.1f} ms, update {out[2]*1e3:.1f} ms host-side)')
