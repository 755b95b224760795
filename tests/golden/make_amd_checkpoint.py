"""Runs on an MI355X (gpurun): trains the tiny config for two epochs with rl_games_amd.A2CAgent and
writes its checkpoint to gpurun_out/amd_checkpoint.pth (copied to tests/golden/ afterwards).  The
CPU test tests/test_vs_reference_cpu.py::test_reference_restores_amd_checkpoint loads it into the
REAL reference agent."""
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rl_games_amd import configs                     # noqa: E402
from rl_games_amd.agent import A2CAgent              # noqa: E402

N, H, O_, A = 32, 8, 6, 2
params = configs.tiny(num_actors=N, horizon=H, obs_dim=O_, act_dim=A, train_dir='/tmp/rlg_amd_ckpt_runs')
params['config']['env_config'].update(seed=78)
stored = copy.deepcopy(params)
torch.manual_seed(21)
agent = A2CAgent('amd', params)
agent.init_tensors()
agent.obs = agent.env_reset()
for _ in range(2):
    agent.update_epoch()
    agent.frame += agent.batch_size
    agent.train_epoch()
agent.last_mean_rewards = -3.5
out_dir = os.path.join(ROOT, 'gpurun_out')
os.makedirs(out_dir, exist_ok=True)
path = agent.save(os.path.join(out_dir, 'amd_checkpoint'))
ck = torch.load(path, map_location='cpu', weights_only=False)
ck['_meta'] = {'params': stored, 'env': {'num_envs': N, 'obs_dim': O_, 'act_dim': A, 'seed': 78}}
torch.save(ck, path)          # CPU tensors: loadable on a host without a GPU
print('written', path, os.path.getsize(path) // 1024, 'KiB', 'epoch', ck['epoch'], 'lr',
      ck['optimizer']['param_groups'][0]['lr'])
