import sys, copy, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_headline_gpu as T
from rl_games_amd import configs
from rl_games_amd.agent import A2CAgent
N, H = 4096, 16
params = configs.ant_4096(hip_graphs=False)
torch.manual_seed(9)
agent = A2CAgent('ant', copy.deepcopy(params))
agent.init_tensors(); agent.obs = agent.env_reset()
caps = T._capture_rollout(agent)
agent.update_epoch(); res = agent.train_epoch()
oracle = T._oracle_for(params, caps[0], N, 60, 8)
ref = oracle.update(caps[0]['batch'])
truth = T._truth_for(params, caps[0], N, 60, 8)
tru = truth.update(T._batch64(caps[0]['batch']))
final, want, tr = agent.model.state_dict(), oracle.model.full_state_dict(), truth.model.full_state_dict()
for name in ('a2c_network.sigma', 'a2c_network.mu.bias', 'a2c_network.value.bias'):
    g, w, t = final[name].cpu().double().flatten(), want[name].double().flatten(), tr[name].double().flatten()
    print(name)
    print('  agent - oracle', (g - w).abs().tolist()[:8])
    print('  agent - truth ', (g - t).abs().tolist()[:8])
    print('  oracle - truth', (w - t).abs().tolist()[:8])
    print('  value', w.tolist()[:8])
