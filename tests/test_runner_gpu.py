"""The REAL reference `Runner` drives this repository's agents ON THE DEVICE (SURVEY.md 8(b) seam 1;
rl_games/torch_runner.py:117-120 register_builder, :233-315 run_train -> agent.train(), :342 run).

The reference is imported through tests/golden/ref_import.py and is Python: it does not travel to the GPU box, so
these tests run only on a machine that has BOTH an MI355X and the reference checkout (they skip on the driver's GPU
box; rounds 4 - 5 ran them there from a staged copy of the reference, which round 6 removed - the record of those
runs is profiles/r5_INDEX.md).  The reference is the CALLER in these tests, never the thing under test: every kernel that
runs belongs to rl_games_amd (the A2CAgent asserts its engine / MFMA weight-gradient path)."""
import copy
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _reference():
    import ref_import
    try:
        ref_import.enable()
    except ref_import.ReferenceUnavailable as e:            # (the GPU box: no reference checkout)
        pytest.skip(str(e))
    from rl_games.torch_runner import Runner
    return Runner


class _HostVecEnv:
    """An IVecEnv (rl_games/common/ivecenv.py:1-36) that speaks numpy like the reference's CPU vec-envs do
    (BASELINE.json configs[0]: `GymnasiumVecEnv`: float64 observations, bool dones, numpy actions expected,
    next_step autoreset => the masked-rows path)."""

    def __init__(self, num_envs, obs_dim, discrete_actions):
        from rl_games_amd.synthetic_env import SyntheticTensorEnv
        self.inner = SyntheticTensorEnv(num_envs, obs_dim, 0, device='cpu', seed=77, discrete_actions=discrete_actions,
                                        autoreset_mode='next_step', p_done=0.1)
        self.steps = 0

    def reset(self):
        return self.inner.reset().numpy().astype(np.float64)

    def step(self, actions):
        assert isinstance(actions, np.ndarray) and actions.shape[0] == self.inner.num_envs
        self.steps += 1
        obs, rewards, dones, infos = self.inner.step(None)
        return (obs.numpy().astype(np.float64), rewards.numpy().astype(np.float64), dones.bool().numpy(),
                {'time_outs': infos['time_outs'].numpy()})

    def get_env_info(self):
        return self.inner.get_env_info()

    def __getattr__(self, name):          # has_action_masks, set_train_info, get_env_state ...
        return getattr(self.inner, name)


def _runner_with_our_agents(params, env, results):
    """Runner().load(params) with rl_games_amd's agents registered under the reference's algo names; the agents are
    thin recording subclasses so that the test sees what `run_train` discards: agent.train()'s return value."""
    Runner = _reference()
    from rl_games_amd.agent import A2CAgent
    from rl_games_amd.discrete_agent import DiscreteA2CAgent

    def recording(cls):
        class Recording(cls):
            def train(self):
                out = super().train()
                results.append((self, out))
                return out
        return Recording

    runner = Runner()
    runner.algo_factory.register_builder('a2c_continuous', lambda **kw: recording(A2CAgent)(**kw))
    runner.algo_factory.register_builder('a2c_discrete', lambda **kw: recording(DiscreteA2CAgent)(**kw))
    runner.load({'params': copy.deepcopy(params)})
    runner.params['config']['vec_env'] = env
    runner.params['config']['env_info'] = env.get_env_info()
    return runner


@pytest.mark.parametrize('torch_compile', [False, 'runner_default'])
def test_reference_runner_trains_our_continuous_agent_on_the_device(tmp_path, torch_compile):
    """Runner.run({'train': True}) -> run_train -> algo_factory.create(name, base_name='run', params) -> agent.train()
    (torch_runner.py:233-315, :342) on SyntheticTensorEnv (device tensors, zero-copy mode): 3 epochs, the
    (last_mean_rewards, epoch_num) contract (a2c_common.py:1782), checkpoints in the reference's on-disk format - and the
    reference's own PpoPlayerContinuous restores what our agent wrote (players.py:17-82, torch_ext.py:73-112).
    'runner_default': the params carry no `torch_compile` key, so the Runner re-assigns agent.model to
    torch.compile(agent.model) (:307) - the wrapper must stay harmless: our launches never call model.forward."""
    from rl_games_amd import configs
    from rl_games_amd.synthetic_env import SyntheticTensorEnv
    params = configs.tiny(num_actors=128, horizon=8, obs_dim=12, act_dim=3, max_epochs=3, save_frequency=1,
                          save_best_after=0, train_dir=str(tmp_path), full_experiment_name='runner_run', device=DEV)
    if torch_compile is False:
        params['config']['torch_compile'] = False
    params['config']['env_config']['p_done'] = 0.2
    env = SyntheticTensorEnv(128, 12, 3, device=DEV, seed=5, p_done=0.2)
    results = []
    runner = _runner_with_our_agents(params, env, results)
    runner.run({'train': True})
    assert len(results) == 1
    agent, (last_mean_rewards, epoch_num) = results[0]
    assert type(agent).__mro__[1].__module__ == 'rl_games_amd.agent'
    assert epoch_num == 3 and agent.frame == 3 * 128 * 8 and np.isfinite(last_mean_rewards)
    assert agent._engine is not None and agent._engine.last_dw_path == 'mfma'       # our kernels did the work
    assert agent.is_tensor_obses
    for p in agent.model.parameters():
        assert torch.isfinite(p).all()
    nn_dir = os.path.join(str(tmp_path), 'runner_run', 'nn')
    files = sorted(os.listdir(nn_dir))
    last = [f for f in files if f.startswith('last_tiny_ep_3')]
    assert last, files
    ck = torch.load(os.path.join(nn_dir, last[0]), map_location='cpu', weights_only=False)
    assert ck['epoch'] == 3 and ck['frame'] == 3 * 128 * 8
    assert not any(k.startswith('_orig_mod.') for k in ck['model'])              # torch_ext.py:73-112's key format
    # the reference's player restores it and its deterministic action is our policy mean
    from rl_games.algos_torch.players import PpoPlayerContinuous as RefPlayer
    pp = copy.deepcopy(params)
    pp['config'].update(device='cpu', device_name='cpu', env_info=env.get_env_info(), torch_compile=False,
                        player={'games_num': 1, 'print_stats': False})
    pp['config']['vec_env'] = None
    player = RefPlayer(pp)
    player.restore(os.path.join(nn_dir, last[0]))
    sd = player.model.state_dict()
    mine = {k.replace('_orig_mod.', ''): v for k, v in agent.model.state_dict().items()}
    assert sorted(sd.keys()) == sorted(mine.keys())
    for k, v in sd.items():
        assert torch.equal(v.cpu(), mine[k].detach().cpu()), k
    obs = 3.0 * torch.randn(16, 12) + 1.0
    player.has_batch_dimension = True
    a_ref = player.get_action(obs.clone(), is_deterministic=True)
    agent.set_eval()
    mu = agent.get_action_values({'obs': obs.to(DEV)})['mus']
    assert torch.allclose(a_ref, torch.clamp(mu, -1.0, 1.0).cpu(), rtol=1e-5, atol=1e-6)


def test_reference_runner_trains_our_discrete_agent_on_a_numpy_vec_env(tmp_path):
    """BASELINE.json configs[0] through the same seam: a2c_discrete on a CPU vec-env that speaks numpy with next_step
    autoreset (gymnasium_vecenv.py:245 => mask_autoreset_rows, a2c_common.py:347-348 => the masked path), 3 epochs."""
    from rl_games_amd import configs
    params = configs.cartpole_discrete(num_actors=16, device=DEV, max_epochs=3, train_dir=str(tmp_path), save_best_after=0,
                                       full_experiment_name='runner_cartpole', torch_compile=False)
    _reference()                                             # (puts the gymnasium stub on sys.path)
    env = _HostVecEnv(16, 4, 2)
    results = []
    runner = _runner_with_our_agents(params, env, results)
    runner.run({'train': True})
    agent, (last_mean_rewards, epoch_num) = results[0]
    assert type(agent).__mro__[1].__module__ == 'rl_games_amd.discrete_agent'
    assert epoch_num == 3 and agent.frame == 3 * 16 * params['config']['horizon_length']
    assert env.steps == 3 * params['config']['horizon_length']
    assert agent.mask_autoreset_rows and not agent.is_tensor_obses
    assert np.isfinite(last_mean_rewards)
    for p in agent.model.parameters():
        assert torch.isfinite(p).all()
