"""BASELINE INFRASTRUCTURE (never imported by rl_games_amd): the UNTOUCHED reference agent on the host's cores.

tools/cpu_reference_baseline.py calls this in the build container (the calibration of bench.py's "port" baseline): it builds
rl_games.algos_torch.a2c_continuous.A2CAgent through the reference's own torch_runner.Runner (torch_runner.py:
algo_factory, :217-226 torch_threads), `device: cpu`, on the same synthetic tensor env and the same parameters as the
MI355X run, and times A2CAgent.train_epoch (a2c_common.py:1517-1584) with perf_counter - SURVEY.md 8(d).
The reference comes from /root/reference (tests/golden/ref_import.py); ReferenceUnavailable where that does not exist
(the GPU box: the reference is Python and does not travel - bench.py times the oracle's port there)."""
import copy
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
_GOLDEN = os.path.join(os.path.dirname(HERE), 'tests', 'golden')


def _ref_import():
    if _GOLDEN not in sys.path:
        sys.path.insert(0, _GOLDEN)
    import ref_import
    return ref_import


def available():
    ri = _ref_import()
    return os.path.isdir(os.path.join(ri.REFERENCE, 'rl_games'))


def reference_agent(params, env):
    """The reference's own agent for `params` (the dict rl_games_amd.configs builds) on the injected vec env."""
    ri = _ref_import()
    ri.enable()
    from rl_games.torch_runner import Runner
    runner = Runner()
    p = copy.deepcopy(params)
    p['config']['env_info'] = env.get_env_info()
    runner.load({'params': p})
    runner.params['config']['vec_env'] = env
    runner.params['config']['env_info'] = env.get_env_info()
    agent = runner.algo_factory.create(runner.algo_name, base_name='cpu_baseline', params=runner.params)
    agent.init_tensors()
    agent.obs = agent.env_reset()
    return agent, ri.source()


def time_epochs(agent, epochs, is_reference, budget_s=None):
    """1 warm-up epoch + up to `epochs` timed ones (stops early once `budget_s` seconds of timed epochs are spent);
    returns (warm-up seconds, [timed seconds])."""
    warm, times = None, []
    for e in range(epochs + 1):
        if is_reference:
            agent.epoch_num += 1
        t0 = time.perf_counter()
        agent.train_epoch()
        dt = time.perf_counter() - t0
        if e == 0:
            warm = dt
        else:
            times.append(dt)
            if budget_s is not None and sum(times) >= budget_s:
                break
    return warm, times
