"""Times the UNTOUCHED reference agent (rl_games.algos_torch.a2c_continuous.A2CAgent on CPU) next to
the oracle port (oracle/ppo_epoch_oracle.OracleAgent) on the same host, same config, same
synthetic env - the calibration `bench.py` ships as `cpu_baseline.calibration`.

Build container only (needs /root/reference); the GPU box has no reference, so bench.py times
the port there and reports this file's `ref_over_port` ratio next to it.

    python tools/cpu_reference_baseline.py [--envs 4096] [--epochs 2] [--out profiles/cpu_baseline_calibration.json]

Rows (BASELINE.md section 3 / SURVEY 8d): reference default threading (torch_threads =
min(4, cores), torch_runner.py:217-226) AND all cores; whole train_epoch and the leaf
functions (_pytorch_gae, RunningMeanStd.forward, calc_losses chain, policy_kl).
"""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

import torch  # noqa: E402


from oracle.reference_baseline import reference_agent as _reference_agent, time_epochs as _time_epochs  # noqa: E402


def reference_agent(params, env):
    return _reference_agent(params, env)[0]


def time_epochs(agent, epochs, is_reference):
    return _time_epochs(agent, epochs, is_reference)[1]


def leaf_timings(threads, H=32, N=65536, mb=32768, O_=108, A=21):
    """Leaf functions of the reference alone (CPU), at the headline sizes."""
    import ref_import
    ref_import.enable()
    from rl_games.algos_torch import torch_ext
    from rl_games.algos_torch.running_mean_std import RunningMeanStd
    from rl_games.common import common_losses
    from rl_games.triton_kernels.gae_kernel import _pytorch_gae
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(0)
    out = {}

    def best(fn, reps=3):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return min(ts)

    r, v = torch.randn(H, N, 1, generator=g), torch.randn(H, N, 1, generator=g)
    d = (torch.rand(H, N, generator=g) < 0.05).float()
    lv, ld = torch.randn(N, 1, generator=g), (torch.rand(N, generator=g) < 0.05).float()
    out['_pytorch_gae_32x65536_s'] = best(lambda: _pytorch_gae(r, v, d, lv, ld, 0.99, 0.95))
    rms = RunningMeanStd((O_,))
    rms.train()
    x = torch.randn(mb, O_, generator=g)
    out['RunningMeanStd_forward_train_32768x108_s'] = best(lambda: rms(x))
    old_nlp = torch.randn(mb, generator=g)
    nlp = old_nlp + 0.2 * torch.randn(mb, generator=g)
    adv = torch.randn(mb, generator=g)
    out['actor_loss_32768_s'] = best(lambda: common_losses.actor_loss(old_nlp, nlp, adv, True, 0.2))
    v_old = torch.randn(mb, 1, generator=g)
    vv = v_old + 0.3 * torch.randn(mb, 1, generator=g)
    R = torch.randn(mb, 1, generator=g)
    out['critic_loss_32768_s'] = best(lambda: common_losses.critic_loss(None, v_old, vv, 0.2, R, True))
    mu0, mu1 = torch.randn(mb, A, generator=g), torch.randn(mb, A, generator=g)
    s0, s1 = torch.rand(mb, A, generator=g) + 0.5, torch.rand(mb, A, generator=g) + 0.5
    out['policy_kl_32768x21_s'] = best(lambda: torch_ext.policy_kl(mu0, s0, mu1, s1, True))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=4096)
    ap.add_argument('--epochs', type=int, default=2)
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'cpu_baseline_calibration.json'))
    args = ap.parse_args()

    from oracle.ppo_epoch_oracle import OracleAgent
    from rl_games_amd import configs
    from rl_games_amd.synthetic_env import SyntheticTensorEnv
    cores = os.cpu_count()
    rows = {}
    for label, threads in (('default_threads', max(1, min(4, cores))), ('all_cores', cores)):
        torch.set_num_threads(threads)
        params = configs.humanoid_65536(num_actors=args.envs, minibatch_size=32768, device='cpu',
                                        train_dir='/tmp/rlg_cpu_baseline_runs')
        params['seed'] = 7
        env = SyntheticTensorEnv(args.envs, 108, 21, device='cpu', seed=1234)
        ref = reference_agent(params, env)
        t_ref = time_epochs(ref, args.epochs, True)
        del ref
        env = SyntheticTensorEnv(args.envs, 108, 21, device='cpu', seed=1234)
        port = OracleAgent(copy.deepcopy(params), env, seed=0)
        t_port = time_epochs(port, args.epochs, False)
        del port
        steps = args.envs * 32
        ref_v, port_v = steps / (sum(t_ref) / len(t_ref)), steps / (sum(t_port) / len(t_port))
        rows[label] = {'threads': threads, 'reference_env_steps_per_s': ref_v, 'port_env_steps_per_s': port_v,
                       'ref_over_port': ref_v / port_v, 'reference_epoch_s': t_ref, 'port_epoch_s': t_port,
                       'leaf_s': leaf_timings(threads)}
        print(label, json.dumps(rows[label]), flush=True)
    out = {
        'what': 'untouched reference a2c_continuous.A2CAgent.train_epoch on CPU vs oracle/ppo_epoch_oracle.OracleAgent '
                '(the port bench.py times on the GPU box), same config / env / host',
        'config': f'humanoid-shaped obs 108 act 21, {args.envs} envs x 32, MLP [400,200,100], minibatch 32768, '
                  f'5 mini-epochs, fp32, 1 warm-up + {args.epochs} timed epochs',
        'host_cores': cores, 'torch': torch.__version__, 'rows': rows,
        'source': 'tools/cpu_reference_baseline.py (build container; /root/reference imported with the test stubs)',
    }
    with open(args.out, 'w') as f:
        json.dump(out, f, indent=1)
    print('wrote', args.out)


if __name__ == '__main__':
    main()
