"""bench.py - PPO env-steps/s of the MI355X hot path on BASELINE.json's headline workload.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full PPO epoch of rl_games' ContinuousA2CBase.train_epoch on the
Isaac-Humanoid-shaped synthetic workload (obs 108, act 21, 65,536 envs x horizon 32, MLP
[400,200,100], minibatch 32,768, 5 mini-epochs = 320 optimiser steps): rollout with policy
inference and every buffer write, GAE, dataset preparation, every minibatch forward / fused
loss / backward / clip / Adam / adaptive-lr step.  Inputs are generated on the device (no PCIe
in the timed region).  With N GPUs the 65,536 envs (and the minibatch) are sharded N ways
("strong" scaling) with one gradient all-reduce per optimiser step.

Prints ONE JSON line on rank 0 with the contract keys plus `roofline` (the fused GAE kernel:
17 algorithmic bytes per env-step / its mean launch duration from HIP events recorded on the
launch stream inside the timed region) and, at N=1, `cpu_baseline` (the CPU port of the
reference epoch - oracle/ppo_epoch_oracle.py - timed on this box's host cores on a bounded
sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

GLOBAL_ENVS = 65536
HORIZON = 32
GLOBAL_MINIBATCH = 32768
HBM_PEAK_GBS = 8000.0         # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP32_MFMA_PEAK_TFLOPS = 157.3  # dense fp32 MFMA peak (MI355X_MICROARCH.md)
GAE_BYTES_PER_ENV_STEP = 17   # r 4 + v 4 + done 1 read, returns 4 + advantages 4 written


def cpu_baseline(sample_envs, threads):
    """The oracle epoch (CPU port of the reference path) on `sample_envs` envs x 32."""
    from oracle.ppo_epoch_oracle import OracleAgent
    from rl_games_amd import configs
    from rl_games_amd.synthetic_env import SyntheticTensorEnv
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        params = configs.humanoid_65536(num_actors=sample_envs, minibatch_size=GLOBAL_MINIBATCH, device='cpu')
        env = SyntheticTensorEnv(sample_envs, 108, 21, device='cpu', seed=1234)
        agent = OracleAgent(params, env, seed=0)
        agent.train_epoch()                               # warm-up epoch
        times = [agent.train_epoch()['total_time'] for _ in range(2)]
        per_epoch = sum(times) / len(times)
        return {
            'value': sample_envs * HORIZON / per_epoch, 'unit': 'env-steps/s', 'cores': threads,
            'kind': 'port',
            'sample': f'{sample_envs} envs x {HORIZON} (1/{GLOBAL_ENVS // sample_envs} of the workload), same '
                      f'model/minibatch 32768 x 5 mini-epochs, 1 warm-up + 2 timed epochs, torch CPU threads '
                      f'{threads} (reference default torch_threads=min(4,cores)), host cores {os.cpu_count()}',
            'seconds_per_epoch': per_epoch,
        }
    finally:
        torch.set_num_threads(prev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample-envs', type=int, default=4096)
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f'--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node '
                             f'{args.gpus} bench.py ...`')
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback for the hot path)')
    from rl_games_amd import distributed as rdist
    dev_index = rdist.local_device_index(local_rank)     # = local_rank outside the single-GPU test mode
    torch.cuda.set_device(dev_index)
    device = f'cuda:{dev_index}'
    multi = world > 1
    if multi:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        import torch.distributed as dist
        dist.init_process_group(rdist.backend_for(True), rank=rank, world_size=world)

    from rl_games_amd import configs
    from rl_games_amd.agent import A2CAgent
    if GLOBAL_ENVS % world or GLOBAL_MINIBATCH % world:
        raise SystemExit('world size must divide 65536')
    envs = GLOBAL_ENVS // world
    params = configs.humanoid_65536(num_actors=envs, minibatch_size=GLOBAL_MINIBATCH // world, device=device,
                                    multi_gpu=multi)
    params['config']['env_config']['seed'] = 1234 + rank
    # GEMM solution selection: shipped TunableOp file; shapes missing from it (other library
    # versions) are tuned during the untimed warm-up epochs.
    params['config']['gemm_tuning_online'] = args.warmup >= 1
    torch.manual_seed(42 + rank)
    agent = A2CAgent('bench', params)
    agent.init_tensors()
    agent.obs = agent.env_reset()
    agent.broadcast_parameters()

    def barrier():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        agent.update_epoch()
        agent.train_epoch()
    agent.kernel_timers = {}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        agent.update_epoch()
        agent.train_epoch()
    barrier()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    in_sync = None
    if multi:   # outside the timed region: every rank must hold bit-identical parameters and lr
        probe = torch.stack([agent.optimizer.flat_params.double().sum(),
                             agent.optimizer.flat_params.double().abs().sum(),
                             torch.tensor(agent.optimizer.last_and_next_lr()[1], dtype=torch.float64, device=device)])
        lo, hi = probe.clone(), probe.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        in_sync = bool(torch.equal(lo, hi))

    pairs = agent.kernel_timers.get('gae_envmajor_fused', [])
    gae_us = sum(p.elapsed_us() for p in pairs) / max(len(pairs), 1)
    gae_bytes = envs * HORIZON * GAE_BYTES_PER_ENV_STEP
    achieved = gae_bytes / (gae_us * 1e-6) / 1e9 if gae_us > 0 else 0.0

    # second roofline: the f32-MFMA weight-gradient launch (the largest single kernel of the epoch).
    # Timed with HIP events around back-to-back launches on the last minibatch's own operands,
    # AFTER the timed region (inside it the launch is a node of a replayed HIP graph).
    mfma = None
    eng = getattr(agent, '_engine', None)
    if eng is not None and getattr(eng, 'last_dw_jobs', None):
        jobs, plan = eng.last_dw_jobs
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            plan.launch(jobs)
        reps = 20
        ev0.record()
        for _ in range(reps):
            plan.launch(jobs)
        ev1.record()
        torch.cuda.synchronize()
        us = ev0.elapsed_time(ev1) * 1e3 / reps
        rows = jobs[0][0].shape[0]
        dw_traffic = None
        try:   # HBM bytes per launch measured offline with rocprofv3 --pmc (profiles/r1_dw_pmc.txt)
            with open(os.path.join(ROOT, 'profiles', 'dw_pmc_traffic.json')) as f:
                dw_traffic = json.load(f).get(str(rows), {}).get('traffic_bytes')
        except Exception:
            dw_traffic = None
        flops = sum(2.0 * rows * g.shape[0] * g.shape[1] for _, _, g in jobs)
        mfma = {'kernel': 'rlg::mlp_dw_kernel + rlg::mlp_dw_finalize_kernel (all weight gradients, one launch)',
                'bound': 'mfma', 'achieved': flops / us / 1e6, 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': flops / us / 1e6 / FP32_MFMA_PEAK_TFLOPS, 'traffic': dw_traffic,
                'algorithmic_flops_per_launch': flops, 'avg_launch_us': us, 'launches': reps,
                'note': 'useful flops 2*rows*sum(No*Mi) (tile padding not counted); dense fp32 MFMA peak '
                        '(v_mfma_f32_32x32x2_f32, MI355X_MICROARCH.md); timed after the timed region'}

    traffic = None
    try:   # PMC counters cannot be read inside a normal run: measured offline, see profiles/r1_gae_pmc.txt
        with open(os.path.join(ROOT, 'profiles', 'gae_pmc_traffic.json')) as f:
            traffic = json.load(f).get(f'{envs}x{HORIZON}', {}).get('traffic_bytes')
    except Exception:
        traffic = None

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        out = {
            'metric': 'ppo_env_steps_per_sec', 'value': GLOBAL_ENVS * HORIZON * args.steps / elapsed,
            'unit': 'env-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {
                'workload': 'Isaac-Humanoid-shaped PPO epoch (BASELINE.json configs[2]/[3]): obs 108, act 21, '
                            '65536 envs x horizon 32 global, MLP [400,200,100] elu, fixed sigma',
                'global_envs': GLOBAL_ENVS, 'horizon': HORIZON, 'envs_per_gpu': envs,
                'minibatch_per_gpu': GLOBAL_MINIBATCH // world, 'mini_epochs': 5,
                'optimizer_steps_per_epoch': 5 * (envs * HORIZON) // (GLOBAL_MINIBATCH // world),
                'parallelism': f'dp{world}', 'step': 'one train_epoch (rollout+GAE+dataset+update)',
                'lr_schedule': 'adaptive (device side)', 'mixed_precision': False,
            },
            'roofline': {
                'kernel': 'rlg::gae_envmajor_kernel<32,false> (GAE + returns + advantages + fp64 moments)',
                'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                'traffic_note': 'HBM bytes/launch from rocprofv3 --pmc FETCH_SIZE(x2)/WRITE_SIZE, profiles/r1_gae_pmc.txt',
                'algorithmic_bytes_per_launch': gae_bytes, 'avg_launch_us': gae_us, 'launches': len(pairs),
                'timing': 'HIP start/stop events attached to each in-epoch GAE dispatch on its launch stream (hipExtLaunchKernelGGL), timed region',
            },
        }
        if mfma is not None:
            out['roofline_mfma'] = mfma
        if in_sync is not None:
            out['config']['ranks_in_sync'] = in_sync
        if world == 1 and not args.no_cpu_baseline:
            threads = max(1, min(4, os.cpu_count() or 1))
            out['cpu_baseline'] = cpu_baseline(args.cpu_sample_envs, threads)
            out['gpu_over_cpu'] = out['value'] / out['cpu_baseline']['value']
        print(json.dumps(out))
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
