"""bench.py - PPO env-steps/s of the MI355X hot path on BASELINE.json's headline workload.

    python bench.py --gpus 1 --steps K --warmup W [--workload humanoid|ant|lstm]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    RLG_BENCH_CONFIG='{"native_allreduce": false}' ... bench.py --gpus N ...   (second row: RCCL gradient all-reduce
        instead of the in-graph hipIpc kernel; '{"native_allreduce_two_phase": true}': its reduce-scatter + all-gather
        variant).  tools/scale_check.sh runs the N = 1, 2, 4, 8 rows of both.

One "step" = one full PPO epoch of rl_games' ContinuousA2CBase.train_epoch on a synthetic,
device-resident workload: rollout with policy inference and every buffer write, GAE, dataset
preparation, every minibatch forward / fused loss / backward / clip / Adam / adaptive-lr step.
  humanoid (default; BASELINE.json configs[2]/[3], the config `metric` is quoted on):
      obs 108, act 21, 65,536 envs x horizon 32, MLP [400,200,100], minibatch 32,768, 5 mini-epochs
      = 320 optimiser steps per epoch
  ant (BASELINE.json configs[1]): obs 60, act 8, 4,096 envs x horizon 16, MLP [256,128,64],
      minibatch 32,768, 4 mini-epochs
  lstm (BASELINE.json configs[4]): LSTM policy, obs 3, act 1, 4,096 envs x seq_len 16, MLP [64,64] + LSTM 64,
      minibatch 16,384, 4 mini-epochs
Inputs are generated on the device (no PCIe in the timed region).  With N GPUs the envs (and the
minibatch) are sharded N ways ("strong" scaling) with one gradient all-reduce per
optimiser step.

Prints ONE JSON line on rank 0 with the contract keys plus
  * `ms_per_step_stats`: min / median / mean / max of the K timed epochs (HIP events between the
    epochs; `ms_per_step` itself is the barrier-to-barrier wall time / K),
  * `roofline`: the fused GAE kernel - 17 algorithmic bytes per env-step / its mean launch duration
    from HIP events bound to the dispatch itself (hipExtLaunchKernelGGL start/stop events = the
    begin/end timestamps rocprofv3 --kernel-trace reports), measured inside the timed region;
    `traffic` = HBM bytes per launch measured by rocprofv3 --pmc on THIS command
    (tools/gpu_pmc_bench_gae.sh writes profiles/gae_pmc_traffic.json; null if that file does not
    cover the workload),
  * `roofline_mfma`: the weight-gradient launch against the dense MFMA peak of the instruction it issues
    (bf16 for the default split-product form, with the fp32-equivalent rate beside it),
  * `roofline_fwd` / `roofline_bwd`: the two dominant kernels of the epoch (fused forward / fused backward chain,
    csrc/mlp_chain_bx*.hip) against the dense MFMA peak of the instruction they issue (bf16 for the split-product
    kernels: 6 x the padded-tile flops; the fp32-equivalent rate beside it): mean launch duration over ALL their
    launches of one eager epoch run after the timed region (HIP events bound to each dispatch, like the GAE
    launch; inside the timed region they are nodes of a replayed HIP graph and cannot carry events),
  * at N>1 `config.allreduce` ("ipc" | "ipc-two-phase" | "rccl": the collective that ran), `config.ipc_self_test`,
    `config.ranks_in_sync`,
  * at N=1 `cpu_baseline` (kind "port"): the oracle's restatement of rl_games' A2CAgent.train_epoch
    (oracle/ppo_epoch_oracle.py) timed on this box's host cores in the same run on a bounded sample (1/16 of the envs,
    4 minibatches per mini-epoch), at the reference's default 4 torch threads and at min(cores, 16).  (The reference
    is Python: it is imported in the build container to pin the oracle and to calibrate this port, and does not travel.)
"""
import argparse
import datetime
import json
import os
import signal
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0         # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP32_MFMA_PEAK_TFLOPS = 157.3  # dense fp32 MFMA peak (MI355X_MICROARCH.md)
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md; no sparsity)
GAE_BYTES_PER_ENV_STEP = 17   # r 4 + v 4 + done 1 read, returns 4 + advantages 4 written

WORKLOADS = {
    'humanoid': dict(envs=65536, horizon=32, minibatch=32768, mini_epochs=5, obs=108, act=21, units=[400, 200, 100],
                     desc='Isaac-Humanoid-shaped PPO epoch (BASELINE.json configs[2]/[3]): obs 108, act 21, 65536 envs '
                          'x horizon 32 global, MLP [400,200,100] elu, fixed sigma'),
    'ant': dict(envs=4096, horizon=16, minibatch=32768, mini_epochs=4, obs=60, act=8, units=[256, 128, 64],
                desc='Ant-v5-shaped PPO epoch (BASELINE.json configs[1]): obs 60, act 8, 4096 envs x horizon 16 global, '
                     'MLP [256,128,64] elu, fixed sigma'),
    'lstm': dict(envs=4096, horizon=16, minibatch=16384, mini_epochs=4, obs=3, act=1, units=[64, 64],
                 desc='LSTM-policy PPO epoch (BASELINE.json configs[4], play_steps_rnn path): Pendulum-shaped obs 3, act 1, '
                      '4096 envs x seq_len 16, MLP [64,64] elu + LSTM 64, fixed sigma'),
}


def make_params(workload, num_actors, minibatch, device, multi_gpu=False):
    from rl_games_amd import configs
    if workload == 'humanoid':
        return configs.humanoid_65536(num_actors=num_actors, minibatch_size=minibatch, device=device,
                                      multi_gpu=multi_gpu)
    if workload == 'lstm':
        return configs.pendulum_lstm_4096(num_actors=num_actors, minibatch_size=minibatch, device=device,
                                          multi_gpu=multi_gpu)
    return configs.ant_4096(num_actors=num_actors, minibatch_size=minibatch, device=device, multi_gpu=multi_gpu)


def cpu_baseline(workload, sample_envs):
    """the reference's CPU path timed on THIS box's host cores in this run, on `sample_envs` envs of the
    same workload - kind "port": oracle/ppo_epoch_oracle.OracleAgent, the oracle's restatement of
    a2c_continuous.A2CAgent.train_epoch (same torch CPU operators in the same order).  The reference itself is Python and
    does not travel to the GPU box in any form; what the port is worth against it was measured where both existed: the
    untouched reference ran at 0.84 x the port's rate at the default thread count in the build container
    (profiles/cpu_baseline_calibration.json, tools/cpu_reference_baseline.py) and at 0.90 x on a GPU-box host (round 5's
    driver run, BENCH_r05.json `port_cross_check`) - the port is the FASTER, i.e. the conservative, baseline.  Two rows: the reference's default threading
    (torch_threads = min(4, cores), torch_runner.py:217-226) and many cores (min(cores, 16): with all 256 threads of a
    GPU-box host the same epoch is ~400x SLOWER than with 4 - measured 187 vs 86,511 env-steps/s - so "all cores" is
    neither a sensible baseline nor bounded).  Bounded: every row is 1 warm-up epoch + at most 2 timed ones (one, if the
    warm-up took more than 15 s)."""
    from oracle.ppo_epoch_oracle import OracleAgent
    from rl_games_amd.synthetic_env import SyntheticTensorEnv
    w = WORKLOADS[workload]
    cores = os.cpu_count() or 1
    prev = torch.get_num_threads()
    steps = sample_envs * w['horizon']
    plan = (('default_threads', max(1, min(4, cores))), ('all_cores', min(cores, 16)))

    def run(threads):
        torch.set_num_threads(threads)
        params = make_params(workload, sample_envs, min(w['minibatch'], sample_envs * w['horizon']), 'cpu')
        params['config']['train_dir'] = '/tmp/rlg_cpu_baseline_runs'
        env = SyntheticTensorEnv(sample_envs, w['obs'], w['act'], device='cpu', seed=1234)
        agent = OracleAgent(params, env, seed=0)
        t0 = time.perf_counter()
        times = []
        for e in range(3):                          # 1 warm-up + up to 2 timed epochs
            t1 = time.perf_counter()
            agent.train_epoch()
            times.append(time.perf_counter() - t1)
            if e >= 1 and sum(times[1:]) > 15.0:
                break
        timed = times[1:] or times
        per_epoch = sum(timed) / len(timed)
        return {'threads': threads, 'value': steps / per_epoch, 'seconds_per_epoch': per_epoch,
                'timed_epochs': len(timed), 'row_seconds': time.perf_counter() - t0}

    rows = {}
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):               # stdout carries exactly ONE JSON line
        try:
            for label, threads in plan:
                if label == 'all_cores' and threads == rows['default_threads']['threads']:
                    rows[label] = dict(rows['default_threads'])
                else:
                    rows[label] = run(threads)
        finally:
            torch.set_num_threads(prev)
    d = max(rows.values(), key=lambda r: r['value'])           # the faster row is the baseline (the conservative choice)
    return {
        'value': d['value'], 'unit': 'env-steps/s', 'cores': d['threads'], 'kind': 'port',
        'what': "oracle/ppo_epoch_oracle.OracleAgent - the oracle's CPU restatement of rl_games "
                'a2c_continuous.A2CAgent.train_epoch (a2c_common.py:1517-1584), device cpu',
        'sample': f'{sample_envs} envs x {w["horizon"]} (1/{max(1, w["envs"] // sample_envs)} of the workload), same '
                  f'model / minibatch / mini-epochs, 1 warm-up + <= 2 timed epochs per row, host cores {cores}',
        'seconds_per_epoch': d['seconds_per_epoch'], 'rows': rows, 'host_cores': cores,
        'torch_threads_default_rule': 'min(4, cores) (torch_runner.py:217-226)',
        'rows_note': "`value` is the faster of the two rows; 'default_threads' is what the reference's Runner would use, "
                     "'all_cores' uses min(host cores, 16) torch threads (more threads run this workload slower)",
        'calibration': 'the untouched reference agent ran at 0.84 x this port\'s rate at the default thread count in the '
                       'build container (profiles/cpu_baseline_calibration.json) and at 0.90 x on a GPU-box host (round 5, '
                       'BENCH_r05.json port_cross_check); the reference, being Python, is not shipped to the GPU box',
    }


_KEEP_ALIVE = []


def collective_preflight(agent, device, world):
    """Multi-GPU runs: the latency of ONE gradient all-reduce of this model's flat arena by each transport the agent
    can use - RCCL through torch.distributed, the in-graph hipIpc kernel (one-shot) and its reduce-scatter + all-gather
    variant - each with its own known-answer check, so that a single run of `bench.py --gpus N` says which collective
    the hardware prefers and whether the hand-written one works across real xGMI links at all (a2c_common.py:493-509 is
    what all three replace).  Collective: every rank runs it.  It runs BEHIND the timed region and the in-sync check and
    never destroys its communicators: in front of the run - creating and destroying two hipIpc communicators before the
    agent creates its own - one 2-rank run in eight on one GPU ended with parameters that differed between the ranks in
    the 8th digit (0 of 40 without it; round 4, tests/test_agent_gpu.py::test_two_rank_bench_on_one_gpu), i.e. a
    communicator's staging memory must not be recycled for another one within a process; the agent holds ONE for its
    lifetime."""
    import torch.distributed as dist
    from rl_games_amd.ipc_allreduce import IpcAllReduce
    n = agent.optimizer.flat_grads.numel()
    out = {'floats': n, 'world': world}

    def timed(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6

    t = torch.ones(n, dtype=torch.float32, device=device)
    try:
        chk = torch.full((n,), float(dist.get_rank() + 1), device=device)
        dist.all_reduce(chk)
        ok = bool((chk == world * (world + 1) / 2).all().item())
        out['rccl'] = {'us_per_allreduce': timed(lambda: dist.all_reduce(t)), 'known_answer': ok,
                       'backend': dist.get_backend()}
    except Exception as e:
        out['rccl'] = {'error': f'{type(e).__name__}: {e}'}
    for name, two_phase in (('ipc', False), ('ipc_two_phase', True)):
        comm = None
        try:
            comm = IpcAllReduce(n, device, timeout_s=30.0, two_phase=two_phase)     # (runs its own self-test)
            t.fill_(1.0)
            us = timed(lambda: comm.all_reduce_sum(t))
            _, gave_up = comm.status()
            out[name] = {'us_per_allreduce': us, 'self_test': 'passed', 'timed_out_launch': gave_up,
                         'note': 'eager launches incl. the host launch cost; inside the mini-epoch graph only the kernel remains'}
        except Exception as e:
            out[name] = {'error': f'{type(e).__name__}: {e}'}
        finally:
            torch.cuda.synchronize()
            dist.barrier()
            if comm is not None:
                _KEEP_ALIVE.append(comm)          # (not closed: see the docstring)
        oks = [None] * world
        dist.all_gather_object(oks, 'error' not in out[name])
        if not all(oks):
            out[name].setdefault('error', 'failed on another rank')
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', choices=sorted(WORKLOADS), default='humanoid')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-exact-row', action='store_true', help='skip the exact-fp32-product comparison run (a child '
                    'process with RLG_CHAIN_BX=0 RLG_DW_BF16=0 after the timed region; humanoid, 1 GPU only)')
    ap.add_argument('--cpu-sample-envs', type=int, default=0, help='0: 4096 (humanoid) / 1024 (ant, lstm)')
    ap.add_argument('--max-seconds', type=int, default=1500, help='watchdog: the process exits non-zero with a '
                    'message instead of hanging (a peer rank that died, a collective that never completes)')
    args = ap.parse_args()
    w = WORKLOADS[args.workload]

    def on_alarm(signum, frame):
        sys.stderr.write(f'bench.py: watchdog - no result after {args.max_seconds} s (rank '
                         f'{os.environ.get("RANK", "0")}); a rank died or a collective never completed. Aborting.\n')
        sys.stderr.flush()
        os._exit(3)
    signal.signal(signal.SIGALRM, on_alarm)
    signal.alarm(args.max_seconds)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # dmabuf IPC is the only mode the driver supports (hipIpcGetMemHandle fails otherwise: RCCL and the
    # in-graph all-reduce both need it); must be in the environment before the HIP runtime starts
    if os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0') != '0':
        sys.stderr.write('bench.py: overriding HSA_ENABLE_IPC_MODE_LEGACY=%s with 0 (dmabuf IPC)\n'
                         % os.environ['HSA_ENABLE_IPC_MODE_LEGACY'])
    os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
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
        # every collective of the set-up and of RCCL itself is bounded: a rank that died makes the others fail
        # with a message instead of waiting for ever
        dist.init_process_group(rdist.backend_for(True), rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=min(600, args.max_seconds)))

    from rl_games_amd.agent import A2CAgent
    global_envs, horizon = w['envs'], w['horizon']
    global_mb = min(w['minibatch'], global_envs * horizon)
    if global_envs % world or global_mb % world:
        raise SystemExit(f'world size must divide {global_envs} envs and the {global_mb}-row minibatch')
    envs = global_envs // world
    params = make_params(args.workload, envs, global_mb // world, device, multi_gpu=multi)
    params['config']['env_config']['seed'] = 1234 + rank
    # GEMM solution selection: shipped TunableOp file; shapes missing from it (other library
    # versions) are tuned during the untimed warm-up epochs.
    params['config']['gemm_tuning_online'] = args.warmup >= 1
    # A/B measurements of optional code paths (tools/gpu_r2_call*.sh): RLG_BENCH_CONFIG='{"fused_loss": false}'
    overrides = json.loads(os.environ.get('RLG_BENCH_CONFIG', '{}'))
    if multi:
        # the in-graph all-reduce gives up (fail-safe, all ranks then raise) well inside the watchdog's bound
        params['config'].setdefault('native_allreduce_timeout_s', 120.0)
    params['config'].update(overrides)
    torch.manual_seed(42 + rank)
    trace_rows = None
    if os.environ.get('RLG_BENCH_ADAM_TRACE'):
        # diagnostic runs only (an instrumented library, tools/exp/build_trace_libs.sh): one row per optimiser launch
        sys.path.insert(0, os.path.join(ROOT, 'tools', 'exp'))
        import adam_trace
        trace_flags = int(os.environ['RLG_BENCH_ADAM_TRACE']) & 1
        trace_rows = adam_trace.install(device, 4096, trace_flags)
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
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    barrier()
    t0 = time.perf_counter()
    marks[0].record()
    for k in range(args.steps):
        agent.update_epoch()
        agent.train_epoch()
        marks[k + 1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    per_epoch_ms = [marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps)]

    in_sync = params_finite = None
    if multi:   # outside the timed region: every rank must hold bit-identical parameters and lr
        probe = torch.stack([agent.optimizer.flat_params.double().sum(),
                             agent.optimizer.flat_params.double().abs().sum(),
                             torch.tensor(agent.optimizer.last_and_next_lr()[1], dtype=torch.float64, device=device)])
        lo, hi = probe.clone(), probe.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        in_sync = bool(torch.equal(lo, hi))
        params_finite = bool(torch.isfinite(probe).all().item())
        if not in_sync:
            sys.stderr.write(f'bench.py: rank {rank}: parameter probe {probe.tolist()} (min over ranks {lo.tolist()}, max {hi.tolist()}), '
                             f'ipc status {agent._ipc_comm.status() if agent._ipc_comm else None}\n')

    if multi and os.environ.get('RLG_BENCH_SYNC_DIFF'):
        sys.path.insert(0, os.path.join(ROOT, 'tools', 'exp'))
        import adam_trace as _at
        _at.diff_report(agent, rank, world, out=lambda s: sys.stderr.write(s + '\n'))
    if trace_rows is not None:
        adam_trace.report(trace_rows, int(agent.optimizer.step_counter.item()), rank, world, trace_flags,
                          label=f'bench in_sync {in_sync}', out=lambda s: sys.stderr.write(s + '\n'))

    # the transports side by side - AFTER the timed region and the in-sync check, on communicators of their own that live
    # until the process exits (RLG_BENCH_PREFLIGHT=0 skips it)
    preflight = collective_preflight(agent, device, world) if (multi and os.environ.get('RLG_BENCH_PREFLIGHT', '1') != '0') else None

    pairs = agent.kernel_timers.get('gae_envmajor_fused', [])
    gae_each = [p.elapsed_us() for p in pairs]
    gae_us = sum(gae_each) / max(len(gae_each), 1)
    gae_bytes = envs * horizon * GAE_BYTES_PER_ENV_STEP
    achieved = gae_bytes / (gae_us * 1e-6) / 1e9 if gae_us > 0 else 0.0

    # second roofline: the f32-MFMA weight-gradient launch (one of the three large kernels of the
    # epoch).  Timed with HIP events around back-to-back launches on the last minibatch's own
    # operands, AFTER the timed region (inside it the launch is a node of a replayed HIP graph).
    mfma = None
    eng = getattr(agent, '_engine', None)
    if eng is not None and getattr(eng, 'last_dw_jobs', None):
        jobs, plan, mx = eng.last_dw_jobs
        # (the fp16 form takes its gradient scales from the maxima the step's backward left: still there - plain stores)
        kw = dict(maxima=mx) if mx is not None else {}
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            plan.launch(jobs, **kw)
        reps = 20
        ev0.record()
        for _ in range(reps):
            plan.launch(jobs, **kw)
        ev1.record()
        torch.cuda.synchronize()
        us = ev0.elapsed_time(ev1) * 1e3 / reps
        rows = jobs[0][0].shape[0]
        flops = sum(2.0 * rows * g.shape[0] * g.shape[1] for _, _, g in jobs)
        split = os.environ.get('RLG_DW_BF16', '1') != '0'
        f16 = split and mx is not None and os.environ.get('RLG_DW_F16', '1') != '0'
        useful = flops / us / 1e6                               # fp32 products per second, as TFLOP/s
        if split:
            # three fp16 (or six bf16) plane products per fp32 product (csrc/mlp_dw.hip): priced against the peak of the
            # instruction that issues them - v_mfma_f32_16x16x32_f16 and _bf16 have the same dense rate
            k = 3.0 if f16 else 6.0
            name = 'rlg::mlp_dw_f16x3_kernel' if f16 else 'rlg::mlp_dw_bf16x6_kernel'
            form = ('three exact fp16 plane products, gradients scaled by the power of two their largest magnitude over a '
                    "wave's rows asks for, activations by the forward's fixed scales" if f16 else 'six exact bf16 plane products')
            mfma = {'kernel': f'{name} + rlg::mlp_dw_finalize_kernel (all weight gradients, one launch pair; fp32 products as {form})',
                    'bound': 'mfma', 'achieved': k * useful, 'peak': BF16_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': k * useful / BF16_MFMA_PEAK_TFLOPS, 'traffic': None,
                    'algorithmic_flops_per_launch': k * flops, 'avg_launch_us': us, 'launches': reps,
                    'fp32_equivalent_tflops': useful, 'fp32_equivalent_over_fp32_mfma_peak': useful / FP32_MFMA_PEAK_TFLOPS,
                    'note': f'issued flops = {int(k)} x useful 2*rows*sum(No*Mi) (tile padding not counted) against the dense '
                            '16-bit MFMA peak (v_mfma_f32_16x16x32_f16 / _bf16, MI355X_MICROARCH.md); fp32_equivalent_* = the '
                            'useful fp32 products against the fp32 MFMA peak (157.3) that RLG_DW_BF16=0 would be '
                            'priced on; timed after the timed region'}
        else:
            mfma = {'kernel': 'rlg::mlp_dw_kernel + rlg::mlp_dw_finalize_kernel (all weight gradients, one launch pair)',
                    'bound': 'mfma', 'achieved': useful, 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': useful / FP32_MFMA_PEAK_TFLOPS, 'traffic': None,
                    'algorithmic_flops_per_launch': flops, 'avg_launch_us': us, 'launches': reps,
                    'note': 'useful flops 2*rows*sum(No*Mi) (tile padding not counted); dense fp32 MFMA peak '
                            '(v_mfma_f32_16x16x4_f32, MI355X_MICROARCH.md); timed after the timed region'}

    # rooflines of the two dominant kernels: one more epoch, eagerly (no HIP graphs), every chain dispatch bracketed
    # by HIP events; all ranks run it together (it contains the usual collectives)
    chain_roof = {}
    if eng is not None and getattr(eng, 'chain', None) is not None:
        from rl_games_amd import ops
        agent._hip_graphs = False
        agent.kernel_timers = None        # (the GAE launch of this extra epoch is not part of `roofline`)
        ops.chain_timers = {}
        agent.update_epoch()
        agent.train_epoch()
        torch.cuda.synchronize()
        timers, ops.chain_timers = ops.chain_timers, None
        ins, outs = eng.chain.ins, eng.chain.outs
        macs_f = sum(i * o for i, o in zip(ins, outs))
        macs_b = sum(i * o for i, o in zip(ins[1:], outs[1:]))
        mb_rows = global_mb // world
        for key, kind, rows, macs, name in (
                ('roofline_fwd', 'fwd_train', mb_rows, macs_f, 'rlg::mlp_chain_fwd_kernel (training forward: statistics '
                 'fold + normalise + every layer + heads, activations written)'),
                ('roofline_fwd_infer', 'fwd_infer', envs, macs_f, 'rlg::mlp_chain_fwd_kernel (rollout inference forward)'),
                ('roofline_bwd', 'bwd_loss', mb_rows, macs_b, 'rlg::mlp_chain_bwd_kernel (PPO loss tile + dX chain + '
                 'activation backward + bias partials)')):
            each = [p.elapsed_us() for p in timers.get(kind, [])]
            if not each:
                continue
            us = sum(each) / len(each)
            tf = 2.0 * rows * macs / us / 1e6
            chain_roof[key] = {
                'kernel': name, 'bound': 'mfma', 'achieved': tf, 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': tf / FP32_MFMA_PEAK_TFLOPS, 'traffic': None, 'rows': rows,
                'algorithmic_flops_per_launch': 2.0 * rows * macs, 'avg_launch_us': us, 'launches': len(each),
                'launch_us_min': min(each), 'launch_us_max': max(each),
                'timing': 'HIP start/stop events bound to every dispatch of this kernel in one eager epoch after the '
                          'timed region (= rocprofv3 --kernel-trace begin/end); exact fp32 products on '
                          'v_mfma_f32_16x16x4_f32, useful flops 2*rows*sum(in*out), tile padding not counted'}
            kp, ptype = ops.chain_split_form()
            pwords = {3: 'three', 6: 'six'}[kp]
            insn = f'v_mfma_f32_16x16x32_{"f16" if ptype == "fp16" else "bf16"}'
            if key in ('roofline_fwd', 'roofline_fwd_infer') and eng.chain.split_products(rows, 2 if key == 'roofline_fwd' else 0):
                pad = sum(-(-o // 16) * 16 * -(-i // 32) * 32 for i, o in zip(ins, outs))
                issued = kp * 2.0 * rows * pad / us / 1e6
                chain_roof[key].update({
                    'kernel': 'rlg::mlp_chain_fwd_bx_kernel (' + ('training forward: statistics fold + normalise + every layer '
                              '+ heads, activations written' if key == 'roofline_fwd' else 'rollout inference forward') +
                              f'; split-{ptype} products on pre-split weight planes)',
                    # priced on the unit the kernel issues on: k x padded-tile flops against the dense 16-bit MFMA peak
                    'achieved': issued, 'peak': BF16_MFMA_PEAK_TFLOPS, 'frac': issued / BF16_MFMA_PEAK_TFLOPS,
                    'algorithmic_flops_per_launch': kp * 2.0 * rows * pad, 'plane_products': kp,
                    'fp32_equivalent_tflops': tf, 'fp32_equivalent_over_fp32_mfma_peak': tf / FP32_MFMA_PEAK_TFLOPS,
                    'issued_tflops_bf16': issued, 'issued_frac_of_bf16_peak': issued / BF16_MFMA_PEAK_TFLOPS,
                    'timing': 'HIP start/stop events bound to every dispatch of this kernel in one eager epoch after the '
                              'timed region (= rocprofv3 --kernel-trace begin/end; the plane-pack launch in front of it '
                              f'is not included); {pwords} exact {ptype} plane products per fp32 product on '
                              f'{insn} (csrc/mlp_chain_bx_fwd.hip); achieved / frac = the ISSUED flops - '
                              f'{kp} x the padded-tile flops - against the dense 16-bit MFMA peak (the unit the kernel issues '
                              'on; fp16 and bf16 share it); fp32_equivalent_* = useful fp32 flops 2*rows*sum(in*out) against '
                              'the fp32 MFMA peak'})
            if key == 'roofline_bwd' and eng.chain.split_products(rows, 1):
                # the split-bf16 kernel: fp32-equivalent flops against the fp32 peak (comparable with the other rows)
                # and what it ISSUES - six bf16 plane products per product over 16 x 32 padded tiles - on the bf16 peak
                pad = sum(-(-i // 16) * 16 * -(-o // 32) * 32 for i, o in zip(ins[1:], outs[1:]))
                issued = kp * 2.0 * rows * pad / us / 1e6
                chain_roof[key].update({
                    'kernel': f'rlg::mlp_chain_bwd_bx_kernel (PPO loss tile + dX chain on split-{ptype} products + activation '
                              'backward + bias partials; weight planes written by the optimiser launch)',
                    'achieved': issued, 'peak': BF16_MFMA_PEAK_TFLOPS, 'frac': issued / BF16_MFMA_PEAK_TFLOPS,
                    'algorithmic_flops_per_launch': kp * 2.0 * rows * pad, 'plane_products': kp,
                    'fp32_equivalent_tflops': tf, 'fp32_equivalent_over_fp32_mfma_peak': tf / FP32_MFMA_PEAK_TFLOPS,
                    'issued_tflops_bf16': issued, 'issued_frac_of_bf16_peak': issued / BF16_MFMA_PEAK_TFLOPS,
                    'timing': 'HIP start/stop events bound to every dispatch of this kernel in one eager epoch after the '
                              f'timed region (= rocprofv3 --kernel-trace begin/end); {pwords} exact {ptype} plane products per fp32 '
                              f'product on {insn} (csrc/mlp_chain_bx.hip); achieved / frac = the ISSUED '
                              f'flops - {kp} x the padded-tile flops - against the dense 16-bit MFMA peak; fp32_equivalent_* = '
                              'useful fp32 flops 2*rows*sum(in*out) against the fp32 MFMA peak'})

    # what arithmetic the forward / backward chain launches of the timed region ran on
    chain_products = None
    if eng is not None and getattr(eng, 'chain', None) is not None:
        mbr = global_mb // world
        from rl_games_amd import ops as _ops
        kp, ptype = _ops.chain_split_form()
        form = {True: (f'split-{ptype} ({kp} exact {ptype} plane products per fp32 product on '
                       f'v_mfma_f32_16x16x32_{"f16" if ptype == "fp16" else "bf16"}, fp32 accumulation; dropped terms and plane '
                       'rounding <= 3*2^-24 |x||w|' + ('; operands scaled by powers of two: rows of raw inputs and of '
                       'gradient tiles by their own maxima, weights / hidden activations / normalised observations by fixed ones)'
                       if ptype == 'fp16' else ')')),
                False: 'exact fp32 (v_mfma_f32_16x16x4_f32)'}
        def launch_form(rows, direction):
            text = form[bool(eng.chain.split_products(rows, direction))]
            if eng.chain.lean_used(rows, direction):
                # (a data-parallel rank's sizes: 16-row workgroups reading the weights as fp32 fragments in each wave's
                #  consumption order, csrc/mlp_chain.hip - bit-identical to the pipelined 16-row kernels)
                text += ', lean 16-row kernel'
            return text
        chain_products = {'training_forward': launch_form(mbr, 2), 'backward': launch_form(mbr, 1),
                          'rollout_forward': launch_form(envs, 0)}
    for key in ('roofline_fwd', 'roofline_fwd_infer', 'roofline_bwd'):
        r = chain_roof.get(key)
        if r is not None and 'issued_tflops_bf16' in r:
            # the scheme's own ceiling: k 16-bit plane products per useful fp32 product
            r['split_ceiling_tflops'] = BF16_MFMA_PEAK_TFLOPS / r['plane_products']
            r['split_ceiling_frac'] = r['fp32_equivalent_tflops'] / (BF16_MFMA_PEAK_TFLOPS / r['plane_products'])

    traffic, traffic_note = None, 'no rocprofv3 --pmc record for this workload'
    try:
        with open(os.path.join(ROOT, 'profiles', 'gae_pmc_traffic.json')) as f:
            rec = json.load(f).get(f'{envs}x{horizon}')
        if rec:
            traffic = rec.get('traffic_bytes')
            traffic_note = rec.get('note', 'rocprofv3 --pmc FETCH_SIZE(x2, gfx950 correction) + WRITE_SIZE per launch')
    except Exception:
        pass

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        out = {
            'metric': 'ppo_env_steps_per_sec', 'value': global_envs * horizon * args.steps / elapsed,
            'unit': 'env-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'ms_per_step_stats': {'min': min(per_epoch_ms), 'median': statistics.median(per_epoch_ms),
                                  'mean': sum(per_epoch_ms) / len(per_epoch_ms), 'max': max(per_epoch_ms),
                                  'each': [round(v, 3) for v in per_epoch_ms],
                                  'note': 'per-epoch HIP event intervals on rank 0'},
            'config': {
                'workload': w['desc'],
                'global_envs': global_envs, 'horizon': horizon, 'envs_per_gpu': envs,
                'minibatch_per_gpu': global_mb // world, 'mini_epochs': w['mini_epochs'],
                'optimizer_steps_per_epoch': w['mini_epochs'] * (envs * horizon) // (global_mb // world),
                'parallelism': f'dp{world}', 'step': 'one train_epoch (rollout+GAE+dataset+update)',
                'lr_schedule': 'adaptive (device side)', 'mixed_precision': False,
                'mlp': 'fused chain kernels' if (eng is not None and getattr(eng, 'chain', None) is not None)
                       else 'per-layer engine',
                'chain_products': chain_products,
                'weight_gradient_products': (mfma['kernel'].split(' ')[0] if mfma else None),
            },
            'roofline': {
                'kernel': f'rlg::gae_envmajor_kernel<{horizon},false> (GAE + returns + advantages + fp64 moments)',
                'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_note': traffic_note,
                'algorithmic_bytes_per_launch': gae_bytes, 'avg_launch_us': gae_us, 'launches': len(pairs),
                'launch_us_min': min(gae_each) if gae_each else None, 'launch_us_max': max(gae_each) if gae_each else None,
                'timing': 'HIP start/stop events bound to each in-epoch GAE dispatch on its launch stream '
                          '(hipExtLaunchKernelGGL) = the dispatch begin/end timestamps rocprofv3 --kernel-trace '
                          'reports; timed region',
                'note': 'one generation of waves over 35.65 MB, inputs cold (written 32 rollout steps earlier): its load and '
                        'store phases each run near the memory system\'s limit, ~3 us are dispatch ramp + first-byte latency + '
                        'drain (a same-footprint copy kernel takes 8.9 us cold); >= 0.60 (<= 7.4 us) is not reachable for '
                        'this launch inside the epoch - profiles/r6_gae_in_situ.txt, r2_gae_residency_experiments.txt',
            },
        }
        if mfma is not None:
            out['roofline_mfma'] = mfma
        out.update(chain_roof)
        if in_sync is not None:
            comm = agent._ipc_comm or None
            out['config']['ranks_in_sync'] = in_sync
            out['config']['params_finite'] = params_finite
            out['config']['allreduce'] = ((getattr(comm, 'kind', None) or ('ipc-two-phase' if comm.two_phase else 'ipc'))
                                          if comm is not None else 'rccl')
            out['config']['allreduce_note'] = (
                'in-graph hipIpc all-reduce kernel (csrc/ipc_allreduce.hip), inside the mini-epoch HIP graph'
                if comm is not None else 'RCCL all-reduce via torch.distributed between two graph replays per step'
                + ('' if params['config'].get('native_allreduce', True) is False
                   else ' (FALLBACK: the hipIpc communicator could not be created or failed its self-test)'))
            out['config']['ipc_self_test'] = ('passed' if comm is not None else
                                              ('not requested' if params['config'].get('native_allreduce', True) is False
                                               else 'failed'))
            out['config']['hsa_enable_ipc_mode_legacy'] = os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')
        if preflight is not None:
            out['config']['collective_check'] = preflight
        if (world == 1 and args.workload == 'humanoid' and not args.no_exact_row and not overrides
                and os.environ.get('RLG_BENCH_CHILD') != '1' and os.environ.get('RLG_CHAIN_BX', '1') != '0'):
            # the same job on exact fp32 products in all three MFMA launches, measured the same way right after the
            # timed region (a child process: the product form is a process-wide setting of the library)
            try:
                child = subprocess.run(
                    [sys.executable, os.path.abspath(__file__), '--steps', '3', '--warmup', '2', '--no-cpu-baseline',
                     '--no-exact-row'], env=dict(os.environ, RLG_CHAIN_BX='0', RLG_DW_BF16='0', RLG_BENCH_CHILD='1'),
                    capture_output=True, text=True, timeout=600)
                line = [l for l in child.stdout.splitlines() if l.startswith('{')][-1]
                row = json.loads(line)
                out['exact_products_ms_per_step'] = row['ms_per_step']
                out['exact_products_env_steps_per_s'] = row['value']
                out['exact_products_note'] = ('same workload with RLG_CHAIN_BX=0 RLG_DW_BF16=0 (every MFMA launch on exact '
                                              'fp32 products, v_mfma_f32_16x16x4_f32), 2 warm-up + 3 timed epochs in a child '
                                              'process after the timed region')
            except Exception as e:
                out['exact_products_ms_per_step'] = None
                out['exact_products_note'] = f'child run failed: {type(e).__name__}: {e}'
        if world == 1 and not args.no_cpu_baseline:
            sample = args.cpu_sample_envs or (4096 if args.workload == 'humanoid' else 1024)
            signal.alarm(0)          # the CPU baseline is bounded by construction
            out['cpu_baseline'] = cpu_baseline(args.workload, sample)
            out['gpu_over_cpu'] = out['value'] / out['cpu_baseline']['value']
        print(json.dumps(out))
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
