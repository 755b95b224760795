"""CPU, world_size 2, gloo: the data-parallel exchange steps of the hot path.

  * the flat gradient arena + KL tail slot through ONE all_reduce(SUM) equals the reference's
    cat -> all_reduce -> /world -> copy-back (a2c_common.py:493-509) plus its separate KL
    all-reduce (:1560) - checked against per-rank gradients computed independently;
  * distributed.StatsSync (pooled moment deltas, a2c_common.py:61-93, all normalisers in ONE collective)
    reproduces the statistics of the pooled stream exactly like the reference's own test
    (tests/test_multigpu_stats_sync.py:20-33, :97-115), does not re-count shared history, and equals the
    reference's per-module merge_rank_stats bit for bit;
  * broadcast mode leaves every rank with rank 0's statistics.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

from oracle import ppo_oracle as O


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _spawn(fn, world=2, *args):
    port = _free_port()
    mp.spawn(_entry, args=(world, port, fn, args), nprocs=world, join=True)


def _entry(rank, world, port, fn, args):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        fn(rank, world, *args)
    finally:
        dist.destroy_process_group()


class _Stats(nn.Module):
    """Buffers of a RunningMeanStd (normalizers.py / running_mean_std.py:46-53) without kernels."""

    def __init__(self, n):
        super().__init__()
        self.register_buffer('running_mean', torch.zeros(n, dtype=torch.float64))
        self.register_buffer('running_var', torch.ones(n, dtype=torch.float64))
        self.register_buffer('count', torch.ones((), dtype=torch.int64))

    def feed(self, x):
        st = {'running_mean': self.running_mean, 'running_var': self.running_var, 'count': self.count}
        _, st = O.running_stats_forward(st, x, training=True)
        self.running_mean, self.running_var, self.count = st['running_mean'], st['running_var'], st['count']


def _grad_worker(rank, world):
    from rl_games_amd import distributed as rdist
    from rl_games_amd.flat_optim import FlatArena
    torch.manual_seed(0)                       # identical parameters on every rank
    net = nn.Sequential(nn.Linear(6, 5), nn.ELU(), nn.Linear(5, 3))
    ref = [p.detach().clone() for p in net.parameters()]
    arena = FlatArena(net.parameters())
    for p, r in zip(net.parameters(), ref):
        assert torch.equal(p.data, r) and p.grad.data_ptr() != 0
    g = torch.Generator().manual_seed(100 + rank)      # rank-local minibatch
    x = torch.randn(32, 6, generator=g)
    arena.zero_grad()
    loss = net(x).pow(2).mean()
    loss.backward()
    arena.kl_slot.fill_(0.01 * (rank + 1))
    local = arena.grads.clone()
    rdist.all_reduce_sum(arena.flat_grads)
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    expect = torch.stack(gathered).sum(0)
    assert torch.allclose(arena.grads, expect, rtol=1e-6, atol=1e-8)
    assert abs(arena.kl_slot.item() - 0.01 * sum(range(1, world + 1))) < 1e-7
    # averaged gradient == mean of per-rank gradients, visible through every param.grad view
    off = 0
    for p in net.parameters():
        n = p.numel()
        assert torch.allclose(p.grad.reshape(-1) / world, expect[off:off + n] / world)
        off += n


def _sync(mods):
    """The product's host orchestration (flat layout, one collective, snapshot bookkeeping) with the
    oracle's arithmetic standing in for the two device kernels (there is no GPU in this test)."""
    from rl_games_amd import distributed as rdist
    return rdist.StatsSync(mods, kernels=O.StatsSyncOracle(mods))


def _stats_worker(rank, world):
    from rl_games_amd import distributed as rdist
    g = torch.Generator().manual_seed(7)
    data = [torch.randn(200 + 50 * r, 4, generator=g) * (1 + r) + r for r in range(world)]
    vdata = [torch.randn(300 + 10 * r, 1, generator=g) * 3 - r for r in range(world)]
    m, v = _Stats(4), _Stats(1)
    m.feed(data[rank])
    v.feed(vdata[rank])
    sync = _sync([m, v])
    calls = []

    def counted_all_reduce(t):
        calls.append(t.numel())
        return rdist.all_reduce_sum(t)
    sync.merge(counted_all_reduce)
    assert calls == [(1 + 2 * 4) + (1 + 2 * 1)]          # ONE collective for both normalisers
    # pooled reference: prior (count 1, mean 0, var 1) per rank + all samples
    for mod, chunks in ((m, data), (v, vdata)):
        n = sum(d.shape[0] for d in chunks) + world
        s1 = sum(d.double().sum(0) for d in chunks)
        s2 = sum((d.double() ** 2).sum(0) for d in chunks) + world * 1.0
        mean = s1 / n
        var = s2 / n - mean ** 2
        assert mod.count.item() == n
        assert torch.allclose(mod.running_mean, mean, atol=1e-5)
        assert torch.allclose(mod.running_var, var, atol=1e-4)
    n = m.count.item()
    # second epoch: only the NEW data is summed across ranks (no geometric count growth)
    extra = [torch.randn(100, 4, generator=g) for _ in range(world)]
    m.feed(extra[rank])
    sync.merge(rdist.all_reduce_sum)
    assert m.count.item() == n + 100 * world
    # every rank ends with the same statistics, bit for bit
    gathered = [torch.zeros_like(m.running_mean) for _ in range(world)]
    dist.all_gather(gathered, m.running_mean)
    assert all(torch.equal(gathered[0], t) for t in gathered)
    # restored stats are shared history: seeding the snapshot must stop them being re-summed
    m2 = _Stats(4)
    m2.feed(data[0])
    s2_ = _sync([m2])
    s2_.seed()
    before = (m2.count.item(), m2.running_mean.clone(), m2.running_var.clone())
    s2_.merge(rdist.all_reduce_sum)
    assert m2.count.item() == before[0]
    assert torch.allclose(m2.running_mean, before[1], atol=1e-12) and torch.allclose(m2.running_var, before[2], atol=1e-9)
    # broadcast mode
    m3 = _Stats(4)
    m3.feed(data[rank])
    _sync([m3]).adopt_rank0(rdist.broadcast_from_rank0)
    ref = _Stats(4)
    ref.feed(data[0])
    assert torch.equal(m3.running_mean, ref.running_mean) and torch.equal(m3.running_var, ref.running_var)
    assert m3.count.item() == ref.count.item()


def _ranks_in_sync_worker(rank, world):
    """distributed.ranks_in_sync: identical bits -> True on every rank; ONE element one ulp apart on one rank (the
    round-5 desynchronisation: 16 exp_avg_sq elements one update apart) -> False on every rank, in one collective; -0.0
    against +0.0 counts as different (bits, not values)."""
    from rl_games_amd import distributed as rdist
    g = torch.Generator().manual_seed(3)
    a, b = torch.randn(1000, generator=g), torch.rand(77, generator=g)
    assert rdist.ranks_in_sync([a, b])
    b2 = b.clone()
    if rank == 1:
        b2[13] = torch.nextafter(b2[13], torch.tensor(2.0))
    assert not rdist.ranks_in_sync([a, b2])
    assert rdist.ranks_in_sync([a])
    z = torch.zeros(4)
    if rank == 0:
        z[2] = -0.0
    assert not rdist.ranks_in_sync([z])
    # two differences that cancel in a plain sum of VALUES do not cancel in the checksum of bit patterns
    c = torch.ones(8)
    if rank == 1:
        c[0], c[1] = 1.5, 0.5
    assert not rdist.ranks_in_sync([c])


def test_ranks_in_sync_checksum_two_ranks():
    _spawn(_ranks_in_sync_worker, 2)


def test_flat_arena_allreduce_two_ranks():
    _spawn(_grad_worker, 2)


def test_running_stats_merge_two_ranks():
    _spawn(_stats_worker, 2)


def test_merge_with_injected_collective():
    """tests/test_multigpu_stats_sync.py:14-17 style: emulate two identical ranks with t.mul_(2)."""
    m = _Stats(3)
    x = torch.randn(500, 3, generator=torch.Generator().manual_seed(1))
    m.feed(x)
    mean, var = m.running_mean.clone(), m.running_var.clone()
    _sync([m]).merge(lambda t: t.mul_(2))
    assert m.count.item() == 2 * 501
    assert torch.allclose(m.running_mean, mean, atol=1e-12)
    assert torch.allclose(m.running_var, var, atol=1e-9)


def test_flat_merge_equals_reference_per_module_merge():
    """The flat-buffer merge (one collective for all normalisers) against the REFERENCE's own
    merge_rank_stats / seed_stats_sync_snapshot (imported when /root/reference exists, else the oracle's
    pooled_merge restatement), bit for bit over three epochs."""
    import sys
    fake_world = 3
    try:
        sys.path.insert(0, '/root/reference')
        from rl_games.common.a2c_common import merge_rank_stats as ref_merge
    except Exception:
        ref_merge = None
    finally:
        if sys.path[0] == '/root/reference':
            sys.path.pop(0)
    g = torch.Generator().manual_seed(3)
    ours = [_Stats(5), _Stats(1)]
    theirs = [_Stats(5), _Stats(1)]
    sync = _sync(ours)
    snaps = [None, None]
    for epoch in range(3):
        for k, d in enumerate((5, 1)):
            x = torch.randn(64, d, generator=g) * (k + 2) + epoch
            ours[k].feed(x)
            theirs[k].feed(x)
        sync.merge(lambda t: t.mul_(fake_world))
        for k, m in enumerate(theirs):
            if ref_merge is not None:
                ref_merge(m, lambda t: t.mul_(fake_world))
            else:
                st = {'count': m.count, 'running_mean': m.running_mean, 'running_var': m.running_var}
                new, snaps[k] = O.pooled_merge(st, snaps[k], lambda t: t.mul_(fake_world))
                m.count.copy_(new['count'])
                m.running_mean.copy_(new['running_mean'])
                m.running_var.copy_(new['running_var'])
        for a, b in zip(ours, theirs):
            assert a.count.item() == b.count.item()
            assert torch.equal(a.running_mean, b.running_mean) and torch.equal(a.running_var, b.running_var)


def test_stats_sync_mode_validation():
    from rl_games_amd import distributed as rdist
    assert rdist.resolve_stats_sync_mode('pooled') == 'pooled'
    with pytest.raises(ValueError):
        rdist.resolve_stats_sync_mode('ring')


# ----------------------------------------------------------------------------- the oracle's data-parallel step (round 5)

def _dp_oracle_agents(world):
    """One OracleAgent per rank on its own env shard (different seeds), the same parameters, datasets prepared."""
    import copy
    from oracle.ppo_epoch_oracle import OracleAgent
    from rl_games_amd import configs
    from rl_games_amd.synthetic_env import SyntheticTensorEnv
    params = configs.tiny(num_actors=64, horizon=8, obs_dim=7, act_dim=3, device='cpu')
    agents = []
    for r in range(world):
        env = SyntheticTensorEnv(64, 7, 3, device='cpu', seed=20 + r)
        a = OracleAgent(copy.deepcopy(params), env, seed=5)         # (seed 5 everywhere: identical initial parameters)
        a.obs = env.reset()
        torch.manual_seed(100 + r)                                  # rank-local action noise
        a.prepare_dataset(a.play_steps())
        agents.append(a)
    return agents


def _dp_oracle_worker(rank, world):
    """oracle.ppo_epoch_oracle.data_parallel_minibatch_step against the REAL reference method: every rank computes its
    own minibatch gradients with the oracle, then runs rl_games' A2CBase.trancate_gradients_and_step (a2c_common.py:
    493-514: cat -> all_reduce(SUM) -> / world_size -> copy back -> clip_grad_norm_ -> optimizer.step) under gloo on a
    stand-in object with exactly the attributes that method reads.  The oracle's single-process restatement of the same
    step must leave the same gradients and parameters, bit for bit, for all steps of a mini-epoch."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, 'golden'))
    import ref_import
    ref_import.enable()
    from rl_games.common import a2c_common
    from oracle.ppo_epoch_oracle import data_parallel_minibatch_step
    import types
    mine = _dp_oracle_agents(world)[rank]               # the reference run: this rank's agent only
    both = _dp_oracle_agents(world)                     # the oracle run: all ranks in this process
    assert all(torch.equal(p, q) for p, q in zip(both[0].model.a2c_network.parameters(),
                                                 both[1].model.a2c_network.parameters()))
    cfg = mine.cfg
    stub = types.SimpleNamespace(multi_gpu=True, world_size=world, model=mine.model.a2c_network,
                                 truncate_grads=cfg['truncate_grads'], grad_norm=cfg['grad_norm'], optimizer=mine.optimizer)
    for i in range(mine.B // mine.mb):
        out = mine.minibatch_backward(i)
        a2c_common.A2CBase.trancate_gradients_and_step(stub)
        kl = out['kl'].clone()
        dist.all_reduce(kl)                                             # a2c_common.py:1559-1561
        kl /= world
        want = data_parallel_minibatch_step(both, i)
        assert torch.equal(kl, want[rank]['kl_mean'])
        for k in ('a_loss', 'c_loss', 'entropy', 'b_loss', 'kl'):
            assert torch.equal(out[k], want[rank][k]), (i, k)
        mine_lr = mine.lr
        # the lr rule on the averaged KL (python floats), like minibatch_apply does for the oracle ranks
        mine.lr = O.adaptive_lr(mine.lr, kl.item(), cfg['kl_threshold'], cfg.get('min_lr', 1e-6), cfg.get('max_lr', 1e-2),
                                cfg.get('lr_multiplier', 1.5))
        for g in mine.optimizer.param_groups:
            g['lr'] = mine.lr
        assert want[rank]['lr'] == mine_lr and both[rank].lr == mine.lr
        for (n, p), q in zip(mine.model.a2c_network.named_parameters(), both[rank].model.a2c_network.parameters()):
            assert torch.equal(p.grad, q.grad), (i, n)                  # averaged, clipped
            assert torch.equal(p, q), (i, n)                            # stepped
    # the ranks of the oracle run stay identical
    for p, q in zip(both[0].model.a2c_network.parameters(), both[1].model.a2c_network.parameters()):
        assert torch.equal(p, q)


def test_oracle_data_parallel_step_equals_the_reference_method_two_ranks():
    sys_path_ok = os.path.isdir('/root/reference/rl_games') or os.path.isfile(
        os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle', '_ref', 'rl_games_ref.zip'))
    if not sys_path_ok:
        pytest.skip('reference not present')
    _spawn(_dp_oracle_worker, 2)
