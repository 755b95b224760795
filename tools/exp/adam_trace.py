"""Host side of the optimiser-launch trace (csrc/adam_trace.hpp; libraries built by tools/exp/build_trace_libs.sh):
install() hands the library a zeroed row buffer BEFORE the agent captures its graphs, report() gathers every rank's rows
and says which quantity of which step differs first - across the ranks, between a rank's "as stored" checksums of step s
and its "as loaded" checksums of step s + 1, and between plain and coherent (system-scope) loads.  Used by
tools/exp/adam_trace_probe.py and, with RLG_BENCH_ADAM_TRACE=1, by bench.py itself."""
import ctypes
import os

import torch
import torch.distributed as dist

WORDS = 32
NAMES = ['step', 'clip|norm', 'lr', 'kl|scale', 'in.g', 'in.p', 'in.m', 'in.v', 'out.g', 'out.p', 'out.m', 'out.v',
         'coh.g', 'coh.p', 'coh.m', 'coh.v', 'stale_count']


def install(device, cap, flags):
    from rl_games_amd import _lib
    lib = _lib.load()
    fn = getattr(lib, 'rlg_debug_adam_trace', None)
    if fn is None:
        raise SystemExit('this library has no rlg_debug_adam_trace: run tools/exp/build_trace_libs.sh and set RLG_HIP_LIB')
    rows = torch.zeros(cap, WORDS, dtype=torch.int64, device=device)
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    fn(rows.data_ptr(), cap, flags)
    return rows


def report(rows, nsteps, rank, world, flags, label='', out=print):
    mine = rows[:nsteps].cpu()
    gathered = [torch.zeros_like(mine) for _ in range(world)] if rank == 0 else None
    if world > 1:
        dist.gather(mine, gathered, dst=0)
    else:
        gathered = [mine]
    if rank != 0:
        return
    g = torch.stack(gathered)                                    # [world, steps, 32]
    out(f'TRACE {label} world {world} steps {nsteps} flags {flags}')
    ndiff = 0
    for s in range(nsteps):
        cols = [c for c in range(17) if any(int(g[r, s, c]) != int(g[0, s, c]) for r in range(1, world))]
        if cols:
            ndiff += 1
            if ndiff <= 6:
                out(f'  step {s + 1}: ranks differ in {[NAMES[c] for c in cols]}')
                for c in cols[:6]:
                    out(f'      {NAMES[c]} {[hex(int(g[r, s, c]) & (2 ** 64 - 1)) for r in range(world)]}')
    out(f'  steps with a cross-rank difference: {ndiff} of {nsteps}')
    for r in range(world):
        lost = []
        for s in range(nsteps - 1):
            for c, name in ((1, 'p'), (2, 'm'), (3, 'v')):
                if int(g[r, s, 8 + c]) != int(g[r, s + 1, 4 + c]):
                    lost.append((s + 1, name))
        stale = [(s + 1, int(g[r, s, 16])) for s in range(nsteps) if int(g[r, s, 16]) != 0]
        coh = []
        if flags & 1:
            for s in range(nsteps):
                for c, name in ((0, 'g'), (1, 'p'), (2, 'm'), (3, 'v')):
                    if int(g[r, s, 4 + c]) != int(g[r, s, 12 + c]):
                        coh.append((s + 1, name))
        missing = [s + 1 for s in range(nsteps) if int(g[r, s, 0]) != s + 1]
        out(f'  rank {r}: stored(s) != loaded(s+1): {len(lost)} {lost[:8]}; plain != coherent elements: {stale[:8]}; '
            f'checksum plain != coherent: {len(coh)} {coh[:8]}; rows without their step number: {len(missing)} {missing[:6]}')
        for s, cnt in stale[:3]:
            recs = []
            for j in range(min(cnt, 7)):
                w0, w1 = int(g[r, s - 1, 17 + 2 * j]) & (2 ** 64 - 1), int(g[r, s - 1, 18 + 2 * j]) & (2 ** 64 - 1)
                recs.append((w0 & (2 ** 48 - 1), 'gpmv'[w0 >> 48], hex(w1 & 0xffffffff), hex(w1 >> 32)))
            out(f'     rank {r} step {s}: (index, array, plain, coherent) {recs}')


def diff_report(agent, rank, world, out=print):
    """Element-level comparison of rank 1's optimiser arrays with rank 0's: how many elements differ, where (parameter
    tensor, row, column), by how much."""
    opt = agent.optimizer
    names = [(n, p) for n, p in agent.model.named_parameters()]
    spans = []
    for (n, p), (off, cnt) in zip(names, opt.offsets):
        spans.append((off, cnt, n, tuple(p.shape)))
    for name, t in (('params', opt.flat_params), ('exp_avg', opt.exp_avg), ('exp_avg_sq', opt.exp_avg_sq),
                    ('grads', opt.grads)):
        mine = t.detach().cpu()
        if rank == 1:
            dist.send(mine, dst=0)
        elif rank == 0:
            other = mine.clone()
            dist.recv(other, src=1)
            d = (mine.view(torch.int32) != other.view(torch.int32)).nonzero().flatten()
            if d.numel() == 0:
                out(f'  DIFF {name}: identical')
                continue
            ulps = (mine.view(torch.int32)[d].long() - other.view(torch.int32)[d].long()).abs()
            out(f'  DIFF {name}: {d.numel()} of {mine.numel()} elements differ; |bit distance| min {int(ulps.min())} median '
                f'{int(ulps.median())} max {int(ulps.max())}; max rel {float(((mine[d] - other[d]).abs() / other[d].abs().clamp_min(1e-30)).max()):.2e}')
            out(f'      values (rank 0, rank 1, rel): ' + ', '.join(f'({float(mine[i]):.6e}, {float(other[i]):.6e}, '
                f'{float((mine[i] - other[i]) / other[i]):+.2e})' for i in d[:6].tolist()))
            where = []
            for i in d[:400].tolist():
                for off, cnt, n, shape in spans:
                    if off <= i < off + cnt:
                        loc = i - off
                        where.append((n, loc // shape[-1] if len(shape) == 2 else 0, loc % shape[-1]))
                        break
            by = {}
            for n, r, c in where:
                by.setdefault(n, []).append((r, c))
            for n, rc in by.items():
                rows_ = sorted(set(r for r, _ in rc))
                cols_ = sorted(set(c for _, c in rc))
                out(f'      {n}: {len(rc)} elements; rows {rows_[:24]}{"..." if len(rows_) > 24 else ""} cols '
                    f'{cols_[:24]}{"..." if len(cols_) > 24 else ""}; first {rc[:6]}')
