"""Static audit of the MFMA instructions in the built library (runs anywhere, no GPU).

    python tools/audit_mfma.py [rl_games_amd/librlg_hip.so]

What it guards.  On gfx950 the result registers of an MFMA must not be read by a non-MFMA instruction
(VALU incl. v_accvgpr_read, LDS / global stores, addresses) - nor as SrcA / SrcB of another MFMA - before
`passes + 2` (f32 operands) / `passes + 3` (bf16 / f16 / 8-bit operands) wait states have gone by; only the
SrcC dependency of a following MFMA is interlocked by the hardware.  Measured with
tools/exp/mfma_valu_read_probe.hip: v_mfma_f32_16x16x4_f32 needs 10 wait states before the first read of
a result register other than the first one, v_mfma_f32_16x16x32_bf16 needs 8 before any read; earlier
reads return the OLD register contents, silently.  hipcc (ROCm 7.2) inserts the s_nops in straight-line
code but MISSED them across a branch: in the 8-wave backward instance of csrc/mlp_chain.hip the last
MFMAs of a K-tail case were followed by `s_cbranch_execnz` to the merge point, whose first instruction
(v_pk_add_f32 of accumulator registers 2,3) ran 2 wait states after the MFMA - fragment registers 2
and 3 came out wrong for K <= 32 (tools/exp/chain_nopin_repro.py reproduces it).  The overlap of an MFMA
destination with its own sources, suspected first, is harmless (tools/exp/mfma_overlap_probe.hip,
mfma_chain_probe.hip: every form gives identical results, dependent chains included).

The audit walks every control-flow path behind each MFMA (branch targets decoded from the
disassembly) and fails on a read of a result register inside the window.  A wait state is one issue slot of the
wave (4 cycles: the measured windows, 10 and 8, are the 40- and 32-cycle latencies of the two shapes); they
are counted the way LLVM's hazard recognizer does - one per instruction, N + 1 for s_nop N - except that a
following MFMA cannot start before the passes of the audited one have left the matrix core.  Exit status 1
and a listing.

Second guard (round 4): a VGPR written by a VALU instruction must not be read by an MFMA (SrcA / SrcB / SrcC) before 2
wait states have gone by.  hipcc places the `s_nop 1` itself - for instructions it can see: rounds 2 - 3 produced the
bf16 planes with `v_cvt_pk_bf16_f32` inside an asm statement, 78 MFMAs of the weight-gradient kernel ran one wait
state behind the conversion of their operand, and its (2, 2) tile picked up the previous batch's plane now and then
(csrc/split_bf16.hpp; tools/exp/determinism_probe.py).  The audit flags every such pair whatever produced it."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'

LINE = re.compile(r'^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):')
REG = re.compile(r'\b([av])(?:\[(\d+):(\d+)\]|(\d+))')
# passes of the MFMA shapes the kernels may use (4 cycles each); wait states = passes + 2 / + 3
PASSES = (
    (re.compile(r'v_mfma_f32_32x32x2_?f32'), 16, 2),
    (re.compile(r'v_mfma_f32_16x16x4_?f32'), 8, 2),    # measured: 10
    (re.compile(r'v_mfma_f32_32x32x1_'), 16, 2),
    (re.compile(r'v_mfma_f32_16x16x1_'), 8, 2),
    (re.compile(r'v_mfma_f32_4x4x1_'), 2, 2),
    (re.compile(r'v_mfma_f32_32x32x16_'), 16, 3),
    (re.compile(r'v_mfma_f32_16x16x32_'), 8, 0),      # measured: 8 (hipcc leaves 8)
    (re.compile(r'v_mfma_f32_32x32x8_'), 16, 3),
    (re.compile(r'v_mfma_f32_16x16x16_'), 8, 3),
)
STORES = ('global_store', 'buffer_store', 'flat_store', 'scratch_store', 'ds_write', 'ds_store', 'global_atomic',
          'buffer_atomic', 'flat_atomic', 'ds_add', 'ds_max', 'ds_min')
READS_DST = ('v_fmac', 'v_mac', 'v_pk_fmac', 'v_dot2c', 'v_dot4c', 'v_dot8c', 'v_movrel', 'v_writelane',
             'v_cndmask')     # (v_cndmask listed only to stay conservative with tied forms)


def passes_of(mnemonic):
    for pat, passes, _ in PASSES:
        if pat.match(mnemonic):
            return passes
    return 1


def _window(mnemonic):
    for pat, passes, extra in PASSES:
        if pat.match(mnemonic):
            return passes + extra
    return 19 if mnemonic.startswith('v_mfma') or mnemonic.startswith('v_smfmac') else 0


def _regs(text):
    out = set()
    for m in REG.finditer(text):
        k, lo, hi, one = m.groups()
        if one is not None:
            out.add((k, int(one)))
        else:
            out.update((k, r) for r in range(int(lo), int(hi) + 1))
    return out


def _split_operands(ops):
    return [o.strip() for o in ops.split(',')] if ops else []


def _parse(text, names=None):
    """-> list of functions, each a list of (address, mnemonic, operand text); `names` (a list) receives their symbols."""
    funcs, cur = [], None
    for line in text.splitlines():
        head = re.match(r'^[0-9a-f]+ <(.*)>:$', line)
        if head:
            cur = []
            funcs.append(cur)
            if names is not None:
                names.append(head.group(1))
            continue
        m = LINE.match(line)
        if m and cur is not None:
            cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return funcs


def _successors(func):
    index = {addr: i for i, (addr, _, _) in enumerate(func)}
    succ = []
    for i, (addr, mn, ops) in enumerate(func):
        s = []
        if mn in ('s_endpgm', 's_setpc_b64', 's_swappc_b64'):
            succ.append(s)
            continue
        if mn.startswith('s_branch') or mn.startswith('s_cbranch'):
            simm = int(ops.split()[0])
            if simm >= 1 << 15:
                simm -= 1 << 16
            t = index.get(addr + 4 + 4 * simm)
            if t is not None:
                s.append(t)
            if mn.startswith('s_cbranch') and i + 1 < len(func):
                s.append(i + 1)
        elif i + 1 < len(func):
            s.append(i + 1)
        succ.append(s)
    return succ


def _sources_and_dests(mn, ops):
    """-> (registers read, registers written) of a non-MFMA instruction (conservative)."""
    o = _split_operands(ops)
    if not o:
        return set(), set()
    if mn.startswith(STORES):
        return _regs(ops), set()
    dst = _regs(o[0])
    src = set()
    for t in o[1:]:
        src |= _regs(t)
    if mn.startswith(READS_DST):
        src |= dst
    return src, dst


def audit_function(func):
    """-> (number of MFMAs, [(mfma line, reader line, wait states)])."""
    succ = _successors(func)
    count, bad = 0, []
    for i, (addr, mn, ops) in enumerate(func):
        win = _window(mn)
        if not win:
            continue
        count += 1
        o = _split_operands(ops)
        live0 = frozenset(_regs(o[0]))
        # breadth-first over (instruction, wait states so far, result registers not yet overwritten)
        seen, todo = set(), [(j, 0, live0) for j in succ[i]]
        while todo:
            j, ws, live = todo.pop()
            if ws >= win or not live or (j, ws, live) in seen:
                continue
            seen.add((j, ws, live))
            _, mn2, ops2 = func[j]
            step = 1
            if mn2 == 's_nop':
                step = int(ops2.split()[0]) + 1
            elif _window(mn2):
                o2 = _split_operands(ops2)
                if (_regs(o2[1]) | _regs(o2[2])) & live:      # SrcA / SrcB: not interlocked
                    bad.append((f'{addr:x}: {mn} {ops}', f'{func[j][0]:x}: {mn2} {ops2}', ws))
                    continue
                live = live - _regs(o2[0]) if _regs(o2[0]) != set(live0) else live    # in place: same hazard class
                # the next MFMA cannot start before the passes of this one are through the matrix core
                step = max(passes_of(mn) - ws, 0) + 1
            elif mn2.startswith(('v_', 'ds_', 'global_', 'buffer_', 'flat_', 'scratch_')):
                src, dst = _sources_and_dests(mn2, ops2)
                if src & live:
                    bad.append((f'{addr:x}: {mn} {ops}', f'{func[j][0]:x}: {mn2} {ops2}', ws))
                    continue
                live = live - dst
            for k in succ[j]:
                todo.append((k, ws + step, live))
    return count, bad


VALU_TO_MFMA_WAIT_STATES = 2


def audit_valu_feeds(func):
    """-> [(valu line, mfma line, wait states)]: MFMAs that read a VALU result too early."""
    succ = _successors(func)
    bad = []
    for i, (addr, mn, ops) in enumerate(func):
        if not mn.startswith('v_') or _window(mn) or mn.startswith(('v_cmp', 'v_readlane', 'v_readfirstlane')):
            continue
        o = _split_operands(ops)
        dst = frozenset(r for r in _regs(o[0]) if r[0] in 'va') if o else frozenset()
        if not dst:
            continue
        todo = [(j, 0) for j in succ[i]]
        seen = set()
        while todo:
            j, ws = todo.pop()
            if ws >= VALU_TO_MFMA_WAIT_STATES or (j, ws) in seen:
                continue
            seen.add((j, ws))
            _, mn2, ops2 = func[j]
            step = 1
            if mn2 == 's_nop':
                step = int(ops2.split()[0]) + 1
            elif _window(mn2):
                o2 = _split_operands(ops2)
                if (_regs(o2[1]) | _regs(o2[2]) | _regs(o2[3] if len(o2) > 3 else '')) & dst:
                    bad.append((f'{addr:x}: {mn} {ops}', f'{func[j][0]:x}: {mn2} {ops2}', ws))
                continue            # (an MFMA in between is far more than two wait states)
            for k in succ[j]:
                todo.append((k, ws + step))
    return bad


def warn_pk_add_into_cvt_f64(func, name=''):
    """WARNING class (round 6, never a failure): a packed fp32 add / mul (v_pk_add_f32, v_pk_mul_f32, v_pk_fma_f32) whose
    result is converted to fp64 (v_cvt_f64_f32) within the next 2 instructions, in a kernel WITHOUT MFMAs.  This is the
    sequence the round-5 diagnosis of the two-rank `exp_avg_sq` desynchronisation ended at (profiles/r5_two_rank_sync.txt,
    section 5: the corrupted 16-lane slots of the deleted row-per-thread Adam launches coincided with it; the same
    sequence ships in adam_pack_kernel and is clean there in 89 two-rank runs).  No cause was established - round 6
    only added that v_pk_add_f32 shares a pipe with the matrix core (it does not overlap with MFMAs of ANY wave of the
    SIMD, profiles/r6_coexec_bf16.txt) - so a build that lands on the pattern is listed, for whoever sees rank drift
    (the agent's per-epoch `multi_gpu_param_check` is the run-time guard).  -> ['kernel: address  pk-op -> cvt']"""
    out = []
    if any(mn.startswith('v_mfma') for _, mn, _ in func):
        return out
    for i, (addr, mn, ops) in enumerate(func):
        if not (mn.startswith('v_pk_add_f32') or mn.startswith('v_pk_mul_f32') or mn.startswith('v_pk_fma_f32')):
            continue
        o = _split_operands(ops)
        dst = set(_regs(o[0])) if o else set()
        for j in range(i + 1, min(i + 3, len(func))):
            mn2, ops2 = func[j][1], func[j][2]
            if mn2.startswith('v_cvt_f64_f32'):
                o2 = _split_operands(ops2)
                if len(o2) > 1 and dst & set(_regs(o2[1])):
                    out.append(f'{name}: {addr:#x}  {mn} {ops}  ->  {mn2} {ops2}')
    return out


def audit(lib_path, warnings=None):
    """-> (number of MFMA instructions, [offending 'mfma -> reader (wait states)' lines]); `warnings` (a list) receives the
    lines of the warning classes."""
    work = tempfile.mkdtemp(prefix='rlg_audit_')
    try:
        local = os.path.join(work, os.path.basename(lib_path))
        shutil.copy(lib_path, local)
        subprocess.run([OBJDUMP, '--offloading', local], check=True, capture_output=True, cwd=work)
        objs = [os.path.join(work, f) for f in os.listdir(work) if 'amdgcn' in f]
        if not objs:
            raise RuntimeError('no device code object found in ' + lib_path)
        count, bad = 0, []
        for co in sorted(objs):
            text = subprocess.run([OBJDUMP, '-d', co], check=True, capture_output=True, text=True).stdout
            names = []
            for k, func in enumerate(_parse(text, names)):
                if warnings is not None:
                    warnings += warn_pk_add_into_cvt_f64(func, names[k])
                n, b = audit_function(func)
                count += n
                bad += [f'{m}  ->  {r}   ({ws} wait states)' for m, r, ws in b]
                bad += [f'VALU result read by an MFMA: {v}  ->  {r}   ({ws} wait states)' for v, r, ws in audit_valu_feeds(func)]
        return count, bad
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == '__main__':
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'rl_games_amd', 'librlg_hip.so')
    warns = []
    n, bad = audit(path, warns)
    print(f'{path}: {n} MFMA instructions, {len(bad)} reads inside a hazard window (MFMA result -> reader, VALU result -> MFMA)')
    for b in bad[:60]:
        print('   ', b)
    if warns:
        print(f'warning: {len(warns)} packed fp32 result(s) converted to fp64 right behind (no MFMA in the kernel) - see '
              'warn_pk_add_into_cvt_f64:')
        for w in warns[:20]:
            print('   ', w)
    sys.exit(1 if bad else 0)
