"""Static audit of the MFMA instructions in the built library (runs anywhere, no GPU).

    python tools/audit_mfma.py [rl_games_amd/librlg_hip.so]

hipcc (ROCm 7.2) may allocate the destination of a VGPR-form v_mfma_f32_16x16x4_f32 so that it
PARTIALLY overlaps its own SrcC, or contains its SrcA / SrcB register (LLVM treats that as legal for
128-bit results).  On gfx950 this produced wrong halves of the result fragment (found with the 8-wave
instances of csrc/mlp_chain.hip: v_mfma v[16:19], v7, v17, v[18:21]).  The kernels pin their accumulators
to AGPRs (asm volatile("" : "+a"(acc))), which keeps every MFMA in place; this audit fails if a build
contains the pattern again.  Exit status 1 and a listing when it finds one."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
PAT = re.compile(r'(v_mfma\S*)\s+([av])\[(\d+):(\d+)\],\s*([^,\s]+),\s*([^,\s]+),\s*([av])\[(\d+):(\d+)\]')


def _regs(tok):
    m = re.match(r'([av])\[(\d+):(\d+)\]$', tok) or re.match(r'([av])(\d+)$', tok)
    if not m:
        return None
    g = m.groups()
    return (g[0], int(g[1]), int(g[-1]))


def audit(lib_path):
    """-> (number of MFMA instructions, [offending disassembly lines])."""
    work = tempfile.mkdtemp(prefix='rlg_audit_')
    try:
        local = os.path.join(work, os.path.basename(lib_path))
        shutil.copy(lib_path, local)
        subprocess.run([OBJDUMP, '--offloading', local], check=True, capture_output=True, cwd=work)
        objs = [os.path.join(work, f) for f in os.listdir(work) if 'amdgcn' in f]
        if not objs:
            raise RuntimeError('no device code object found in ' + lib_path)
        count, bad = 0, []
        for co in objs:
            text = subprocess.run([OBJDUMP, '-d', co], check=True, capture_output=True, text=True).stdout
            for line in text.splitlines():
                m = PAT.search(line)
                if not m:
                    continue
                count += 1
                _, dk, d0, d1, sa, sb, ck, c0, c1 = m.groups()
                d0, d1, c0, c1 = int(d0), int(d1), int(c0), int(c1)
                partial = dk == ck and not (d1 < c0 or c1 < d0) and (d0, d1) != (c0, c1)
                inside = False
                for tok in (sa, sb):
                    r = _regs(tok)
                    inside = inside or (r is not None and r[0] == dk and not (r[2] < d0 or d1 < r[1]))
                if partial or inside:
                    bad.append(line.strip())
        return count, bad
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == '__main__':
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'rl_games_amd', 'librlg_hip.so')
    n, bad = audit(path)
    print(f'{path}: {n} MFMA instructions, {len(bad)} with a destination overlapping a source')
    for b in bad:
        print('   ', b)
    sys.exit(1 if bad else 0)
