"""Loops of a disassembled kernel (tools/disasm_kernel.sh output): per backward branch the instruction mix of its body.
    python tools/loop_stats.py /tmp/k.s [--show N]   (N: print the body of the loop with the N-th most MFMAs)"""
import re, sys
lines = open(sys.argv[1]).read().splitlines()
show = int(sys.argv[sys.argv.index('--show') + 1]) if '--show' in sys.argv else None
ins = []
for ln in lines:
    m = re.match(r'\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):', ln)
    if m:
        ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
addr_index = {a: i for i, (a, _, _) in enumerate(ins)}
loops = []
for i, (a, op, args) in enumerate(ins):
    if op.startswith('s_cbranch') or op == 's_branch':
        off = int(args.split()[0])
        if off >= 32768:
            off -= 65536
        tgt = a + 4 + 4 * off
        if tgt <= a and tgt in addr_index:
            body = ins[addr_index[tgt]:i + 1]
            kinds = {}
            for _, o, _ in body:
                k = ('mfma' if 'mfma' in o else 'vmem' if o.startswith(('buffer_', 'global_', 'flat_', 'scratch_')) else
                     'lds' if o.startswith('ds_') else 'salu' if o.startswith('s_') else 'valu')
                kinds[k] = kinds.get(k, 0) + 1
            waits = [ar for _, o, ar in body if o == 's_waitcnt']
            nops = sum(int(ar.split()[0]) + 1 for _, o, ar in body if o == 's_nop')
            lanes = sum(1 for _, o, _ in body if o in ('v_readlane_b32', 'v_writelane_b32'))
            loops.append((kinds.get('mfma', 0), len(body), kinds, waits, nops, lanes, addr_index[tgt], i))
loops.sort(key=lambda t: -t[0])
for n, (mf, ln, kinds, waits, nops, lanes, b, e) in enumerate(loops[:12]):
    print(f'loop {n}: {ln} instrs {kinds} nop-states {nops} lane-moves {lanes} waitcnts {len(waits)}: {waits[:12]}')
if show is not None and show < len(loops):
    _, _, _, _, _, _, b, e = loops[show]
    for a, o, ar in ins[b:e + 1]:
        print(f'  {o} {ar}')
