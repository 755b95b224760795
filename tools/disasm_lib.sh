#!/bin/bash
# disassembly of every device code object of a built library: tools/disasm_lib.sh <lib.so> > /tmp/lib.s
W=$(mktemp -d); cp "$1" $W/lib.so
(cd $W && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so >/dev/null 2>&1)
for co in $W/*amdgcn*; do /opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn $co; done
rm -rf $W
