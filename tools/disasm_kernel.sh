#!/bin/bash
# disassembly of one kernel of a built object: tools/disasm_kernel.sh mlp_chain bwd_pipe > /tmp/k.s
OBJ=/root/repo/rl_games_amd/csrc/build/$1.o
T=/tmp/_rlg_dis; mkdir -p $T; rm -f $T/*
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $OBJ
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn $T/dev.co | awk -v pat="$2" '
  /^[0-9a-f]+ <.*>:$/ { on = (index($0, pat) > 0) }
  on { print }'
