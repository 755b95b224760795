timeout 600 python tools/exp/fuzz_chain.py 80 1 2>&1 | grep "BAD\|bad of\|Error\|error" | head -30
