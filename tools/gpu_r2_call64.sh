timeout 600 python tools/exp/fuzz_chain.py 250 7 2>&1 | grep -c ": ok"
timeout 600 python tools/exp/fuzz_chain.py 250 7 2>&1 | grep "BAD\|bad of\|Error\|error" | head -30
