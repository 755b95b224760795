timeout 600 python tools/soak.py 150 2>&1 | tail -20
