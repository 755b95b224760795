python tools/exp/debug_loss_bwd.py 2>&1 | grep "rows"
