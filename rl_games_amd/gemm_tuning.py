"""hipBLASLt / rocBLAS solution selection for the fp32 GEMMs that still go through torch.

Since round 2 the MLP and LSTM configurations run every product in this library's own MFMA kernels; library GEMMs are
left only on the fall-back paths (the per-layer engine of `fused_mlp: False`, policy shapes outside the engines'
envelope, the discrete agent's autograd trunks, a central value network's first layer over a state width that is no
multiple of 4).  The libraries' default heuristics pick poor kernels for this path's skinny shapes (e.g. dW = dZ^T A with K = 32,768 and
a 400x200 output ran at ~36 TFLOP/s); PyTorch's TunableOp facility benchmarks the libraries'
own solutions per shape and remembers the best one.  `tuning/tunableop_gfx950.csv` holds the
selections for the BASELINE.json shapes on MI355X (this image's library versions - the file's
validator rows are checked by TunableOp when it is read, a mismatch falls back to the default
heuristics).  Nothing here changes numerics class: same libraries, fp32 in / fp32 accumulate.
"""
import os

import torch

TUNED_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tuning', 'tunableop_gfx950.csv')
_state = {'enabled': False}


def enable(tuned_file=TUNED_FILE, allow_tuning=False, max_tuning_ms=30):
    """Turn TunableOp on with the shipped selections.  allow_tuning=True additionally tunes
    shapes that are not in the file the first time they are seen (seconds per new shape)."""
    if _state['enabled'] or not torch.cuda.is_available():
        return _state['enabled']
    if os.environ.get('RLG_NO_GEMM_TUNING', '0') not in ('0', ''):
        return False
    import torch.cuda.tunable as tunable
    tunable.enable(True)
    tunable.set_max_tuning_duration(max_tuning_ms)
    if tuned_file and os.path.exists(tuned_file):
        try:
            tunable.read_file(tuned_file)
        except Exception as e:  # pragma: no cover
            print(f'rl_games_amd: could not read {tuned_file}: {e}')
    tunable.tuning_enable(bool(allow_tuning))
    _state['enabled'] = True
    return True


def write_results(path):
    """Dump the selections made so far in TunableOp's CSV format."""
    import torch.cuda.tunable as tunable
    with open(path, 'w') as f:
        for v in tunable.get_validators():
            f.write('Validator,' + ','.join(str(x) for x in v) + '\n')
        for r in tunable.get_results():
            f.write(','.join(str(x) for x in r) + '\n')
