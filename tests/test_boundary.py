"""CPU: the C-ABI boundary.  librlg_hip.so loads (no GPU needed to dlopen it), exports every
symbol include/rlg_hip.h declares, the Python binding table covers exactly those symbols, the
product package never touches oracle/, and it fails loudly without a GPU."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'rlg_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(?:int|long long)\s+(rlg_\w+)\s*\(', text)))


def test_header_declares_symbols():
    syms = _declared_symbols()
    assert 'rlg_gae_envmajor_fused' in syms and 'rlg_gae_strided' in syms


def test_library_exports_every_declared_symbol():
    from rl_games_amd import _lib
    lib = _lib.load()
    for name in _declared_symbols():
        assert hasattr(lib, name), f'{name} declared in include/rlg_hip.h but not exported'


def test_binding_table_matches_header():
    from rl_games_amd import _lib
    assert sorted(_lib.exported_prototypes()) == _declared_symbols()


def test_host_queries_without_gpu():
    from rl_games_amd import _lib
    lib = _lib.load()
    assert lib.rlg_gae_envmajor_supported(32) == 1
    assert lib.rlg_gae_envmajor_supported(30) == 0
    assert lib.rlg_gae_envmajor_supported(128) == 0
    assert lib.rlg_gae_envmajor_num_partials(65536) == 1024
    assert lib.rlg_gae_envmajor_num_partials(65) == 2


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'rl_games_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.hpp', '.h', '.cpp')):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f
                assert 'liboracle' not in src and 'ppo_oracle' not in src, f


def test_cpu_tensors_fail_loudly():
    from rl_games_amd.gae import compute_gae
    from rl_games_amd._lib import HipLibraryError
    x = torch.zeros(4, 2, 1)
    with pytest.raises(HipLibraryError):
        compute_gae(x, x, torch.zeros(4, 2), torch.zeros(2, 1), torch.zeros(2), 0.99, 0.95)


def test_no_mfma_result_is_read_inside_its_hazard_window():
    """Build audit (tools/audit_mfma.py): on gfx950 a VALU / LDS / store read of an MFMA result is NOT
    interlocked (10 wait states for v_mfma_f32_16x16x4_f32, 8 for v_mfma_f32_16x16x32_bf16, measured with
    tools/exp/mfma_valu_read_probe.hip) and hipcc counts one short across branches; the kernels carry an
    explicit s_nop in front of their epilogues.  Fails if any control-flow path of a build reads a result
    register too early."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('audit_mfma', os.path.join(ROOT, 'tools', 'audit_mfma.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not os.path.exists(mod.OBJDUMP):
        pytest.skip('llvm-objdump not available')
    from rl_games_amd import _lib
    count, bad = mod.audit(_lib.LIB_PATH)
    assert count > 1000, count            # the fused MLP kernels are in there
    assert not bad, bad[:5]
