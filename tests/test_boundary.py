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
                # ... nor the reference package itself or the test helper that makes it importable in the build container
                assert 'rl_games_ref' not in src and 'ref_import' not in src and 'oracle/_ref' not in src, f
                assert not re.search(r'^\s*(from|import)\s+rl_games(\.|\s|$)', src, flags=re.M), f


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


def _audit_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location('audit_mfma', os.path.join(ROOT, 'tools', 'audit_mfma.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _listing(instrs):
    """[(mnemonic, operands)] -> the (address, mnemonic, operands) form of audit_function, 4 bytes apart."""
    return [(0x100 + 4 * i, mn, ops) for i, (mn, ops) in enumerate(instrs)]


def test_mfma_audit_catches_the_hazards_it_is_meant_to():
    """The audit itself, on hand-written listings: the sequence that produced wrong results on gfx950 (a
    result read 2 wait states behind the MFMA, reached through a taken branch), its straight-line form, a
    read as SrcA of another MFMA - and the forms that are fine (10 wait states, SrcC chains, overwritten
    results, the 8-wait-state window of the bf16 shape)."""
    mod = _audit_module()
    mfma = ('v_mfma_f32_16x16x4_f32', 'v[16:19], v7, v17, v[16:19]')
    read = ('v_pk_add_f32', 'v[0:1], v[14:15], v[18:19]')
    # straight line, too early / late enough
    n, bad = mod.audit_function(_listing([mfma, ('s_nop', '7'), read, ('s_endpgm', '')]))
    assert n == 1 and len(bad) == 1 and bad[0][2] == 8
    n, bad = mod.audit_function(_listing([mfma, ('s_nop', '9'), read, ('s_endpgm', '')]))
    assert n == 1 and not bad
    # the shape of the real defect: MFMA, taken branch over a block that has its own wait states, early read
    prog = [mfma, ('s_cbranch_execnz', '3'), ('s_nop', '15'), ('v_mov_b32_e32', 'v40, v16'), ('s_endpgm', ''),
            ('s_waitcnt', 'vmcnt(0)'), read, ('s_endpgm', '')]
    n, bad = mod.audit_function(_listing(prog))
    assert [b[2] for b in bad] == [2], bad                         # only the path through the branch is short
    # SrcC of a following MFMA is interlocked; SrcA / SrcB are not
    chain = [mfma, ('v_mfma_f32_16x16x4_f32', 'v[20:23], v8, v9, v[16:19]'), ('s_nop', '15'), ('s_nop', '15'), ('s_endpgm', '')]
    assert not mod.audit_function(_listing(chain))[1]
    as_a = [mfma, ('v_mfma_f32_16x16x4_f32', 'v[20:23], v16, v9, v[20:23]'), ('s_nop', '15'), ('s_nop', '15'), ('s_endpgm', '')]
    assert len(mod.audit_function(_listing(as_a))[1]) == 1
    # a result register that is overwritten first is no longer the MFMA's
    over = [mfma, ('v_mov_b32_e32', 'v18, v1'), ('v_mov_b32_e32', 'v19, v1'), read, ('s_nop', '15'), ('s_endpgm', '')]
    assert not mod.audit_function(_listing(over))[1]
    # stores read their data operand; AGPR results are read by v_accvgpr_read
    st = [mfma, ('s_nop', '3'), ('global_store_dwordx4', 'v[2:3], v[16:19], off'), ('s_endpgm', '')]
    assert len(mod.audit_function(_listing(st))[1]) == 1
    agpr = [('v_mfma_f32_16x16x4_f32', 'a[0:3], v7, v17, a[0:3]'), ('s_nop', '8'), ('v_accvgpr_read_b32', 'v5, a3'), ('s_endpgm', '')]
    assert [b[2] for b in mod.audit_function(_listing(agpr))[1]] == [9]
    # bf16 16x16x32: 8 wait states
    bf = ('v_mfma_f32_16x16x32_bf16', 'v[16:19], v[4:7], v[8:11], v[16:19]')
    assert len(mod.audit_function(_listing([bf, ('s_nop', '6'), read, ('s_endpgm', '')]))[1]) == 1
    assert not mod.audit_function(_listing([bf, ('s_nop', '7'), read, ('s_endpgm', '')]))[1]
    # an MFMA in between cannot start before the passes of the first one have left the matrix core
    between = [mfma, ('v_mfma_f32_16x16x4_f32', 'v[24:27], v8, v9, v[24:27]'), ('s_nop', '0'), read, ('s_nop', '15'), ('s_nop', '15'), ('s_endpgm', '')]
    assert not [b for b in mod.audit_function(_listing(between))[1] if b[0].startswith('100:')]


def test_mfma_audit_flags_a_valu_result_read_by_an_mfma_too_early():
    """The second guard of the audit: a VGPR written by a VALU instruction needs 2 wait states before an MFMA reads
    it as SrcA / SrcB / SrcC.  The defect it is there for: `v_cvt_pk_bf16_f32` inside an asm statement (invisible to
    hipcc's hazard recogniser) one wait state in front of the MFMA that consumed the plane (rounds 2 - 3)."""
    mod = _audit_module()
    cvt = ('v_cvt_pk_bf16_f32', 'v43, v58, v43')
    bf = ('v_mfma_f32_16x16x32_bf16', 'v[2:5], v[28:31], v[40:43], v[2:5]')
    tail = [('s_nop', '15'), ('s_endpgm', '')]
    assert [b[2] for b in mod.audit_valu_feeds(_listing([cvt, bf] + tail))] == [0]
    assert [b[2] for b in mod.audit_valu_feeds(_listing([cvt, ('s_nop', '0'), bf] + tail))] == [1]
    assert [b[2] for b in mod.audit_valu_feeds(_listing([cvt, ('v_and_b32_e32', 'v20, 0xffff0000, v39'), bf] + tail))] == [1]
    assert not mod.audit_valu_feeds(_listing([cvt, ('s_nop', '1'), bf] + tail))
    assert not mod.audit_valu_feeds(_listing([cvt, ('v_mov_b32_e32', 'v1, v0'), ('v_mov_b32_e32', 'v6, v0'), bf] + tail))
    # SrcC counts; a VALU result nobody multiplies with does not; through a taken branch the distance is the path's
    as_c = ('v_mfma_f32_16x16x32_bf16', 'v[40:43], v[28:31], v[32:35], v[40:43]')
    assert len(mod.audit_valu_feeds(_listing([cvt, as_c] + tail))) == 1
    assert not mod.audit_valu_feeds(_listing([('v_mov_b32_e32', 'v90, v0'), bf] + tail))
    prog = [cvt, ('s_cbranch_execnz', '3'), ('s_nop', '3'), bf, ('s_endpgm', ''), bf] + tail
    assert sorted(b[2] for b in mod.audit_valu_feeds(_listing(prog))) == [1]


def test_dw_plan_defaults_follow_the_product_form():
    """rlg_mlp_dw_plan is host code: the default workgroup target (target_blocks <= 0) gives the split-bf16
    form (default) 16 / 16 / 32 / 64 K-slices for the BASELINE MLP at 32,768 rows and the exact-f32 form
    (RLG_DW_BF16=0, read once per process) 64 each; a rank's 4,096-row minibatch never gets fewer than two
    batches per wave."""
    import ctypes
    import subprocess
    import sys
    code = ('import ctypes, json; from rl_games_amd import _lib; lib = _lib.load(); out = {}\n'
            'for rows in (32768, 4096, 64):\n'
            '    ks = []\n'
            '    for No, Mi in ((400, 108), (200, 400), (100, 200), (22, 100)):\n'
            '        p = (ctypes.c_int * 4)(); need = lib.rlg_mlp_dw_plan(rows, No, Mi, 0, p)\n'
            '        assert need == p[3] * No * Mi, (need, list(p))\n'
            '        ks.append(p[3])\n'
            '    out[rows] = ks\n'
            'print(json.dumps(out))')
    import json
    got = {}
    for form in ('1', '0'):
        res = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, cwd=ROOT, timeout=120,
                             env=dict(os.environ, RLG_DW_BF16=form))
        assert res.returncode == 0, res.stderr[-1500:]
        got[form] = {int(k): v for k, v in json.loads(res.stdout.strip().splitlines()[-1]).items()}
    assert got['1'][32768] == [16, 16, 32, 64] and got['0'][32768] == [64, 64, 64, 64], got
    for form, batch in (('1', 8), ('0', 4)):
        for rows, ks in got[form].items():
            steps = (rows + 3) // 4
            assert all(k >= 1 and (k == 1 or steps // (k * 4) >= batch) for k in ks), (form, rows, ks)
    from rl_games_amd import _lib
    p = (ctypes.c_int * 4)()
    assert _lib.load().rlg_mlp_dw_plan(64, 10, 7, 0, p) == -1        # 70 elements: not a multiple of 4
