"""Per-kernel register / LDS / scratch usage of the built library (AMDGPU code-object metadata).
    python tools/kernel_resources.py [pattern] [lib]"""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1] if len(sys.argv) > 1 else ''
lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, 'rl_games_amd', 'librlg_hip.so')
tmp = '/tmp/_rlg_co'
os.makedirs(tmp, exist_ok=True)
import glob
objs = sorted(glob.glob(os.path.join(ROOT, 'rl_games_amd', 'csrc', 'build', '*.o'))) if len(sys.argv) <= 2 else [lib]
out = ''
for o in objs:
    fat, co = f'{tmp}/fat.bin', f'{tmp}/dev.co'
    for f in (fat, co):
        if os.path.exists(f):
            os.remove(f)
    if subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-objcopy', '--dump-section', f'.hip_fatbin={fat}', o],
                      capture_output=True).returncode != 0:
        continue                       # (an object without device code)
    subprocess.run(['/opt/rocm/lib/llvm/bin/clang-offload-bundler', '--unbundle', '--type=o', f'--input={fat}',
                    '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', f'--output={co}'], check=True)
    out += subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-readelf', '--notes', co], capture_output=True, text=True).stdout
cur = {}
rows = []
for line in out.splitlines():
    m = re.match(r'\s+\.(\w+):\s+(.*)', line) or re.match(r'\s+- \.(\w+):\s+(.*)', line)
    if not m:
        continue
    k, v = m.group(1), m.group(2).strip()
    if k == 'agpr_count' and cur.get('name'):
        pass
    if k in ('agpr_count', 'vgpr_count', 'sgpr_count', 'private_segment_fixed_size', 'group_segment_fixed_size',
             'vgpr_spill_count', 'sgpr_spill_count', 'name', 'symbol'):
        cur[k] = v
    if k == 'wavefront_size':
        rows.append(cur); cur = {}
for r in rows:
    n = r.get('name', '?')
    d = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()
    d = re.sub(r'\(.*', '', d)
    if pat and pat not in d:
        continue
    print(f"{d[:70]:70s} vgpr {r.get('vgpr_count','?'):>4} agpr {r.get('agpr_count','?'):>4} sgpr {r.get('sgpr_count','?'):>4} "
          f"scratch {r.get('private_segment_fixed_size','?'):>5} lds {r.get('group_segment_fixed_size','?'):>6} "
          f"vspill {r.get('vgpr_spill_count','0')} sspill {r.get('sgpr_spill_count','0')}")
