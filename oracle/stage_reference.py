"""TEST / BASELINE INFRASTRUCTURE (never imported by rl_games_amd): stages the UNTOUCHED reference package for the
GPU box, where /root/reference does not exist.

    python oracle/stage_reference.py            (run by __graft_entry__.build() wherever /root/reference is present)

Writes oracle/_ref/rl_games_ref.zip = the reference's own `rl_games/**/*.py`, byte for byte, as one importable
archive (zipimport), plus oracle/_ref/MANIFEST.json (file list with sha256, the reference path it was taken from).
oracle/_ref/ is git-ignored: no reference source enters this repository's history; the archive travels to the GPU
box with the gpurun snapshot like the built .so files.  Consumers: bench.py's `cpu_baseline` leg (kind "reference":
rl_games.algos_torch.a2c_continuous.A2CAgent.train_epoch on the box's host cores, a2c_continuous.py:136-234,
torch_runner.py:217-226) through tests/golden/ref_import.enable().  The product path never touches it."""
import hashlib
import json
import os
import sys
import zipfile

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get('RLG_REFERENCE', '/root/reference')
OUT_DIR = os.path.join(HERE, '_ref')
ARCHIVE = os.path.join(OUT_DIR, 'rl_games_ref.zip')


def stage(verbose=True):
    src = os.path.join(REFERENCE, 'rl_games')
    if not os.path.isdir(src):
        if verbose:
            print(f'stage_reference: {src} not present - nothing staged')
        return None
    os.makedirs(OUT_DIR, exist_ok=True)
    manifest = {'source': src, 'files': {}}
    tmp = ARCHIVE + '.tmp'
    with zipfile.ZipFile(tmp, 'w', zipfile.ZIP_DEFLATED) as z:
        for root, dirs, files in os.walk(src):
            dirs[:] = sorted(d for d in dirs if d != '__pycache__')
            for name in sorted(files):
                if not name.endswith('.py'):
                    continue
                path = os.path.join(root, name)
                rel = os.path.relpath(path, REFERENCE)
                data = open(path, 'rb').read()
                # fixed timestamps: the archive is reproducible
                info = zipfile.ZipInfo(rel, date_time=(2020, 1, 1, 0, 0, 0))
                info.compress_type = zipfile.ZIP_DEFLATED
                z.writestr(info, data)
                manifest['files'][rel] = hashlib.sha256(data).hexdigest()
    os.replace(tmp, ARCHIVE)
    with open(os.path.join(OUT_DIR, 'MANIFEST.json'), 'w') as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    if verbose:
        print(f'stage_reference: {len(manifest["files"])} files of {src} -> {ARCHIVE}')
    return ARCHIVE


if __name__ == '__main__':
    sys.exit(0 if stage() or not os.path.isdir(os.path.join(REFERENCE, 'rl_games')) else 1)
