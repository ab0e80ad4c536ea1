"""Assemble oracle/_ref/ -- the UNMODIFIED reference, packaged so that it can travel to the GPU box and be TIMED there.

TEST / MEASUREMENT INFRASTRUCTURE ONLY.  Nothing in nero_amd/ imports, opens or executes anything under oracle/.  oracle/_ref/ is
git-ignored (the history never holds reference sources) but not gpurun-ignored: like the built .so files it rides along with the
snapshot, so that bench.py's `cpu_baseline` (kind "reference") and `reference_gpu_baseline` legs can run the reference's own
NeROShapeRenderer.render + loss + backward (+ torch.optim.Adam) on the GPU box's host cores and on its MI355X
(network/renderer.py:608-627, train/trainer.py:127-140).  /root/reference does not exist on the GPU box.

    python oracle/make_ref.py            (build container only: needs /root/reference; __graft_entry__.build() calls it when it can)

Output (all under oracle/_ref/):
    nero_ref.zip            network/*.py, utils/*.py and dataset/*.py (imported by network/renderer.py:10) of the reference, byte for byte, as an importable zip archive (zipimport) --
                            a packaged artefact like a compiled library, not a source tree -- plus empty __init__.py markers
    assets/bsdf_256_256.bin the FG table the reference loads by relative path (network/field.py:510)
    MANIFEST.json           sha256 of every packaged file and of its /root/reference original (they must agree: unmodified)
"""
import hashlib
import json
import os
import shutil
import sys
import zipfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '_ref')
REF = os.environ.get('NERO_REFERENCE_ROOT', '/root/reference')
PACKAGES = ('network', 'utils', 'dataset')
ASSETS = ('assets/bsdf_256_256.bin',)


def _sha(b):
    return hashlib.sha256(b).hexdigest()


def available():
    return os.path.isdir(os.path.join(REF, 'network'))


def built():
    return os.path.exists(os.path.join(OUT, 'nero_ref.zip')) and os.path.exists(os.path.join(OUT, ASSETS[0]))


def make(force=False):
    """-> True when oracle/_ref/ is in place (freshly assembled or already there), False when the reference tree is absent."""
    if built() and not force:
        return True
    if not available():
        return False
    os.makedirs(OUT, exist_ok=True)
    manifest = {'reference_root': REF, 'files': {}}
    zpath = os.path.join(OUT, 'nero_ref.zip')
    with zipfile.ZipFile(zpath, 'w', zipfile.ZIP_DEFLATED) as z:
        for pkg in PACKAGES:
            names = sorted(n for n in os.listdir(os.path.join(REF, pkg)) if n.endswith('.py'))
            if '__init__.py' not in names:
                z.writestr(f'{pkg}/__init__.py', '')
            for n in names:
                data = open(os.path.join(REF, pkg, n), 'rb').read()
                z.writestr(f'{pkg}/{n}', data)
                manifest['files'][f'{pkg}/{n}'] = _sha(data)
    with zipfile.ZipFile(zpath) as z:                      # read back: what travels equals what lies under /root/reference
        for name, h in manifest['files'].items():
            assert _sha(z.read(name)) == h == _sha(open(os.path.join(REF, name), 'rb').read()), name
    for a in ASSETS:
        dst = os.path.join(OUT, a)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(os.path.join(REF, a), dst)
        manifest['files'][a] = _sha(open(dst, 'rb').read())
    with open(os.path.join(OUT, 'MANIFEST.json'), 'w') as f:
        json.dump(manifest, f, indent=1)
    return True


if __name__ == '__main__':
    ok = make(force='--force' in sys.argv)
    print('oracle/_ref ready' if ok else f'{REF} not present: nothing assembled')
