"""CPU tier: the packaged reference (oracle/make_ref.py -> oracle/_ref/, used by bench.py's `cpu_baseline` kind "reference" and
`reference_gpu_baseline` legs) is byte-identical to /root/reference where that exists, and oracle/run_ref.py runs it end to end."""
import hashlib
import json
import os
import subprocess
import sys
import zipfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, 'oracle', '_ref')


def _have():
    sys.path.insert(0, ROOT)
    from oracle import make_ref
    return make_ref.make()


def test_packaged_reference_is_byte_identical_and_runs():
    if not _have():
        pytest.skip('neither oracle/_ref/ nor /root/reference is present')
    man = json.load(open(os.path.join(REF, 'MANIFEST.json')))
    with zipfile.ZipFile(os.path.join(REF, 'nero_ref.zip')) as z:
        names = set(z.namelist())
        for name, h in man['files'].items():
            if name.startswith('assets/'):
                data = open(os.path.join(REF, name), 'rb').read()
            else:
                assert name in names
                data = z.read(name)
            assert hashlib.sha256(data).hexdigest() == h, name
            src = os.path.join('/root/reference', name)
            if os.path.exists(src):                               # the build container: what travels equals what lies there
                assert hashlib.sha256(open(src, 'rb').read()).hexdigest() == h, name
    assert {'network/renderer.py', 'network/field.py', 'utils/ref_utils.py', 'utils/raw_utils.py'} <= set(man['files'])
    env = dict(os.environ, NERO_REFERENCE_ROOT=REF)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'oracle', 'run_ref.py'), '--device', 'cpu', '--rays', '48', '--samples', '16', '16', '8',
                        '--warmup', '0', '--steps', '1', '--threads', '4'], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    line = [ln for ln in p.stdout.splitlines() if ln.startswith('{')][-1]
    rec = json.loads(line)
    assert rec['ok'] and rec['rays_per_s'] > 0 and rec['reference_root'] == REF, (rec, p.stderr[-500:])
