"""CPU tier (hipcc cross-compiles without a GPU): no packed fp32 VALU arithmetic in any kernel of the library.

Round 5 traced a run-to-run difference of the SDF gradients to `v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32`: on gfx950 a wave executing them
while another wave of the same SIMD executes MFMAs occasionally gets a quarter-wave of results a few ulp off (nero_amd/csrc/common.h,
DESIGN.md 9.3).  hipcc forms them by SLP vectorisation (off: -fno-slp-vectorize in __graft_entry__.build) or from ext_vector_type arithmetic
(written component-wise in the sources).  This test compiles every translation unit with the build's own flags and scans the ISA."""
import glob
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PACKED = re.compile(r'\bv_pk_(mul|add|fma)_f32\b')


def _flags():
    src = open(os.path.join(ROOT, '__graft_entry__.py')).read()
    m = re.search(r"flags = \[([^\]]*)\]", src)
    return [f.strip().strip("'") for f in m.group(1).split(',')]


def _scan(path):
    p = subprocess.run(['hipcc'] + _flags() + ['-S', '--cuda-device-only', '-o', '-', path], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    hits, kernel = {}, None
    for line in p.stdout.splitlines():
        if line.endswith(':') and not line.startswith(('.', '\t', ' ', ';')):
            kernel = line[:-1]
        elif PACKED.search(line):
            hits[kernel] = hits.get(kernel, 0) + 1
    return os.path.basename(path), hits


def test_no_packed_fp32_instructions_in_any_kernel():
    flags = _flags()
    assert '-fno-slp-vectorize' in flags and '--offload-arch=gfx950' in flags
    units = sorted(glob.glob(os.path.join(ROOT, 'nero_amd', 'csrc', '*.hip')))
    assert len(units) >= 15
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
        res = dict(ex.map(_scan, units))
    bad = {u: h for u, h in res.items() if h}
    assert not bad, bad


def test_paired_chain_kernels_fit_two_workgroups_per_cu():
    """the two-workgroups-per-CU forward / tangent kernels (mlp_f16p.hip, the default on large launches) only pay while two of them share a
    CU: 256 threads each = two waves per SIMD = at most 256 VGPRs per wave, and a spill would put scratch traffic into their epilogues.  The
    code object's own metadata: <= 256 VGPRs, no spilled VGPR, no scratch for fwd_p_kernel and tan_p_kernel (the opt-in bwd_p_kernel is
    allowed its 14 spills: DESIGN.md section 3)."""
    path = os.path.join(ROOT, 'nero_amd', 'csrc', 'mlp_f16p.hip')
    p = subprocess.run(['hipcc'] + _flags() + ['-S', '--cuda-device-only', '-o', '-', path], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    meta = {}
    text = p.stdout[p.stdout.index('amdhsa.kernels:'):]
    for entry in re.split(r'\n  - ', text)[1:]:          # one YAML list item per kernel (the field order inside an item is alphabetical)
        nm = re.search(r'\.name:\s+(\S+)', entry)
        if nm is None:                                    # (the amdhsa.version list behind the kernels)
            continue
        name = nm.group(1)
        meta[name] = {m.group(1): int(m.group(2)) for m in
                      re.finditer(r'\.(vgpr_count|vgpr_spill_count|private_segment_fixed_size|max_flat_workgroup_size):\s+(\d+)', entry)}
    # (the 8-wave instantiations <8> -- four waves per SIMD at 128 registers, NERO_F16_PW=8 -- are a measured-and-dropped experiment of
    #  round 6 and do spill; the default organisation is <4>)
    for k in ('fwd_p_kernelILi4E', 'tan_p_kernelILi4E'):
        hit = [v for n, v in meta.items() if k in n]
        assert len(hit) == 1, (k, list(meta))
        v = hit[0]
        assert v['vgpr_count'] <= 256 and v['vgpr_spill_count'] == 0 and v['private_segment_fixed_size'] == 0 and v['max_flat_workgroup_size'] == 256, (k, v)
