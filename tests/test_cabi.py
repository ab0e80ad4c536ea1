"""CPU tier: the C-ABI library loads and exports every symbol include/nero_hip.h declares (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    hdr = open(os.path.join(ROOT, 'include', 'nero_hip.h')).read()
    names = sorted(set(re.findall(r'\b(nero_[a-z0-9_]+)\s*\(', hdr)))
    assert len(names) >= 30
    lib = ctypes.CDLL(os.path.join(ROOT, 'nero_amd', 'libnero_hip.so'))
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    lib.nero_last_error.restype = ctypes.c_char_p
    assert lib.nero_version() >= 100
    assert isinstance(lib.nero_last_error(), bytes)


def test_struct_layouts_match_header():
    """ctypes mirrors of the descriptor structs must have the sizes the C compiler gives them."""
    import subprocess, tempfile
    from nero_amd import _lib as L
    src = '#include <stdio.h>\n#include "nero_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(nero_fwd_layer), sizeof(nero_fwd_chain), sizeof(nero_tan_layer), sizeof(nero_tan_chain), sizeof(nero_bwd_layer), sizeof(nero_bwd_chain), sizeof(nero_dw_job), sizeof(nero_pack_job), sizeof(nero_wn_job), sizeof(nero_adam_job));}'
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, 's.c')
        open(c, 'w').write(src)
        exe = os.path.join(td, 's')
        subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), c, '-o', exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    mine = [ctypes.sizeof(t) for t in (L.FwdLayer, L.FwdChain, L.TanLayer, L.TanChain, L.BwdLayer, L.BwdChain, L.DwJob, L.PackJob, L.WnJob, L.AdamJob)]
    assert sizes == mine, (sizes, mine)


def test_dw_workspace_is_worst_case_sized():
    """regression: the split-K slice count is not monotonic in the row count, so the workspace query must not depend on it"""
    lib = ctypes.CDLL(os.path.join(ROOT, 'nero_amd', 'libnero_hip.so'))
    base = 256 * (256 * 256 + 256)
    for n in (0, 1, 100, 8000, 8192, 8193, 300000, 1 << 22):
        assert lib.nero_dw_workspace_floats(n) >= base
    assert lib.nero_dw_workspace_floats(1 << 22) >= ((1 << 22) // 128) * 1028


def test_f16_paired_selector_round_trip():
    """nero_f16_paired (no GPU needed: a host-side selection): a negative mask only queries, a mask is kept modulo its four bits, the call
    returns the previous selection; the default is forward + tangent (3) unless NERO_F16_PAIRED says otherwise"""
    from nero_amd import chain as CH
    prev = CH.f16_paired()
    try:
        if 'NERO_F16_PAIRED' not in os.environ:
            assert prev == 3
        assert CH.f16_paired(8 | 5) == prev
        assert CH.f16_paired() == 13
        assert CH.f16_paired(0x47) == 13              # only bits 0-3 are kept
        assert CH.f16_paired(-1) == 7
    finally:
        CH.f16_paired(prev)
    assert CH.f16_paired() == prev
