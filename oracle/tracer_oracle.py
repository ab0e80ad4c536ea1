"""Brute-force closest-hit ray/mesh oracle (Moeller-Trumbore over every triangle, float64).  TEST INFRASTRUCTURE ONLY.
The reference's tracer is the un-vendored third-party `_raytracing` CUDA extension (no source, no pin, no tests under
/root/reference): parity of the HIP BVH is therefore pinned against this restatement of the contract NeRO relies on
(raytracing/raytracer.py:21-54; network/renderer.py:719-729): closest hit with t > 0, geometric face normal from the vertex
winding, depth >= 10 <=> miss.  'parity unpinned' w.r.t. the third-party binary itself."""
import os

import numpy as np


def trace_bruteforce(verts, tris, o, d, miss_depth=10.0):
    v0 = verts[tris[:, 0]].astype(np.float64)
    e1 = verts[tris[:, 1]].astype(np.float64) - v0
    e2 = verts[tris[:, 2]].astype(np.float64) - v0
    o = o.astype(np.float64)
    d = d.astype(np.float64)
    n = o.shape[0]
    depth = np.full(n, miss_depth)
    tri = np.full(n, -1, np.int64)
    for r in range(n):
        p = np.cross(d[r], e2)
        det = np.einsum('ij,ij->i', e1, p)
        ok = np.abs(det) > 1e-20
        inv = np.where(ok, 1.0 / np.where(ok, det, 1.0), 0.0)
        tv = o[r] - v0
        u = np.einsum('ij,ij->i', tv, p) * inv
        q = np.cross(tv, e1)
        v = (q @ d[r]) * inv
        t = np.einsum('ij,ij->i', e2, q) * inv
        hit = ok & (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (t > 0) & (t < miss_depth)
        if hit.any():
            tt = np.where(hit, t, np.inf)
            k = int(np.argmin(tt))
            depth[r], tri[r] = tt[k], k
    nrm = np.zeros((n, 3))
    h = tri >= 0
    nn = np.cross(e1[tri[h]], e2[tri[h]])
    nrm[h] = nn / np.linalg.norm(nn, axis=1, keepdims=True)
    pos = o + depth[:, None] * d
    return pos, nrm, depth, tri


_C_SRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc', 'tracer_oracle.c')
_C_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_build', 'libtracer_oracle.so')


def build_c(force=False):
    """gcc-compile oracle/csrc/tracer_oracle.c (the C restatement of trace_bruteforce) into oracle/_build/.  -> library path"""
    import subprocess
    if force or not os.path.exists(_C_LIB) or os.path.getmtime(_C_LIB) < os.path.getmtime(_C_SRC):
        os.makedirs(os.path.dirname(_C_LIB), exist_ok=True)
        subprocess.check_call(['gcc', '-O2', '-fopenmp', '-shared', '-fPIC', '-o', _C_LIB, _C_SRC, '-lm'])
    return _C_LIB


def trace_bruteforce_margins(verts, tris, o, d, miss_depth=10.0, eps_edge=2e-5, eps_t=2e-6):
    """trace_bruteforce through its C restatement (oracle/csrc/tracer_oracle.c: identical float64 arithmetic and predicates,
    OpenMP over rays; tests/test_tracer_oracle_c.py pins it to the numpy version above), plus a per-ray AMBIGUITY flag: True
    when the fp64 answer sits within `eps_edge` (barycentric units) of a triangle edge that could change the closest hit, or a
    candidate intersection lies within `eps_t` of the ray origin (Stage-II secondary rays start 1e-5 off the surface along the
    direction, network/field.py:859: the triangle they left sits at t = -1e-5 cos(theta), i.e. at ~0 for grazing directions).
    The defaults are ~20x the float32 rounding of Moeller-Trumbore at this scene scale (coordinates ~0.5, ulp 6e-8): a float32
    tracer may legitimately answer differently on exactly those rays and on no others.
    -> pos [n,3], nrm [n,3], depth [n], tri [n], ambiguous [n] bool"""
    import ctypes as C
    lib = C.CDLL(build_c())
    V = np.ascontiguousarray(verts, dtype=np.float32)
    F = np.ascontiguousarray(tris, dtype=np.int32)
    O = np.ascontiguousarray(o, dtype=np.float64)
    D = np.ascontiguousarray(d, dtype=np.float64)
    n = O.shape[0]
    pos, nrm, depth = np.empty((n, 3)), np.empty((n, 3)), np.empty(n)
    tri, amb = np.empty(n, np.int64), np.zeros(n, np.uint8)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.tracer_oracle_trace.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double,
                                        C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = lib.tracer_oracle_trace(P(V), V.shape[0], P(F), F.shape[0], P(O), P(D), n, miss_depth, eps_edge, eps_t, P(pos), P(nrm), P(depth),
                                 P(tri), P(amb))
    assert rc == 0
    return pos, nrm, depth, tri, amb.astype(bool)
