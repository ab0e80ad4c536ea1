"""Brute-force closest-hit ray/mesh oracle (Moeller-Trumbore over every triangle, float64).  TEST INFRASTRUCTURE ONLY.
The reference's tracer is the un-vendored third-party `_raytracing` CUDA extension (no source, no pin, no tests under
/root/reference): parity of the HIP BVH is therefore pinned against this restatement of the contract NeRO relies on
(raytracing/raytracer.py:21-54; network/renderer.py:719-729): closest hit with t > 0, geometric face normal from the vertex
winding, depth >= 10 <=> miss.  'parity unpinned' w.r.t. the third-party binary itself."""
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
