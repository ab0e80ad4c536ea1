"""Drop-in for the reference's `raytracing.RayTracer` (raytracing/raytracer.py:8-54): same constructor and trace() contract,
backed by the HIP BVH in libnero_hip.so instead of the un-vendored `_raytracing` CUDA extension."""
import ctypes as C

import numpy as np
import torch

from . import _lib as L


class RayTracer:
    def __init__(self, vertices, triangles):
        if torch.is_tensor(vertices):
            vertices = vertices.detach().cpu().numpy()
        if torch.is_tensor(triangles):
            triangles = triangles.detach().cpu().numpy()
        assert triangles.shape[0] > 8, "BVH needs at least 8 triangles."          # same guard as the reference wrapper (:16)
        self._v = np.ascontiguousarray(vertices, dtype=np.float32)
        self._f = np.ascontiguousarray(triangles, dtype=np.int32)
        self._h = None                       # the device BVH is built on first use (construction works without a GPU)

    def _handle(self):
        if self._h is None:
            h = C.c_void_p()
            L.check(L.lib.nero_bvh_create(self._v.ctypes.data_as(C.c_void_p), self._v.shape[0], self._f.ctypes.data_as(C.c_void_p),
                                          self._f.shape[0], C.byref(h)))
            self._h = h
        return self._h

    def __del__(self):
        try:
            if getattr(self, '_h', None):
                L.lib.nero_bvh_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def trace_grouped(self, rays_o, rays_d, group, heavy_from, inplace=False):
        """trace() with a launch-order hint (nero_bvh_trace_grouped): rays in groups of `group`, entries [heavy_from, group) of every
        group started first (Stage II: the specular directions of a surface point).  Same outputs as trace()."""
        return self.trace(rays_o, rays_d, inplace, _order=(int(group), int(heavy_from)))

    def trace_masked(self, rays_o, rays_d, skip, inplace=False, chunk_order=None):
        """trace() that does not traverse the rays flagged in `skip` (uint8 tensor [n] or a device pointer, or None; nero_bvh_trace_masked): they
        are reported as misses.  Every other ray: the outputs of trace() bit for bit.  chunk_order: a permutation of range(k) -- the rays come
        in groups of 64 k and the launch starts chunk chunk_order[0] of every group first, then chunk_order[1], ... (nero_bvh_trace_ordered:
        same outputs, another launch order)."""
        return self.trace(rays_o, rays_d, inplace, _skip=skip, _chunks=chunk_order)

    def trace(self, rays_o, rays_d, inplace=False, _order=None, _skip=None, _chunks=None):
        rays_o = rays_o.float().contiguous()
        rays_d = rays_d.float().contiguous()
        if not rays_o.is_cuda:
            rays_o = rays_o.cuda()
        if not rays_d.is_cuda:
            rays_d = rays_d.cuda()
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.view(-1, 3)
        rays_d = rays_d.view(-1, 3)
        n = rays_o.shape[0]
        positions = rays_o if inplace else torch.empty_like(rays_o)
        face_normals = rays_d if inplace else torch.empty_like(rays_d)
        depth = torch.empty(n, dtype=torch.float32, device=rays_o.device)
        if inplace:                      # the kernel reads o/d before it writes: each thread owns its ray
            pass
        if _chunks is not None:
            sp = None if _skip is None else (_skip.data_ptr() if torch.is_tensor(_skip) else int(_skip))
            arr = (C.c_int * len(_chunks))(*[int(c) for c in _chunks])
            L.check(L.lib.nero_bvh_trace_ordered(self._handle(), C.c_void_p(rays_o.data_ptr()), C.c_void_p(rays_d.data_ptr()), n, C.c_void_p(sp), arr,
                                                 len(_chunks), C.c_void_p(positions.data_ptr()), C.c_void_p(face_normals.data_ptr()),
                                                 C.c_void_p(depth.data_ptr()), L.stream_ptr()))
        elif _skip is not None:
            sp = _skip.data_ptr() if torch.is_tensor(_skip) else int(_skip)
            if torch.is_tensor(_skip):
                assert _skip.dtype == torch.uint8 and _skip.is_cuda and _skip.numel() == n and _skip.is_contiguous()
            L.check(L.lib.nero_bvh_trace_masked(self._handle(), C.c_void_p(rays_o.data_ptr()), C.c_void_p(rays_d.data_ptr()), n, C.c_void_p(sp),
                                                C.c_void_p(positions.data_ptr()), C.c_void_p(face_normals.data_ptr()),
                                                C.c_void_p(depth.data_ptr()), L.stream_ptr()))
        elif _order is not None:
            L.check(L.lib.nero_bvh_trace_grouped(self._handle(), C.c_void_p(rays_o.data_ptr()), C.c_void_p(rays_d.data_ptr()), n,
                                                 C.c_void_p(positions.data_ptr()), C.c_void_p(face_normals.data_ptr()),
                                                 C.c_void_p(depth.data_ptr()), _order[0], _order[1], L.stream_ptr()))
        else:
            L.check(L.lib.nero_bvh_trace(self._handle(), C.c_void_p(rays_o.data_ptr()), C.c_void_p(rays_d.data_ptr()), n,
                                         C.c_void_p(positions.data_ptr()), C.c_void_p(face_normals.data_ptr()),
                                         C.c_void_p(depth.data_ptr()), L.stream_ptr()))
        return positions.view(*prefix, 3), face_normals.view(*prefix, 3), depth.view(*prefix)
