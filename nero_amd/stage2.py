"""ctypes face of the C-level Stage-II shading driver (include/nero_hip.h: nero_stage2_*; nero_amd/csrc/stage2_driver.hip):
predict_materials and the Monte-Carlo shader (MCShadingNetwork.shade_mixed / get_lights, network/field.py:856-1012) run as a handful of C
calls instead of the ~120 launches nero_amd/material_step.py sequences from Python.  The mesh tracer stays a Python-visible call between
nero_stage2_rays and nero_stage2_shade_fwd (nero_amd.raytracing.RayTracer in production, the oracle tracer in teacher-forced tests)."""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .chain import GEMM_MODE
from .fields import fibonacci_az_el
from .stage1 import Linear, _p

N_LIN = 32
_fp = C.c_void_p


class Weights(C.Structure):
    _fields_ = [('lin', Linear * N_LIN)]


class Cfg(C.Structure):
    _fields_ = [('diffuse_sample_num', C.c_int), ('specular_sample_num', C.c_int), ('human_lights', C.c_int), ('sphere_direction', C.c_int),
                ('geometry_type', C.c_int), ('light_exp_max', C.c_float), ('inner_light_exp_max', C.c_float), ('gemm_fwd', C.c_int),
                ('gemm_bwd', C.c_int), ('gemm_dw', C.c_int)]


_lib = L.lib
_lib.nero_stage2_pack_bytes.restype = C.c_size_t
_lib.nero_stage2_pack_bytes.argtypes = [_fp]
_lib.nero_stage2_workspace_bytes.restype = C.c_size_t
_lib.nero_stage2_workspace_bytes.argtypes = [_fp, C.c_int, C.c_int]
_lib.nero_stage2_destroy.restype = None
_lib.nero_stage2_destroy.argtypes = [_fp]
_lib.nero_stage2_pack.argtypes = [_fp, C.POINTER(Weights), _fp, _fp]
_lib.nero_stage2_predict_fwd.argtypes = [_fp, _fp, C.c_int, _fp, _fp, C.c_size_t, _fp]
_lib.nero_stage2_rays.argtypes = [_fp, C.c_int] + [_fp] * 11
_lib.nero_stage2_shade_fwd.argtypes = [_fp] + [_fp] * 8 + [C.POINTER(C.c_int), C.POINTER(C.c_int), _fp]
_lib.nero_stage2_shade_bwd.argtypes = [_fp, _fp, _fp, C.POINTER(Weights), _fp, _fp]
_lib.nero_stage2_predict_bwd.argtypes = [_fp, _fp, C.POINTER(Weights), _fp]

GEOMETRY_TYPES = {'schlick': 0, 'ggx_smith': 1}


def supported():
    return GEMM_MODE['fwd'] in (L.GEMM_F16X3, L.GEMM_F16X3P) and GEMM_MODE['bwd'] == L.GEMM_F16X3 and GEMM_MODE['dw'] in (L.GEMM_F16X3, L.GEMM_F16X3P)


class Stage2Driver:
    def __init__(self, shader_cfg, device='cuda'):
        if shader_cfg['outer_light_version'] not in ('direction', 'sphere_direction'):
            raise NotImplementedError(shader_cfg['outer_light_version'])
        if shader_cfg['geometry_type'] not in GEOMETRY_TYPES:
            raise NotImplementedError(shader_cfg['geometry_type'])
        self.device, self.scfg = device, shader_cfg
        self.Dd, self.Ds = shader_cfg['diffuse_sample_num'], shader_cfg['specular_sample_num']
        c = Cfg(self.Dd, self.Ds, int(bool(shader_cfg['human_lights'])), int(shader_cfg['outer_light_version'] == 'sphere_direction'),
                GEOMETRY_TYPES[shader_cfg['geometry_type']], float(shader_cfg['light_exp_max']), float(shader_cfg['inner_light_exp_max']),
                GEMM_MODE['fwd'], GEMM_MODE['bwd'], GEMM_MODE['dw'])
        self.modes = (GEMM_MODE['fwd'], GEMM_MODE['bwd'], GEMM_MODE['dw'])
        h = _fp()
        L.check(_lib.nero_stage2_create(C.byref(c), C.byref(h)))
        self.h = h
        self.n_lin = 32 if c.human_lights else 28
        self._pack_buf = torch.empty(_lib.nero_stage2_pack_bytes(h), dtype=torch.uint8, device=device)
        self._ws, self._w, self._keep = None, Weights(), None

        def table(n):                                  # the fixed Fibonacci (azimuth, elevation) tables, network/field.py:741-749
            az, el = fibonacci_az_el(n)
            return torch.from_numpy(np.stack([az * 0.5 / np.pi, 1 - 2 * el / np.pi], -1).astype(np.float32)).to(device).contiguous()
        self.tab_d, self.tab_s = table(self.Dd), table(self.Ds)

    def __del__(self):
        h = getattr(self, 'h', None)
        if h:
            _lib.nero_stage2_destroy(h)
            self.h = None

    def matches_current_modes(self):
        return self.modes == (GEMM_MODE['fwd'], GEMM_MODE['bwd'], GEMM_MODE['dw'])

    def workspace(self, n_pred, P):
        need = _lib.nero_stage2_workspace_bytes(self.h, n_pred, P)
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def pack(self, eff):
        """eff: effective weights in nero_amd.material_step.flatten_material_effective order"""
        assert len(eff) == 2 * self.n_lin, (len(eff), self.n_lin)
        for i in range(self.n_lin):
            W, b = eff[2 * i], eff[2 * i + 1]
            assert W.is_contiguous() and b.is_contiguous() and W.dtype == torch.float32
            self._w.lin[i].W, self._w.lin[i].b = W.data_ptr(), b.data_ptr()
        self._keep = list(eff)
        L.check(_lib.nero_stage2_pack(self.h, C.byref(self._w), self._pack_buf.data_ptr(), L.stream_ptr()))
        return self


def _grad_table(names, shapes, gv, lo, hi, device):
    """pointer table for Linears lo..hi-1 of `names` (global Linear index = lo + position); fresh zero tensors where no bucket view exists"""
    G, fresh = Weights(), {}
    f32 = dict(dtype=torch.float32, device=device)
    for k in range(len(names) // 2):
        nw, nb = names[2 * k], names[2 * k + 1]
        if nw in gv:
            dW, db = gv[nw], gv[nb]
        else:
            dW, db = torch.zeros(shapes[2 * k], **f32), torch.zeros(shapes[2 * k + 1], **f32)
            fresh[nw], fresh[nb] = dW, db
        assert dW.is_contiguous() and db.is_contiguous()
        G.lin[lo + k].W, G.lin[lo + k].b = dW.data_ptr(), db.data_ptr()
    return G, fresh


class PredictMaterialsC(torch.autograd.Function):
    """nero_amd.material_step.PredictMaterials through nero_stage2_predict_fwd / _bwd.  `n_shade`: how many of the rows are shaded
    afterwards (sizes the step workspace this call opens)."""

    @staticmethod
    def forward(ctx, drv, names, gv, n_shade, x, *params):
        x = x.contiguous().float()
        n = x.shape[0]
        raw = torch.empty((n, 5), dtype=torch.float32, device=x.device)
        ws = drv.workspace(n, n_shade)
        L.check(_lib.nero_stage2_predict_fwd(drv.h, _p(x), n, _p(raw), ws.data_ptr(), ws.numel(), L.stream_ptr()))
        ctx.drv, ctx.names, ctx.gv, ctx.shapes, ctx.x = drv, names, (gv or {}), [tuple(p.shape) for p in params], x
        return raw

    @staticmethod
    def backward(ctx, d_raw):
        drv = ctx.drv
        G, fresh = _grad_table(ctx.names, ctx.shapes, ctx.gv, 0, len(ctx.names) // 2, d_raw.device)
        L.check(_lib.nero_stage2_predict_bwd(drv.h, _p(d_raw.contiguous()), C.byref(G), L.stream_ptr()))
        return (None, None, None, None, None) + tuple(fresh.get(nm) for nm in ctx.names)


class MCShadeC(torch.autograd.Function):
    """nero_amd.material_step.MCShade through nero_stage2_rays / _shade_fwd / _shade_bwd"""

    @staticmethod
    def forward(ctx, drv, tracer, names, gv, pts, view, normals, mat5, rand_d, rand_s, poses, *params):
        dev = pts.device
        f32 = dict(dtype=torch.float32, device=dev)
        Pn = pts.shape[0]
        D = drv.Dd + drv.Ds
        pts, view, normals, mat5 = (t.detach().contiguous().float() for t in (pts, view, normals, mat5))
        rd = rand_d.reshape(-1).contiguous() if rand_d is not None else None
        rs = rand_s.reshape(-1).contiguous() if rand_s is not None else None
        human = bool(drv.scfg['human_lights'])
        if human and poses is None:
            raise ValueError('shader_cfg.human_lights needs human_poses [P,3,4]')
        poses = poses.detach().contiguous().float() if (poses is not None and human) else None
        orig, dirs = torch.empty((Pn * D, 3), **f32), torch.empty((Pn * D, 3), **f32)
        L.check(_lib.nero_stage2_rays(drv.h, Pn, _p(pts), _p(view), _p(normals), _p(mat5), _p(rd), _p(rs), _p(drv.tab_d), _p(drv.tab_s), _p(orig),
                                      _p(dirs), L.stream_ptr()))
        pos, fnrm, depth = tracer.trace(orig, dirs)                       # closest hit, depth >= 10 <=> miss
        pos, fnrm, depth = pos.contiguous(), fnrm.contiguous(), depth.contiguous().reshape(-1)
        rgb, dl, sl, sp = (torch.empty((Pn, 3), **f32) for _ in range(4))
        n_miss, n_hit = C.c_int(0), C.c_int(0)
        L.check(_lib.nero_stage2_shade_fwd(drv.h, _p(pos), _p(fnrm), _p(depth), _p(poses), _p(rgb), _p(dl), _p(sl), _p(sp), C.byref(n_miss),
                                           C.byref(n_hit), L.stream_ptr()))
        ctx.drv, ctx.names, ctx.gv, ctx.shapes, ctx.P = drv, names, (gv or {}), [tuple(p.shape) for p in params], Pn
        ctx.keep = (pts, view, normals, mat5, rd, rs, poses, orig, dirs, pos, fnrm, depth)
        ctx.mark_non_differentiable(sl, sp)
        return rgb, dl, sl, sp

    @staticmethod
    def backward(ctx, d_rgb, d_dl, _d_sl, _d_sp):
        drv = ctx.drv
        dev = d_rgb.device
        G, fresh = _grad_table(ctx.names, ctx.shapes, ctx.gv, 20, 20 + len(ctx.names) // 2, dev)
        d_mat5 = torch.empty((ctx.P, 5), dtype=torch.float32, device=dev)
        L.check(_lib.nero_stage2_shade_bwd(drv.h, _p(d_rgb.contiguous()), _p(d_dl.contiguous() if d_dl is not None else None), C.byref(G),
                                           _p(d_mat5), L.stream_ptr()))
        ctx.keep = None
        return (None, None, None, None, None, None, None, d_mat5, None, None, None) + tuple(fresh.get(nm) for nm in ctx.names)
