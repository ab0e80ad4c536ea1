"""ctypes face of the C-level Stage-II shading driver (include/nero_hip.h: nero_stage2_*; nero_amd/csrc/stage2_driver.hip):
predict_materials and the Monte-Carlo shader (MCShadingNetwork.shade_mixed / get_lights, network/field.py:856-1012) run as a handful of C
calls instead of the ~120 launches nero_amd/material_step.py sequences from Python.  The mesh tracer stays a Python-visible call between
nero_stage2_rays and nero_stage2_shade_fwd (nero_amd.raytracing.RayTracer in production, the oracle tracer in teacher-forced tests)."""
import ctypes as C
import os

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
_lib.nero_stage2_dead_rays.restype = C.c_void_p
_lib.nero_stage2_counts.argtypes = [_fp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
_lib.nero_stage2_dead_rays.argtypes = [_fp]
_lib.nero_stage2_shade_fwd.argtypes = [_fp] + [_fp] * 8 + [C.POINTER(C.c_int), C.POINTER(C.c_int), _fp]
_lib.nero_stage2_shade_bwd.argtypes = [_fp, _fp, _fp, C.POINTER(Weights), _fp, _fp]
_lib.nero_stage2_predict_bwd.argtypes = [_fp, _fp, C.POINTER(Weights), _fp]

GEOMETRY_TYPES = {'schlick': 0, 'ggx_smith': 1}


def supported():
    return all(GEMM_MODE[k] == L.GEMM_F16X3 for k in ('fwd', 'bwd', 'dw'))


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
        self._ws_need = {}

        def table(n):                                  # the fixed Fibonacci (azimuth, elevation) tables, network/field.py:741-749
            az, el = fibonacci_az_el(n)
            return torch.from_numpy(np.stack([az * 0.5 / np.pi, 1 - 2 * el / np.pi], -1).astype(np.float32)).to(device).contiguous()
        self.tab_d, self.tab_s = table(self.Dd), table(self.Ds)

    def __del__(self):
        h = getattr(self, 'h', None)
        if h and _lib is not None:                     # (module globals may already be cleared at interpreter exit)
            _lib.nero_stage2_destroy(h)
            self.h = None

    def matches_current_modes(self):
        return self.modes == (GEMM_MODE['fwd'], GEMM_MODE['bwd'], GEMM_MODE['dw'])

    def workspace(self, n_pred, P):
        if (n_pred, P) not in self._ws_need:          # (cached: the query is a dry run of the whole step on a copy of the handle)
            self._ws_need[(n_pred, P)] = _lib.nero_stage2_workspace_bytes(self.h, n_pred, P)
        need = self._ws_need[(n_pred, P)]
        if self._ws is None or self._ws.numel() < need:
            held = 0 if self._ws is None else self._ws.numel()
            L.check_workspace_fits(need, self.device, held, f'Stage-II step workspace for {P} points per GPU')
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
    missing = [k for k, name in enumerate(names) if name not in gv]
    if missing:                                    # views of ONE zero buffer: one fill launch, not one per tensor (nero_amd.stage1.RenderCoreC.backward)
        sizes = []
        for k in missing:
            n = 1
            for s_ in shapes[k]:
                n *= s_
            sizes.append(n)
        for k, t in zip(missing, torch.zeros(sum(sizes), dtype=torch.float32, device=device).split(sizes)):
            fresh[names[k]] = t.view(shapes[k])
    for k in range(len(names) // 2):
        nw, nb = names[2 * k], names[2 * k + 1]
        dW, db = (gv[nw] if nw in gv else fresh[nw]), (gv[nb] if nb in gv else fresh[nb])
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
        d_raw_c = d_raw.contiguous()
        L.check(_lib.nero_stage2_predict_bwd(drv.h, _p(d_raw_c), C.byref(G), L.stream_ptr()))
        return (None, None, None, None, None) + tuple(fresh.get(nm) for nm in ctx.names)


def descending_chunks(D):
    """launch order of the 64-ray chunks of a surface point's D secondary rays for RayTracer.trace_masked: descending.  Both direction tables run
    from the pole of their lobe outwards (network/field.py:741-749), so the later entries -- of the cosine table and of the GGX table alike -- are
    the grazing directions, the long BVH traversals; started first, they no longer end the launch on a handful of resident waves (tracer
    0.69 -> 0.57 ms per C4 step; same outputs).  None when D is not a whole number of chunks (<= 32), or with NERO_TRACE_ORDER=natural."""
    if D % 64 or D // 64 > 32 or D // 64 < 2 or os.environ.get('NERO_TRACE_ORDER', 'descending') == 'natural':
        return None
    return list(range(D // 64 - 1, -1, -1))


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
        # closest hit, depth >= 10 <=> miss.  A tracer that takes a launch-order hint starts the specular chunks of every point first
        # (nero_bvh_trace_grouped: same outputs); any other RayTracer-shaped object (tests, a reference-side tracer) gets the plain call
        # Round 6: the driver flags the rays whose estimator weight is exactly zero (GGX directions below the shading horizon under the
        # Schlick geometry term: mc_shade.hip, DEAD_SLOT); a tracer that takes the flags does not traverse them (they are the longest rays of
        # the launch), and shade_fwd leaves them out of both light MLPs whatever the tracer reported
        tg = getattr(tracer, 'trace_grouped', None)
        tm = getattr(tracer, 'trace_masked', None)
        dead = _lib.nero_stage2_dead_rays(drv.h)
        order = descending_chunks(D) if tm is not None else None
        if tm is not None and (dead or order is not None):
            pos, fnrm, depth = tm(orig, dirs, dead, chunk_order=order)
        elif tg is not None and os.environ.get('NERO_TRACE_ORDER', 'descending') == 'grouped':
            pos, fnrm, depth = tg(orig, dirs, D, drv.Dd)
        else:
            pos, fnrm, depth = tracer.trace(orig, dirs)
        pos, fnrm, depth = pos.contiguous(), fnrm.contiguous(), depth.contiguous().reshape(-1)
        rgb, dl, sl, sp = (torch.empty((Pn, 3), **f32) for _ in range(4))
        n_miss, n_hit = C.c_int(0), C.c_int(0)
        L.check(_lib.nero_stage2_shade_fwd(drv.h, _p(pos), _p(fnrm), _p(depth), _p(poses), _p(rgb), _p(dl), _p(sl), _p(sp), C.byref(n_miss),
                                           C.byref(n_hit), L.stream_ptr()))
        n_hum = C.c_int(0)
        _lib.nero_stage2_counts(drv.h, None, None, C.byref(n_hum))
        drv.last_counts = (n_miss.value, n_hit.value, Pn * D, n_hum.value)  # (miss rows, hit rows, rays -- the rest are zero-weight rays nobody shades --, human-light rows)
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
        d_rgb_c, d_dl_c = d_rgb.contiguous(), (d_dl.contiguous() if d_dl is not None else None)     # locals: both live across the C call
        L.check(_lib.nero_stage2_shade_bwd(drv.h, _p(d_rgb_c), _p(d_dl_c), C.byref(G),
                                           _p(d_mat5), L.stream_ptr()))
        ctx.keep = None
        return (None, None, None, None, None, None, None, d_mat5, None, None, None) + tuple(fresh.get(nm) for nm in ctx.names)


# ----------------------------------------------------------------------------------------------------------------------
# the training glue around the two calls above as single launches (nero_amd/csrc/mat_loss.hip)
# ----------------------------------------------------------------------------------------------------------------------
class LossCfg(C.Structure):
    _fields_ = [('rgb_l1', C.c_int), ('reg_mat', C.c_int), ('reg_change', C.c_int), ('reg_lambda1', C.c_float), ('hinge_weight', C.c_float),
                ('reg_diffuse', C.c_int), ('reg_diffuse_lambda', C.c_float)]


_lib.nero_mat_reg_points.argtypes = [C.c_int, _fp, _fp, _fp, _fp, C.c_float, _fp, _fp]
_lib.nero_mat_head_fwd.argtypes = [C.c_int, _fp, _fp, _fp]
_lib.nero_mat_head_bwd.argtypes = [C.c_int, _fp, _fp, _fp, _fp]
_lib.nero_mat_loss_partials.argtypes = [C.c_int]
_lib.nero_mat_loss_fwd.argtypes = [C.POINTER(LossCfg), C.c_int, C.c_int] + [_fp] * 8
_lib.nero_mat_loss_bwd.argtypes = [C.POINTER(LossCfg), C.c_int, C.c_int] + [_fp] * 9


def loss_cfg(renderer_cfg, shader_cfg, step, world=1):
    """the switches of NeROMaterialRenderer.train_step's losses (network/renderer.py:837-844, network/field.py:1061-1087) for one step"""
    if renderer_cfg['rgb_loss'] not in ('charbonier', 'l1'):
        raise NotImplementedError(renderer_cfg['rgb_loss'])
    hinge = bool(shader_cfg['reg_min_max']) and step is not None and step < 2000
    return LossCfg(int(renderer_cfg['rgb_loss'] == 'l1'), int(bool(renderer_cfg['reg_mat'])), int(bool(shader_cfg['reg_change'])),
                   float(shader_cfg['reg_lambda1']), float(world) if hinge else 0.0, int(bool(renderer_cfg['reg_diffuse_light'])),
                   float(renderer_cfg['reg_diffuse_light_lambda']))


def reg_points(pts, normals, ang01, eps):
    """[pts ; perturbed pts] as one [2P,3] tensor (regularization_points, network/field.py:1066-1076).  ang01 [P] uniform in [0,1);
    eps: [P] tensor (change_type 'gaussian') or a float ('constant')"""
    P = pts.shape[0]
    pts, normals, ang01 = pts.contiguous().float(), normals.contiguous().float(), ang01.reshape(-1).contiguous().float()
    out = torch.empty((2 * P, 3), dtype=torch.float32, device=pts.device)
    e = eps.reshape(-1).contiguous().float() if torch.is_tensor(eps) else None
    L.check(_lib.nero_mat_reg_points(P, _p(pts), _p(normals), _p(ang01), _p(e), 0.0 if e is not None else float(eps), _p(out), L.stream_ptr()))
    return out


class MaterialHeadC(torch.autograd.Function):
    """raw [n,5] -> (metallic, roughness in [0.04^2, 1], albedo) (predict_materials, network/field.py:915-922)"""

    @staticmethod
    def forward(ctx, raw):
        raw = raw.contiguous()
        mat = torch.empty_like(raw)
        L.check(_lib.nero_mat_head_fwd(raw.shape[0], _p(raw), _p(mat), L.stream_ptr()))
        ctx.raw = raw
        return mat

    @staticmethod
    def backward(ctx, d_mat):
        d_raw = torch.empty_like(ctx.raw)
        d_mat_c = d_mat.contiguous()
        L.check(_lib.nero_mat_head_bwd(ctx.raw.shape[0], _p(ctx.raw), _p(d_mat_c), _p(d_raw), L.stream_ptr()))
        return d_raw


class MaterialLossC(torch.autograd.Function):
    """(mat [P or 2P,5], rgb_lin [P,3], diffuse light [P,3], gt [P,3]) -> loss terms [4] = (total, mean loss_rgb, mean loss_mat_reg,
    mean loss_diffuse_light), rgb_pr [P,3].  Only loss[0] carries a gradient."""

    @staticmethod
    def forward(ctx, lcfg, P, mat, rgb_lin, dl, gt):
        mat, rgb_lin, dl, gt = (t.contiguous().float() for t in (mat, rgb_lin, dl, gt))
        has_reg = int(mat.shape[0] == 2 * P)
        assert mat.shape[0] in (P, 2 * P) and rgb_lin.shape == (P, 3) and dl.shape == (P, 3) and gt.shape == (P, 3)
        f32 = dict(dtype=torch.float32, device=mat.device)
        loss, rgb_pr = torch.empty(4, **f32), torch.empty((P, 3), **f32)
        part = torch.empty(_lib.nero_mat_loss_partials(P), **f32)
        L.check(_lib.nero_mat_loss_fwd(C.byref(lcfg), P, has_reg, _p(mat), _p(rgb_lin), _p(dl), _p(gt), _p(rgb_pr), _p(part), _p(loss), L.stream_ptr()))
        ctx.lcfg, ctx.P, ctx.has_reg, ctx.keep = lcfg, P, has_reg, (mat, rgb_lin, dl, gt)
        ctx.mark_non_differentiable(rgb_pr)
        return loss, rgb_pr

    @staticmethod
    def backward(ctx, d_loss, _d_rgb_pr):
        mat, rgb_lin, dl, gt = ctx.keep
        g = d_loss[0:1].contiguous()                        # (the other three entries are reporting only)
        d_mat, d_rgb, d_dl = torch.empty_like(mat), torch.empty_like(rgb_lin), torch.empty_like(dl)
        L.check(_lib.nero_mat_loss_bwd(C.byref(ctx.lcfg), ctx.P, ctx.has_reg, _p(mat), _p(rgb_lin), _p(dl), _p(gt), _p(g), _p(d_mat), _p(d_rgb), _p(d_dl),
                                       L.stream_ptr()))
        ctx.keep = None
        return None, None, d_mat, d_rgb, d_dl, None
