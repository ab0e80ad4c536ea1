"""ctypes face of the C-level Stage-I step driver (include/nero_hip.h: nero_stage1_*; nero_amd/csrc/stage1_driver.hip).

The driver runs sample_ray, render_core and their backward (network/renderer.py:403-443, 445-463, 550-606) as ONE C call each: the
launch sequence nero_amd/shape_step.py issues from Python (~200 ctypes calls per step) lives in the library, the workspace is one
caller-owned buffer.  This module only moves pointers: it owns the workspace / packed-image tensors, fills the weight and gradient
pointer tables, and wraps fwd / bwd in a torch.autograd.Function so that the loss assembly and the optimiser stay what they were."""
import ctypes as C

import torch

from . import _lib as L
from .chain import GEMM_MODE, row_pad

N_LIN = 49
_fp = C.c_void_p


class Linear(C.Structure):
    _fields_ = [('W', _fp), ('b', _fp)]


class Weights(C.Structure):
    _fields_ = [('lin', Linear * N_LIN)]


class Grads(C.Structure):
    _fields_ = [('lin', Linear * N_LIN)]          # (dW, db): same layout


class Cfg(C.Structure):
    _fields_ = [('n_samples', C.c_int), ('n_importance', C.c_int), ('n_bg_samples', C.c_int), ('up_sample_steps', C.c_int),
                ('clip_sample_variance', C.c_int), ('human_light', C.c_int), ('sphere_direction', C.c_int), ('light_exp_max', C.c_float),
                ('gemm_fwd', C.c_int), ('gemm_tan', C.c_int), ('gemm_bwd', C.c_int), ('gemm_dw', C.c_int)]


class State(C.Structure):
    _fields_ = [('R', C.c_int), ('T', C.c_int), ('n_in', C.c_int), ('n_out', C.c_int), ('pts4', _fp), ('ray_counts', _fp), ('ray_off', _fp),
                ('counts', _fp), ('inner_idx', _fp), ('outer_idx', _fp), ('x4', _fp), ('sdf4', _fp), ('feat', _fp), ('normal', _fp),
                ('geo', _fp), ('weights', _fp)]


_lib = L.lib
_lib.nero_stage1_pack_bytes.restype = C.c_size_t
_lib.nero_stage1_workspace_bytes.restype = C.c_size_t
_lib.nero_stage1_workspace_bytes_for.restype = C.c_size_t
_lib.nero_stage1_workspace_bytes_fwd.restype = C.c_size_t
_lib.nero_stage1_destroy.restype = None
_lib.nero_stage1_sample.argtypes = [_fp, C.c_int] + [_fp] * 8 + [_fp, C.c_size_t, _fp]
_lib.nero_stage1_render_fwd.argtypes = [_fp, C.c_int] + [_fp] * 6 + [C.c_float] + [_fp] * 3 + [C.POINTER(C.c_int), C.POINTER(C.c_int), _fp, C.c_size_t, _fp]
_lib.nero_stage1_render_bwd.argtypes = [_fp, _fp, _fp, _fp, C.POINTER(Grads), _fp, _fp]
_lib.nero_stage1_sdf_from_pe.argtypes = [_fp, _fp, C.c_int, _fp, _fp, C.c_size_t, _fp]
_lib.nero_stage1_pack.argtypes = [_fp, C.POINTER(Weights), _fp, _fp]
_lib.nero_stage1_workspace_bytes.argtypes = [_fp, C.c_int]
_lib.nero_stage1_workspace_bytes_fwd.argtypes = [_fp, C.c_int]
_lib.nero_stage1_workspace_bytes_for.argtypes = [_fp, C.c_int, C.c_int, C.c_int, C.c_int]
_lib.nero_stage1_pack_bytes.argtypes = [_fp]
_lib.nero_stage1_get_state.argtypes = [_fp, C.POINTER(State)]
_lib.nero_stage1_destroy.argtypes = [_fp]


def supported(cfg=None, shader_cfg=None):
    """the C driver packs fp16 two-plane operands only (the default engines) and is laid out for the network shapes of the shipped
    YAMLs: 8 x 256 SDF layers on a PE-6 input, PE-8 positions in front of the light MLPs.  Other values of sdf_n_layers / sdf_freq /
    shader_config.light_pos_freq run on the Python-sequenced chains (nero_amd/shape_step.py), which take any of them."""
    ok = all(GEMM_MODE[k] == L.GEMM_F16X3 for k in ('fwd', 'tan', 'bwd', 'dw'))
    if cfg is not None:
        ok = ok and int(cfg.get('sdf_n_layers', 8)) == 8 and int(cfg.get('sdf_freq', 6)) == 6
    if shader_cfg is not None:
        ok = ok and int(shader_cfg.get('light_pos_freq', 8)) == 8
    return ok


def current_modes():
    return (GEMM_MODE['fwd'], GEMM_MODE['tan'], GEMM_MODE['bwd'], GEMM_MODE['dw'])


def _p(t):
    return None if t is None else t.data_ptr()


class Stage1Driver:
    """one NeROShapeRenderer configuration on the C-level driver.  pack(eff) once per optimisation step, then sample / render."""

    def __init__(self, cfg, shader_cfg, device='cuda'):
        self.device = device
        c = Cfg(cfg['n_samples'], cfg['n_importance'], cfg['n_bg_samples'], cfg['up_sample_steps'], int(bool(cfg['clip_sample_variance'])),
                int(bool(shader_cfg.get('human_light', False))), int(bool(shader_cfg.get('sphere_direction', False))),
                float(shader_cfg.get('light_exp_max', 0.0)), GEMM_MODE['fwd'], GEMM_MODE['tan'], GEMM_MODE['bwd'], GEMM_MODE['dw'])
        self.cfg = c
        self.modes = current_modes()                  # the engines this handle packs for (nero_amd.chain.GEMM_MODE at construction)
        self.T = cfg['n_samples'] + cfg['n_importance'] + cfg['n_bg_samples']
        h = _fp()
        L.check(_lib.nero_stage1_create(C.byref(c), C.byref(h)))
        self.h = h
        self.n_lin = 49 if c.human_light else 45
        self._pack_buf = torch.empty(_lib.nero_stage1_pack_bytes(h), dtype=torch.uint8, device=device)
        self._ws = None
        self._ws_need = {}
        self.forward_only = False              # True (inference driver): the workspace is sized for sampler + forward only
        self._scratch = None
        self._w = Weights()
        self._keep = None

    def matches_current_modes(self):
        """False after nero_amd.chain.set_gemm_mode() selected other engines: the caller then takes the Python-sequenced path"""
        return self.modes == current_modes()

    def __del__(self):
        h = getattr(self, 'h', None)
        if h and _lib is not None:                     # (module globals may already be cleared at interpreter exit)
            _lib.nero_stage1_destroy(h)
            self.h = None

    # ---- memory ---------------------------------------------------------------------------------------------------------------
    def workspace_bytes(self, R, n_in=None, n_out=None):
        """(cached per R: the query is four dry runs of sampler + forward + backward on a copy of the handle, and it used to be repeated
        on every sample() and every render forward)"""
        if n_in is None:
            if R not in self._ws_need:
                q = _lib.nero_stage1_workspace_bytes_fwd if self.forward_only else _lib.nero_stage1_workspace_bytes
                self._ws_need[R] = q(self.h, R)
            return self._ws_need[R]
        return _lib.nero_stage1_workspace_bytes_for(self.h, R, n_in, n_out, 1)

    def workspace(self, R):
        """the step workspace for R rays: the worst case over the data-dependent inner / outer split, allocated once"""
        need = self.workspace_bytes(R)
        if self._ws is None or self._ws.numel() < need:
            held = 0 if self._ws is None else self._ws.numel()
            L.check_workspace_fits(need, self.device, held, f'Stage-I step workspace for {R} rays per GPU (worst case over the inner / outer split)')
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def _view(self, ptr, shape, dtype=torch.float32):
        """tensor view of a state pointer inside the workspace"""
        off = ptr - self._ws.data_ptr()
        n = 1
        for s in shape:
            n *= s
        return self._ws[off:off + n * 4].view(dtype).view(*shape)

    # ---- weights ----------------------------------------------------------------------------------------------------------------
    def pack(self, eff):
        """eff: the effective weights in nero_amd.shape_step.flatten_effective order [W0, b0, W1, b1, ...] (contiguous fp32)"""
        assert len(eff) == 2 * self.n_lin, (len(eff), self.n_lin)
        for i in range(self.n_lin):
            W, b = eff[2 * i], eff[2 * i + 1]
            assert W.is_contiguous() and b.is_contiguous() and W.dtype == torch.float32
            self._w.lin[i].W, self._w.lin[i].b = W.data_ptr(), b.data_ptr()
        self._keep = list(eff)
        L.check(_lib.nero_stage1_pack(self.h, C.byref(self._w), self._pack_buf.data_ptr(), L.stream_ptr()))
        return self

    # ---- the path -----------------------------------------------------------------------------------------------------------------
    def sample(self, o, d, near, far, variance, rand1=None, rand_bg=None):
        R = o.shape[0]
        z = torch.empty((R, self.T), dtype=torch.float32, device=o.device)
        ws = self.workspace(R)
        L.check(_lib.nero_stage1_sample(self.h, R, _p(o), _p(d), _p(near), _p(far), _p(variance), _p(rand1), _p(rand_bg), _p(z),
                                        ws.data_ptr(), ws.numel(), L.stream_ptr()))
        return z

    def sdf_from_pe(self, pe, n):
        """SDFField.sdf_from_pe on the driver's packed SDF chain (its own scratch: the step state in the workspace stays intact)"""
        rp = row_pad(n)
        need = rp * (256 + 4) * 4 + 4096
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.empty(need, dtype=torch.uint8, device=self.device)
        out = torch.empty((rp, 4), dtype=torch.float32, device=pe.device)
        L.check(_lib.nero_stage1_sdf_from_pe(self.h, _p(pe), n, _p(out), self._scratch.data_ptr(), self._scratch.numel(), L.stream_ptr()))
        return out

    def state(self):
        s = State()
        L.check(_lib.nero_stage1_get_state(self.h, C.byref(s)))
        return s


class _SdfAdapter:
    def __init__(self, drv):
        self._drv = drv

    def sdf_from_pe(self, pe, n):
        return self._drv.sdf_from_pe(pe, n)

    def pe_of_rays(self, o, d, z, col0, ncols):
        """PE-6 rows of the points o + z[:, col0 : col0 + ncols] d (the C driver exists for the YAML shapes only: stage1.supported)"""
        R = o.shape[0]
        pe = torch.empty((row_pad(R * ncols), 40), dtype=torch.float32, device=o.device)
        P = C.c_void_p                                   # (no argtypes are declared for this entry point here: hand over real pointers)
        L.check(_lib.nero_ray_points_pe(P(o.data_ptr()), P(d.data_ptr()), P(z.data_ptr()), z.stride(0), col0, ncols, R, P(pe.data_ptr()), L.stream_ptr()))
        return pe


class _KAdapter:
    """what nero_amd.shape_step.occ_loss / secondary_occlusion need of a ShapeKernels object"""

    def __init__(self, drv):
        self.sdf = _SdfAdapter(drv)


class RenderCoreC(torch.autograd.Function):
    """render_core through nero_stage1_render_fwd / _bwd.  Same contract as nero_amd.shape_step.RenderCore: inputs (meta, o, d, z_vals,
    variance, FG_LUT, poses, *effective weights), outputs ray_rgb [R,3], gradient_error [N_in], occ_prob [N_in] (unclamped)."""

    @staticmethod
    def forward(ctx, meta, o, d, z_vals, variance, lut, poses, *params):
        drv = meta['driver']
        R, T = z_vals.shape
        dev = o.device
        f32 = dict(dtype=torch.float32, device=dev)
        rgb = torch.empty((R, 3), **f32)
        gerr, occ = torch.empty(R * T, **f32), torch.empty(R * T, **f32)
        ws = drv.workspace(R)
        n_in, n_out = C.c_int(0), C.c_int(0)
        L.check(_lib.nero_stage1_render_fwd(drv.h, R, _p(o), _p(d), _p(z_vals), _p(variance), _p(lut), _p(poses), float(meta['anneal']), _p(rgb),
                                            _p(gerr), _p(occ), C.byref(n_in), C.byref(n_out), ws.data_ptr(), ws.numel(), L.stream_ptr()))
        n_in, n_out = n_in.value, n_out.value
        st = drv.state()
        rpi = row_pad(n_in)
        S = {'K': _KAdapter(drv), 'R': R, 'T': T, 'n_in': n_in, 'n_out': n_out, 'o': o, 'd': d, 'variance': variance, 'lut': lut,
             'pts4': drv._view(st.pts4, (R * T, 4)), 'weights': drv._view(st.weights, (R, T)),
             'inner_idx': drv._view(st.inner_idx, (max(n_in, 1),), torch.int32)}
        if n_in > 0:
            S.update(x4=drv._view(st.x4, (rpi, 4)), geo=drv._view(st.geo, (rpi, 8)),
                     sctx={'sdf4': drv._view(st.sdf4, (rpi, 4)), 'normal': drv._view(st.normal, (n_in, 3)), 'feat': drv._view(st.feat, (rpi, 256))})
        ctx.drv, ctx.meta_small, ctx.n_in = drv, {k: meta[k] for k in ('names', 'shapes', 'freeze_inv_s', 'grad_views')}, n_in
        ctx.variance = variance
        ctx.keep = (o, d, z_vals, lut, poses)
        meta['_state'] = S
        return rgb, gerr[:n_in], occ[:n_in]

    @staticmethod
    def backward(ctx, d_rgb, d_gerr, d_occ):
        drv, meta, n_in = ctx.drv, ctx.meta_small, ctx.n_in
        dev = d_rgb.device
        f32 = dict(dtype=torch.float32, device=dev)
        gv = meta['grad_views'] or {}
        G = Grads()
        fresh = {}
        missing = [k for k, name in enumerate(meta['names']) if name not in gv]
        if missing:
            # zeros (a network without samples this step -- n_in or n_out == 0 -- is not written), as views of ONE flat buffer: one fill
            # launch instead of one per tensor (92 at the YAML shapes: a fifth of the launches of a 512-ray drop-in step; round 6).  Fresh
            # every backward: torch keeps the views as the parameters' .grad
            sizes = [1]
            for k in missing:
                n = 1
                for s_ in meta['shapes'][k]:
                    n *= s_
                sizes.append(n)
            parts = torch.zeros(sum(sizes), **f32).split(sizes)
            dsum = parts[0]
            for k, t in zip(missing, parts[1:]):
                fresh[meta['names'][k]] = t.view(meta['shapes'][k])
        else:
            dsum = torch.zeros(1, **f32)
        for i in range(drv.n_lin):
            nw, nb = meta['names'][2 * i], meta['names'][2 * i + 1]
            dW, db = (gv[nw] if nw in gv else fresh[nw]), (gv[nb] if nb in gv else fresh[nb])
            assert dW.is_contiguous() and db.is_contiguous()
            G.lin[i].W, G.lin[i].b = dW.data_ptr(), db.data_ptr()
        d_gerr_c = d_gerr.contiguous() if (d_gerr is not None and n_in > 0) else None
        d_occ_c = d_occ.contiguous() if (d_occ is not None and n_in > 0) else None
        d_rgb_c = d_rgb.contiguous()          # (bound to a local: a temporary's block could be re-used by the next .contiguous() while the C call still reads it)
        L.check(_lib.nero_stage1_render_bwd(drv.h, _p(d_rgb_c), _p(d_gerr_c), _p(d_occ_c), C.byref(G), _p(dsum), L.stream_ptr()))
        d_var = None
        if n_in > 0 and not meta['freeze_inv_s']:
            v = ctx.variance.detach()
            inv_s = torch.exp(v * 10.0)
            live = ((inv_s >= 1e-6) & (inv_s <= 1e6)).to(torch.float32)
            d_var = dsum[0] * 10.0 * inv_s * live
        grads = [fresh.get(name) for name in meta['names']]       # None: written in place into the (pre-zeroed) flat bucket
        ctx.keep = None
        return (None, None, None, None, d_var, None, None) + tuple(grads)


# ---- the training glue between the driver calls (nero_amd/csrc/step_glue.hip) ---------------------------------------------------------
_lib.nero_occ_select_workspace.restype = C.c_size_t
_lib.nero_occ_select_workspace.argtypes = [C.c_int]
_lib.nero_near_far_sphere.argtypes = [_fp, _fp, C.c_int, _fp, _fp, _fp]
_lib.nero_occ_select.argtypes = [_fp, C.c_int, _fp, C.c_int, _fp, _fp, _fp, C.c_size_t, _fp]
_lib.nero_occ_gather.argtypes = [_fp, _fp, _fp, C.c_int, _fp, _fp, _fp]
_lib.nero_shape_loss_partials.argtypes = [C.c_int, C.c_int]
_lib.nero_shape_loss.argtypes = [C.c_int, C.c_int, _fp, _fp, C.c_int, _fp, C.c_float] + [_fp] * 11
_lib.nero_var_grad.argtypes = [_fp, _fp, _fp, _fp]
_lib.nero_occ_candidates.argtypes = [_fp, _fp, _fp, _fp, _fp, C.c_int, C.c_float, C.c_int, _fp, _fp]
RGB_LOSS_KIND = {'l2': 0, 'l1': 1, 'smooth_l1': 2, 'charbonier': 3}          # include/nero_hip.h NERO_RGB_*


class ShapeStepGlue:
    """One Stage-I training step (world-local part: render forward, the three loss terms, render backward into the flat gradient bucket)
    with the driver calls glued by nero_amd/csrc/step_glue.hip instead of torch: near / far, the occlusion-loss candidate subset chosen
    ON THE DEVICE (fixed-capacity march, no read-back of the candidate count), loss + backward seeds in two launches.  Same arithmetic
    as NeROShapeRenderer.render(is_train=True) + nero_amd.train.shape_training_loss + autograd (network/renderer.py:445-463, 522-606,
    train/trainer.py:127-137) -- tests/test_step_glue.py compares the two paths gradient by gradient.  The one host synchronisation
    left is the inner / outer sample count inside nero_stage1_render_fwd."""

    def __init__(self, net, drv, grad_views, names):
        self.net, self.drv = net, drv
        self.G = Grads()
        for i in range(drv.n_lin):
            dW, db = grad_views[names[2 * i]], grad_views[names[2 * i + 1]]
            assert dW.is_contiguous() and db.is_contiguous()
            self.G.lin[i].W, self.G.lin[i].b = dW.data_ptr(), db.data_ptr()
        self._keep = (grad_views, names)
        self._bufs = {}

    @staticmethod
    def supported(net):
        c = net.cfg
        # nero_occ_select sorts the kept candidates in LDS: cap <= 4096 (PICK_MAX in step_glue.hip); a larger cap keeps the tensor glue
        return (c.get('std_act', 'exp') == 'exp' and c['rgb_loss'] in RGB_LOSS_KIND
                and 1 <= int(c['occ_loss_max_pn']) <= 4096)

    def _buffers(self, R):
        if R not in self._bufs:
            dev, T, cap = self.drv.device, self.drv.T, int(self.net.cfg['occ_loss_max_pn'])
            f32, i32 = dict(dtype=torch.float32, device=dev), dict(dtype=torch.int32, device=dev)
            n = R * T
            self._bufs[R] = dict(
                near=torch.empty((R, 1), **f32), far=torch.empty((R, 1), **f32), rgb=torch.empty((R, 3), **f32), gerr=torch.empty(n, **f32),
                occ=torch.empty(n, **f32), d_rgb=torch.empty((R, 3), **f32), d_gerr=torch.empty(n, **f32), d_occ=torch.empty(n, **f32),
                flag=torch.empty(n, dtype=torch.uint8, device=dev), cand=torch.empty(cap, **i32), counts=torch.zeros(2, **i32),
                losses=torch.zeros(4, **f32), partials=torch.empty(_lib.nero_shape_loss_partials(R, n), **f32),
                pts=torch.empty((cap, 3), **f32), dirs=torch.empty((cap, 3), **f32), dsum=torch.zeros(1, **f32),
                sel_ws=torch.empty(_lib.nero_occ_select_workspace(n), dtype=torch.uint8, device=dev))
        return self._bufs[R]

    def forward_backward(self, o, d, gt, poses, step, variance_param, eik_weight, frozen, weights=None, rands=None):
        """o, d, gt [R,3] (device, fp32), poses [R,3,4] or None.  weights: device float32 [2] (eikonal, occlusion count weights of the
        data-parallel step) or a callable (n_in, occ_count_tensor) -> such a tensor, or None.  rands: (rand1 [R,1], rand_bg [R,n_bg],
        occ_keys [>= #candidates][, near, far]) to inject the random draws (tests).  Gradients land in the bucket views given at construction
        (the bucket must be zero on entry), d loss / d variance in variance_param.grad unless `frozen`.  -> dict(loss = device
        tensor [4]: total, rgb, eikonal, occlusion; n_in, n_out; occ_counts = device int32 [2] or None)."""
        from .shape_step import secondary_occlusion
        net, drv = self.net, self.drv
        c = net.cfg
        R, T = o.shape[0], drv.T
        B = self._buffers(R)
        st = L.stream_ptr()
        o, d, gt = o.contiguous(), d.contiguous(), gt.contiguous()
        if rands is not None and len(rands) > 3:          # (tests may hand over near / far: for unit directions the kernel's are bit-identical
            B['near'].copy_(rands[3])                     #  to the tensor expression's (tests/test_step_glue.py); for |d| != 1 they can differ
            B['far'].copy_(rands[4])                      #  in the last bit, which the hierarchical sampler amplifies)
        else:
            L.check(_lib.nero_near_far_sphere(_p(o), _p(d), R, _p(B['near']), _p(B['far']), st))
        var = variance_param.detach()
        nb = int(c['n_bg_samples'])
        if c['perturb'] > 0:
            if rands is not None:
                rand1, rand_bg = rands[0].contiguous(), rands[1].contiguous()
            else:
                rand1 = torch.rand([R, 1], device=o.device)              # (the same two draws, in the same order, as NeROShapeRenderer.render)
                rand_bg = torch.rand([R, nb], device=o.device)
        else:
            rand1 = rand_bg = None
        z = drv.sample(o, d, B['near'], B['far'], var, rand1, rand_bg)
        ws = drv.workspace(R)
        n_in, n_out = C.c_int(0), C.c_int(0)
        lut = net.color_network.FG_LUT
        poses_c = poses.to(torch.float32).contiguous() if poses is not None else None
        L.check(_lib.nero_stage1_render_fwd(drv.h, R, _p(o), _p(d), _p(z), _p(var), _p(lut), _p(poses_c), float(net.get_anneal_val(step)),
                                            _p(B['rgb']), _p(B['gerr']), _p(B['occ']), C.byref(n_in), C.byref(n_out), ws.data_ptr(), ws.numel(), st))
        n_in, n_out = n_in.value, n_out.value
        occ_on = bool(c['apply_occ_loss']) and step >= c['occ_loss_step'] and n_in > 0
        cand = counts = gt_occ = None
        if occ_on:
            s = drv.state()
            rpi = row_pad(n_in)
            x4, geo = drv._view(s.x4, (rpi, 4)), drv._view(s.geo, (rpi, 8))
            L.check(_lib.nero_occ_candidates(s.x4, s.sdf4, s.normal, s.inner_idx, _p(d), T, float(c['occ_sdf_thresh']), n_in, _p(B['flag']), st))
            keys = rands[2].to(o.device).contiguous() if (rands is not None and rands[2] is not None) else torch.rand(n_in, dtype=torch.float32, device=o.device)
            # RNG note: this path draws rand(n_in) keys every step >= occ_loss_step; the tensor glue draws rand(Pn) only when the
            # candidates exceed the cap, so NERO_STEP_GLUE=torch|hip runs are not seed-comparable from that step on.
            if keys.numel() < n_in:                       # occ_records_kernel reads keys[ordinal among the candidates], ordinal < n_in
                keys = torch.cat([keys, torch.full((n_in - keys.numel(),), float('inf'), dtype=torch.float32, device=o.device)])
            cap = B['cand'].numel()
            L.check(_lib.nero_occ_select(_p(B['flag']), n_in, _p(keys), cap, _p(B['cand']), _p(B['counts']), B['sel_ws'].data_ptr(), B['sel_ws'].numel(), st))
            L.check(_lib.nero_occ_gather(_p(x4), _p(geo), _p(B['cand']), cap, _p(B['pts']), _p(B['dirs']), st))
            gt_occ = secondary_occlusion(_KAdapter(drv), B['pts'], B['dirs'], var, 64, 16)
            cand, counts = B['cand'], B['counts']
        if callable(weights):
            weights = weights(n_in, counts)
        L.check(_lib.nero_shape_loss(R, RGB_LOSS_KIND[c['rgb_loss']], _p(B['rgb']), _p(gt), n_in, _p(B['gerr']), float(eik_weight), _p(B['occ']),
                                     _p(cand), _p(counts), _p(gt_occ), _p(weights), _p(B['losses']), _p(B['d_rgb']),
                                     _p(B['d_gerr']) if n_in > 0 else None, _p(B['d_occ']) if occ_on else None, _p(B['partials']), st))
        L.check(_lib.nero_stage1_render_bwd(drv.h, _p(B['d_rgb']), _p(B['d_gerr']) if n_in > 0 else None, _p(B['d_occ']) if occ_on else None,
                                            C.byref(self.G), _p(B['dsum']), st))
        if n_in > 0 and not frozen:
            L.check(_lib.nero_var_grad(_p(B['dsum']), _p(var), _p(variance_param.grad), st))
        return {'loss': B['losses'], 'n_in': n_in, 'n_out': n_out, 'occ_counts': counts}
