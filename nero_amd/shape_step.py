"""Stage-I render step on the HIP library: host-side orchestration only (buffer allocation through torch's caching
allocator, kernel sequencing, autograd glue).  Mirrors NeROShapeRenderer.sample_ray / render_core
(network/renderer.py:403-443, 550-606) and AppShadingNetwork.forward (network/field.py:591-651); every arithmetic step
is a call into libnero_hip.so (include/nero_hip.h).  See DESIGN.md §2 for the kernel sequence."""
import ctypes as C
import os

import torch

from . import _lib as L
from .chain import Chain, Dense, Head, row_pad
from .sdf import SDFField

P = C.c_void_p


def _p(t):
    return P(None if t is None else t.data_ptr())


def _st():
    return L.stream_ptr()


def predictor_entries(eff, k_main0, k_aux0=0):
    """eff: 4 (W,b).  Layer 0 may take [main(k_main0) | aux(k_aux0)] columns."""
    (W0, b0), (W1, b1), (W2, b2), (W3, b3) = eff
    return [(Dense(W0, b0, L.ACT_RELU, k_main0, 0, k_aux0, k_main0), None), (Dense(W1, b1, L.ACT_RELU, 256), None),
            (Dense(W2, b2, L.ACT_RELU, 256), None), (None, Head(W3, b3))]


class ShapeKernels:
    """all chains of one NeROShapeRenderer; rebuilt (cheap: descriptors only) whenever the effective weights change."""

    def __init__(self, eff, shader_cfg, device='cuda'):
        self.device = device
        self.human = bool(shader_cfg.get('human_light', False))
        self.sdf = SDFField(eff['sdf'], device)
        nf = eff['nerf']
        ent = []
        for i, (W, b) in enumerate(nf['pts']):
            if i == 0:
                ent.append((Dense(W, b, L.ACT_RELU, 84), None))
            elif i == 5:
                ent.append((Dense(W, b, L.ACT_RELU, 256, 84, 84, 0), None))
            else:
                ent.append((Dense(W, b, L.ACT_RELU, 256), None))
        ent.append((None, Head(*nf['alpha'])))
        self.nerf_trunk = Chain(ent, k_init=88, k_aux=88, aux_wide=True, device=device)
        self.nerf_head = Chain([(Dense(*nf['feature'], L.ACT_NONE, 256), None),
                                (Dense(*nf['views'], L.ACT_RELU, 256, 0, 27, 256), None),
                                (None, Head(*nf['rgb']))], k_init=256, k_aux=32, device=device)
        self.mat = [Chain(predictor_entries(eff[k], 256, 3), k_init=256, k_aux=8, device=device)
                    for k in ('metallic', 'roughness', 'albedo')]
        self.sphere = int(bool(shader_cfg.get('sphere_direction', False)))
        self.ld_outer = 144 if self.sphere else 72          # [IDE(v) | IDE(sphere exit point)] with shader_config.sphere_direction
        self.outer_light = Chain(predictor_entries(eff['outer_light'], self.ld_outer), k_init=self.ld_outer, device=device)
        self.pos_freq = int(shader_cfg.get('light_pos_freq', 8))
        self.pos_dim = 3 + 6 * self.pos_freq                 # network/field.py:515: get_embedder(light_pos_freq, 3)
        r8 = lambda k: (k + 7) // 8 * 8
        self.ld_xi, self.ld_xo = r8(self.pos_dim + 72), r8(self.pos_dim + 39)       # 128 / 96 at the YAMLs' PE-8
        self.inner_light = Chain(predictor_entries(eff['inner_light'], self.pos_dim + 72), k_init=self.ld_xi, device=device)
        self.inner_weight = Chain(predictor_entries(eff['inner_weight'], self.pos_dim + 39), k_init=self.ld_xo, device=device)
        self.human_light = Chain(predictor_entries(eff['human'], 24), k_init=24, device=device) if self.human else None

    def recode_positions(self, x4, n_in, Xi8, Xo8):
        """[PE-8(p) | IDE(refl, rough)] [rows,128] and [PE-8(p) | PE-6(refl)] [rows,96] -> the same rows with PE-f(p), f = light_pos_freq"""
        from .sdf import encode_pe
        rp, pd = Xi8.shape[0], self.pos_dim
        pe = encode_pe(x4, n_in, 3, self.pos_freq, (pd + 7) // 8 * 8)
        Xi = torch.zeros((rp, self.ld_xi), dtype=torch.float32, device=Xi8.device)
        Xo = torch.zeros((rp, self.ld_xo), dtype=torch.float32, device=Xi8.device)
        Xi[:, :pd], Xi[:, pd:pd + 72] = pe[:, :pd], Xi8[:, 51:123]
        Xo[:, :pd], Xo[:, pd:pd + 39] = pe[:, :pd], Xo8[:, 51:90]
        return Xi, Xo

    def pack(self):
        """(re)pack the operand images of all ten networks: ONE zero-filled flat buffer (kept and re-used while its size fits) and
        the pack jobs of every chain batched into a handful of launches (nero_pack_batch takes 64 jobs) -- ~10 launches per step
        instead of ~35 plus as many allocations."""
        from .chain import run_pack_jobs
        parts = [self.sdf, self.nerf_trunk, self.nerf_head, self.outer_light, self.inner_light, self.inner_weight, self.human_light] + self.mat
        parts = [c for c in parts if c is not None]
        sizes = [(c.pack_floats() + 63) // 64 * 64 for c in parts]          # 256-byte aligned images
        total = sum(sizes)
        flat = getattr(self, '_flat', None)
        if flat is None or flat.numel() != total:
            flat = self._flat = torch.zeros(total, dtype=torch.float32, device=self.device)
        else:
            flat.zero_()
        jobs, off = [], 0
        for c, n in zip(parts, sizes):
            jobs += c.pack(flat[off:off + n], run=False)
            off += n
        run_pack_jobs(jobs)
        return self


def flatten_effective(net):
    """-> (names, tensors): effective weights of a NeROShapeRenderer as a flat list of autograd tensors (every weight-normed Linear through
    ONE batched weight-norm node: fields.batched_weight_norm)"""
    from .fields import batched_weight_norm
    return batched_weight_norm(lambda: _flatten_effective(net), owner=net)


def _flatten_effective(net):
    names, ts = [], []

    def add(prefix, wb):
        names.extend([prefix + '.weight', prefix + '.bias'])
        ts.extend(wb)
    for l, wb in enumerate(net.sdf_network.effective()):
        add(f'sdf.{l}', wb)
    nf = net.outer_nerf.effective()
    for i, wb in enumerate(nf['pts']):
        add(f'nerf.pts.{i}', wb)
    for k in ('views', 'feature', 'alpha', 'rgb'):
        add(f'nerf.{k}', nf[k])
    cn = net.color_network
    preds = ['metallic_predictor', 'roughness_predictor', 'albedo_predictor', 'outer_light', 'inner_light', 'inner_weight']
    if cn.cfg['human_light']:
        preds.append('human_light_predictor')
    for pn in preds:
        for i, wb in enumerate(getattr(cn, pn).effective()):
            add(f'{pn}.{i}', wb)
    return names, ts


def unflatten_effective(names, ts):
    d = dict(zip(names, ts))

    def wb(prefix):
        return d[prefix + '.weight'], d[prefix + '.bias']
    eff = {'sdf': [wb(f'sdf.{l}') for l in range(sum(1 for k in d if k.startswith('sdf.') and k.endswith('.weight')))],
           'nerf': {'pts': [wb(f'nerf.pts.{i}') for i in range(8)], 'views': wb('nerf.views'), 'feature': wb('nerf.feature'),
                    'alpha': wb('nerf.alpha'), 'rgb': wb('nerf.rgb')}}
    for short, pn in (('metallic', 'metallic_predictor'), ('roughness', 'roughness_predictor'), ('albedo', 'albedo_predictor'),
                      ('outer_light', 'outer_light'), ('inner_light', 'inner_light'), ('inner_weight', 'inner_weight'),
                      ('human', 'human_light_predictor')):
        if f'{pn}.0.weight' in d:
            eff[short] = [wb(f'{pn}.{i}') for i in range(4)]
    return eff


# ----------------------------------------------------------------------------------------------------------------------
# sampling (no grad)
# ----------------------------------------------------------------------------------------------------------------------
def sample_ray(K, cfg, o, d, near, far, variance, rand1=None, rand_bg=None, trace=None):
    """network/renderer.py:403-443.  o,d [R,3]; near,far [R,1]; variance: 0-dim device tensor; rand1 [R,1] / rand_bg
    [R,n_bg] uniform draws (None = no perturbation).  -> z_vals [R, n_samples+n_importance+n_bg]"""
    dev = o.device
    R = o.shape[0]
    ns, nb, up = cfg['n_samples'], cfg['n_bg_samples'], cfg['up_sample_steps']
    m = cfg['n_importance'] // up
    n_in = ns + m * up
    T = n_in + nb
    z = torch.empty((R, T), dtype=torch.float32, device=dev)
    tab = torch.empty((R, n_in), dtype=torch.float32, device=dev)
    st = _st()
    lib = L.lib
    L.check(lib.nero_coarse_z(_p(near), _p(far), _p(rand1), R, ns, _p(z), T, st))
    pe = K.sdf.pe_of_rays(o, d, z, 0, ns)
    s4 = K.sdf.sdf_from_pe(pe, R * ns)
    L.check(lib.nero_scatter_sdf(_p(s4), 4, R, ns, _p(tab), n_in, st))
    n = ns
    z_new = torch.empty((R, m), dtype=torch.float32, device=dev)
    var_ptr = variance if cfg['clip_sample_variance'] else None
    for i in range(up):
        w_out = inds = index = None
        if trace is not None:
            w_out = torch.empty((R, n - 1), dtype=torch.float32, device=dev)
            inds = torch.empty((R, m), dtype=torch.int32, device=dev)
            index = torch.empty((R, n + m), dtype=torch.int32, device=dev)
            z_before = z[:, :n].clone()
            sdf_before = tab[:, :n].clone()
        L.check(lib.nero_upsample(_p(o), _p(d), _p(z), T, _p(tab), n_in, n, _p(var_ptr), C.c_float(64.0 * 2 ** i), m, R,
                                  _p(z_new), _p(w_out), _p(inds), st))
        last = (i + 1 == up)
        if not last:
            s4 = K.sdf.sdf_from_pe(K.sdf.pe_of_rays(o, d, z_new, 0, m), R * m)
            L.check(lib.nero_merge_sorted(_p(z), T, n, _p(tab), n_in, _p(z_new), m, _p(s4), 4, R, _p(index), st))
        else:
            L.check(lib.nero_merge_sorted(_p(z), T, n, _p(None), 0, _p(z_new), m, _p(None), 0, R, _p(index), st))
        if trace is not None:
            trace.append(dict(z=z_before, sdf=sdf_before, weights=w_out, z_new=z_new.clone(), inds=inds, index=index,
                              z_out=z[:, :n + m].clone()))
        n += m
    L.check(lib.nero_background_z(_p(far), _p(rand_bg), R, nb, _p(z), T, n_in, st))
    return z


# ----------------------------------------------------------------------------------------------------------------------
# render_core as one autograd node
# ----------------------------------------------------------------------------------------------------------------------
class RenderCore(torch.autograd.Function):
    """inputs: (meta dict, o, d, z_vals, variance, FG_LUT, *effective weights).  outputs: ray_rgb [R,3],
    gradient_error [N_in], occ_prob [N_in] (unclamped)."""

    @staticmethod
    def forward(ctx, meta, o, d, z_vals, variance, lut, poses, *params):
        dev = o.device
        lib = L.lib
        st = _st()
        K = meta.get('K')                   # packed once per step by the caller (sampler and render share it)
        if K is None:
            K = ShapeKernels(unflatten_effective(meta['names'], [p.detach() for p in params]), meta['shader_cfg'], dev).pack()
        R, T = z_vals.shape
        f32 = dict(dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        pts4 = torch.empty((R * T, 4), **f32)
        ray_counts, ray_off, counts = torch.empty(R, **i32), torch.empty(R, **i32), torch.empty(2, **i32)
        L.check(lib.nero_render_prep(_p(o), _p(d), _p(z_vals), R, T, _p(pts4), _p(ray_counts), _p(ray_off), _p(counts), st))
        n_in, n_out = (int(v) for v in counts.cpu())                 # the one host sync of the step
        rpi, rpo = row_pad(n_in), row_pad(n_out)
        inner_idx, outer_idx = torch.empty(max(n_in, 1), **i32), torch.empty(max(n_out, 1), **i32)
        L.check(lib.nero_compact(_p(pts4), _p(ray_off), R, T, _p(inner_idx), _p(outer_idx), st))
        alphaRT = torch.zeros(R * T, **f32)
        colorRT = torch.zeros((R * T, 3), **f32)
        S = {'K': K, 'R': R, 'T': T, 'pts4': pts4, 'n_in': n_in, 'n_out': n_out, 'inner_idx': inner_idx, 'outer_idx': outer_idx,
             'meta': {k: v for k, v in meta.items() if k != '_state'},   # a copy: meta['_state'] = S below must not close a cycle
             'o': o, 'd': d, 'variance': variance, 'lut': lut}

        # ---- outer samples: NeRF++ -------------------------------------------------------------------------------
        if n_out > 0:
            pe88, pev32, dist_o = torch.empty((rpo, 88), **f32), torch.empty((rpo, 32), **f32), torch.empty(rpo, **f32)
            L.check(lib.nero_gather_outer(_p(pts4), _p(d), _p(outer_idx), T, n_out, _p(pe88), _p(pev32), _p(dist_o), st))
            trunk = K.nerf_trunk.forward(pe88, pe88, n_out)
            head = K.nerf_head.forward(trunk['saves'][7], pev32, n_out)
            alpha_o, color_o = torch.empty(rpo, **f32), torch.empty((rpo, 3), **f32)
            L.check(lib.nero_nerf_head_fwd(_p(trunk['heads'][8]), _p(head['heads'][2]), _p(dist_o), n_out, _p(alpha_o), _p(color_o), st))
            L.check(lib.nero_scatter_samples(_p(alpha_o), _p(color_o), _p(outer_idx), n_out, _p(alphaRT), _p(colorRT), st))
            S.update(pe88=pe88, pev32=pev32, dist_o=dist_o, trunk=trunk, head=head)

        # ---- inner samples: SDF + split-sum shader ---------------------------------------------------------------
        gerr = torch.zeros(max(n_in, 1), **f32)
        occ_prob = torch.zeros(max(n_in, 1), **f32)
        if n_in > 0:
            x4, pe40 = torch.empty((rpi, 4), **f32), torch.empty((rpi, 40), **f32)
            L.check(lib.nero_gather_inner(_p(pts4), _p(inner_idx), n_in, _p(x4), _p(pe40), st))
            sctx = K.sdf.forward_normal(x4, n_in, pe40 if K.sdf.default_pe else None)     # (another sdf_freq: nero_encode_pe inside)
            alpha_i, geo = torch.empty(rpi, **f32), torch.empty((rpi, 8), **f32)
            L.check(lib.nero_sdf_alpha_fwd(_p(sctx['sdf4']), _p(sctx['normal']), _p(x4), _p(inner_idx), _p(d), T, _p(variance),
                                           C.c_float(meta['anneal']), n_in, _p(alpha_i), _p(geo), _p(gerr), st))
            x8 = torch.zeros((rpi, 8), **f32)
            x8[:, :3] = x4[:, :3]
            feat = sctx['feat']
            mats = [c.forward(feat, x8, n_in) for c in K.mat]
            mat = torch.empty((rpi, 8), **f32)
            Xo2 = torch.empty((2 * rpi, K.ld_outer), **f32)   # rows [0,rpi): IDE(n,1) ; rows [rpi,2rpi): IDE(refl, rough)
            Xi, Xo = torch.empty((rpi, 128), **f32), torch.empty((rpi, 96), **f32)
            L.check(lib.nero_shade_encode(_p(x4), _p(geo), _p(mats[0]['heads'][3]), _p(mats[1]['heads'][3]), _p(mats[2]['heads'][3]),
                                          n_in, _p(mat), _p(Xo2[:rpi]), _p(Xo2[rpi:]), _p(Xi), _p(Xo), K.sphere, st))
            if K.pos_freq != 8:
                # shader_config.light_pos_freq other than the YAMLs' 8: the encoder kernel writes [PE-8(p) | IDE] and [PE-8(p) | PE-6(refl)];
                # the position part is re-encoded at the requested frequency count, the direction parts are kept (network/field.py:515,
                # 556-571: pos_enc only ever sees the detached points)
                Xi, Xo = K.recode_positions(x4, n_in, Xi, Xo)
            f_out = K.outer_light.forward(Xo2, None, rpi + n_in)
            f_in = K.inner_light.forward(Xi, None, n_in)
            f_w = K.inner_weight.forward(Xo, None, n_in)
            f_h = Xh = hmask = None
            if K.human:
                Xh, hmask = torch.empty((rpi, 24), **f32), torch.empty(rpi, **f32)
                L.check(lib.nero_human_encode(_p(x4), _p(geo), _p(mat), _p(inner_idx), T, _p(poses), n_in, _p(Xh), _p(hmask), st))
                f_h = K.human_light.forward(Xh, None, n_in)
            Lh = f_out['heads'][3]
            color_i = torch.empty((rpi, 3), **f32)
            L.check(lib.nero_shade_combine_fwd(_p(geo), _p(mat), _p(Lh[:rpi]), _p(Lh[rpi:]), _p(f_in['heads'][3]), _p(f_w['heads'][3]),
                                               _p(lut), C.c_float(meta['exp_max']), n_in, _p(color_i), _p(occ_prob),
                                               _p(f_h['heads'][3] if f_h else None), _p(hmask), st))
            L.check(lib.nero_scatter_samples(_p(alpha_i), _p(color_i), _p(inner_idx), n_in, _p(alphaRT), _p(colorRT), st))
            S.update(x4=x4, x8=x8, sctx=sctx, geo=geo, mats=mats, mat=mat, Xo2=Xo2, Xi=Xi, Xo=Xo, f_out=f_out, f_in=f_in,
                     f_w=f_w, f_h=f_h, Xh=Xh, hmask=hmask, poses=poses)
        weights, rgb = torch.empty((R, T), **f32), torch.empty((R, 3), **f32)
        L.check(lib.nero_composite_fwd(_p(alphaRT), _p(colorRT), R, T, _p(weights), _p(rgb), st))
        S.update(alphaRT=alphaRT, colorRT=colorRT, weights=weights)
        ctx.S = S
        ctx.n_params = len(params)
        meta['_state'] = S                       # lets the caller reach intermediates (occ loss, validation extras)
        return rgb, gerr[:max(n_in, 0)] if n_in > 0 else gerr[:0], occ_prob[:n_in] if n_in > 0 else occ_prob[:0]

    @staticmethod
    def backward(ctx, d_rgb, d_gerr, d_occ):
        S = ctx.S
        K, R, T, n_in, n_out, meta = S['K'], S['R'], S['T'], S['n_in'], S['n_out'], S['meta']
        dev = S['o'].device
        lib = L.lib
        st = _st()
        f32 = dict(dtype=torch.float32, device=dev)
        d_rgb = d_rgb.contiguous()
        d_aRT, d_cRT = torch.empty(R * T, **f32), torch.empty((R * T, 3), **f32)
        L.check(lib.nero_composite_bwd(_p(S['alphaRT']), _p(S['colorRT']), _p(S['weights']), _p(d_rgb), R, T, _p(d_aRT), _p(d_cRT), st))
        ws = torch.empty(L.lib.nero_dw_workspace_floats(max(n_in + row_pad(n_in), n_out, 1)), **f32)
        G = {}                                   # name -> gradient
        # fused trainer: destinations inside the flat gradient bucket -- the weight-gradient GEMMs write there directly and the
        # corresponding autograd outputs are None (no AccumulateGrad add kernel, no temporary)
        gv = meta.get('grad_views') or {}
        inplace = set()                          # names whose gradient a GEMM wrote straight into its bucket view (explicit, not inferred)

        def outs_of(prefix, idxs):
            o = {i: (gv[f'{prefix}.{i}.weight'], gv[f'{prefix}.{i}.bias']) for i in idxs if f'{prefix}.{i}.weight' in gv}
            inplace.update(f'{prefix}.{i}.{k}' for i in o for k in ('weight', 'bias'))
            return o

        def put_pred(prefix, gr):
            for i in range(3):
                G[f'{prefix}.{i}.weight'], G[f'{prefix}.{i}.bias'] = gr[i]['dW'], gr[i]['db']
            G[f'{prefix}.3.weight'], G[f'{prefix}.3.bias'] = gr[3]['dWh'], gr[3]['dbh']

        if n_out > 0:
            rpo = row_pad(n_out)
            d_ao, d_co = torch.empty(rpo, **f32), torch.empty((rpo, 3), **f32)
            L.check(lib.nero_gather_sample_grads(_p(d_aRT), _p(d_cRT), _p(S['outer_idx']), n_out, _p(d_ao), _p(d_co), st))
            trunk, head = S['trunk'], S['head']
            d_sig4, d_rgb4 = torch.empty((rpo, 4), **f32), torch.empty((rpo, 4), **f32)
            L.check(lib.nero_nerf_head_bwd(_p(trunk['heads'][8]), _p(head['heads'][2]), _p(S['dist_o']), n_out, _p(d_ao), _p(d_co),
                                           _p(d_sig4), _p(d_rgb4), st))
            hb = K.nerf_head.backward(head, n_out, head_dys={2: d_rgb4}, need_dinit=True)
            head_outs = {i: (gv[f'nerf.{n}.weight'], gv[f'nerf.{n}.bias']) for i, n in ((0, 'feature'), (1, 'views')) if f'nerf.{n}.weight' in gv}
            inplace.update(f'nerf.{n}.{k}' for i, n in ((0, 'feature'), (1, 'views')) if i in head_outs for k in ('weight', 'bias'))
            hg = K.nerf_head.weight_grads(head, hb, n_out, trunk['saves'][7], S['pev32'], head_dys={2: d_rgb4}, workspace=ws, outs=head_outs)
            G['nerf.feature.weight'], G['nerf.feature.bias'] = hg[0]['dW'], hg[0]['db']
            G['nerf.views.weight'], G['nerf.views.bias'] = hg[1]['dW'], hg[1]['db']
            G['nerf.rgb.weight'], G['nerf.rgb.bias'] = hg[2]['dWh'], hg[2]['dbh']
            tb = K.nerf_trunk.backward(trunk, n_out, dy=hb['d_init'], head_dys={8: d_sig4})
            tg = K.nerf_trunk.weight_grads(trunk, tb, n_out, S['pe88'], S['pe88'], head_dys={8: d_sig4}, workspace=ws, outs=outs_of('nerf.pts', range(8)))
            for i in range(8):
                G[f'nerf.pts.{i}.weight'], G[f'nerf.pts.{i}.bias'] = tg[i]['dW'], tg[i]['db']
            G['nerf.alpha.weight'], G['nerf.alpha.bias'] = tg[8]['dWh'], tg[8]['dbh']

        d_var = None
        if n_in > 0:
            rpi = row_pad(n_in)
            d_ai, d_ci = torch.empty(rpi, **f32), torch.empty((rpi, 3), **f32)
            L.check(lib.nero_gather_sample_grads(_p(d_aRT), _p(d_cRT), _p(S['inner_idx']), n_in, _p(d_ai), _p(d_ci), st))
            geo, mat, f_out, f_in, f_w = S['geo'], S['mat'], S['f_out'], S['f_in'], S['f_w']
            Lh = f_out['heads'][3]
            dLh = torch.empty((2 * rpi, 4), **f32)
            dLi, dLo = torch.empty((rpi, 4), **f32), torch.empty((rpi, 4), **f32)
            dmat, d_geo = torch.empty((rpi, 8), **f32), torch.zeros((rpi, 8), **f32)
            d_occ_c = d_occ.contiguous() if d_occ is not None else None
            f_h = S['f_h']
            dLhum = torch.empty((rpi, 4), **f32) if f_h else None
            L.check(lib.nero_shade_combine_bwd(_p(geo), _p(mat), _p(Lh[:rpi]), _p(Lh[rpi:]), _p(f_in['heads'][3]), _p(f_w['heads'][3]),
                                               _p(S['lut']), C.c_float(meta['exp_max']), n_in, _p(d_ci), _p(d_occ_c),
                                               _p(dLh[:rpi]), _p(dLh[rpi:]), _p(dLi), _p(dLo), _p(dmat), _p(d_geo),
                                               _p(f_h['heads'][3] if f_h else None), _p(S['hmask']), _p(dLhum), st))
            n2 = rpi + n_in
            ob = K.outer_light.backward(f_out, n2, head_dys={3: dLh}, need_dinit=True)
            put_pred('outer_light', K.outer_light.weight_grads(f_out, ob, n2, S['Xo2'], None, head_dys={3: dLh}, workspace=ws, outs=outs_of('outer_light', range(3))))
            ib = K.inner_light.backward(f_in, n_in, head_dys={3: dLi}, need_dinit=True)
            put_pred('inner_light', K.inner_light.weight_grads(f_in, ib, n_in, S['Xi'], None, head_dys={3: dLi}, workspace=ws, outs=outs_of('inner_light', range(3))))
            wb = K.inner_weight.backward(f_w, n_in, head_dys={3: dLo})
            put_pred('inner_weight', K.inner_weight.weight_grads(f_w, wb, n_in, S['Xo'], None, head_dys={3: dLo}, workspace=ws, outs=outs_of('inner_weight', range(3))))
            extra = None
            if f_h:
                hb = K.human_light.backward(f_h, n_in, head_dys={3: dLhum}, need_dinit=True)
                put_pred('human_light_predictor', K.human_light.weight_grads(f_h, hb, n_in, S['Xh'], None, head_dys={3: dLhum}, workspace=ws,
                                                                         outs=outs_of('human_light_predictor', range(3))))
                extra = torch.empty((rpi, 4), **f32)
                L.check(lib.nero_human_encode_bwd(_p(S['x4']), _p(geo), _p(mat), _p(S['inner_idx']), T, _p(S['poses']), n_in,
                                                  _p(hb['d_init']), _p(extra), st))
            dmr, drr, dar = (torch.empty((rpi, 4), **f32) for _ in range(3))
            dX = ob['d_init']
            dXi = ib['d_init']
            if K.pos_freq != 8:                      # the IDE columns of d Xi back at 51..122, where nero_shade_encode_bwd reads them
                dXi128 = torch.zeros((rpi, 128), **f32)
                dXi128[:, 51:123] = dXi[:, K.pos_dim:K.pos_dim + 72]
                dXi = dXi128
            L.check(lib.nero_shade_encode_bwd(_p(geo), _p(mat), _p(dX[:rpi]), _p(dX[rpi:]), _p(dXi), _p(dmat), n_in,
                                              _p(d_geo), _p(dmr), _p(drr), _p(dar), _p(extra), _p(S['x4']), K.sphere, st))
            d_feat = torch.empty((rpi, 256), **f32)
            feat = S['sctx']['feat']
            for j, (c, name, dh) in enumerate(zip(K.mat, ('metallic_predictor', 'roughness_predictor', 'albedo_predictor'), (dmr, drr, dar))):
                mb = c.backward(S['mats'][j], n_in, head_dys={3: dh}, need_dinit=True, dinit_out=d_feat, accumulate_dinit=(j > 0))
                put_pred(name, c.weight_grads(S['mats'][j], mb, n_in, feat, S['x8'], head_dys={3: dh}, workspace=ws, outs=outs_of(name, range(3))))
            d_sdf4, d_grad, dinv = torch.empty((rpi, 4), **f32), torch.empty((rpi, 3), **f32), torch.empty(rpi, **f32)
            d_gerr_c = d_gerr.contiguous() if d_gerr is not None else None
            L.check(lib.nero_sdf_alpha_bwd(_p(S['sctx']['sdf4']), _p(S['sctx']['normal']), _p(S['x4']), _p(S['inner_idx']), _p(S['d']), T,
                                           _p(S['variance']), C.c_float(meta['anneal']), n_in, _p(d_ai), _p(d_gerr_c), _p(d_geo),
                                           _p(d_sdf4), _p(d_grad), _p(dinv), st))
            sg = K.sdf.backward(S['sctx'], d_sdf4, d_feat, d_grad, workspace=ws, outs=outs_of('sdf', range(K.sdf.last)))
            for l in range(K.sdf.n_lin):
                G[f'sdf.{l}.weight'], G[f'sdf.{l}.bias'] = sg[l]
            if not meta['freeze_inv_s']:
                v = S['variance'].detach()
                inv_s = torch.exp(v * 10.0)
                live = ((inv_s >= 1e-6) & (inv_s <= 1e6)).to(torch.float32)
                d_var = dinv[:n_in].sum() * 10.0 * inv_s * live
        grads = []
        for name, shape in zip(meta['names'], meta['shapes']):
            g, v = G.get(name), gv.get(name)
            if name in inplace:
                # the GEMM OVERWROTE the bucket view: handing a tensor to autograd as well would add it on top
                assert g is None or g.data_ptr() == v.data_ptr(), f'{name}: written in place AND returned'
                grads.append(None)
            elif g is None and v is not None:
                grads.append(None)               # no gradient this step: the bucket was zeroed
            else:
                grads.append(g if g is not None else torch.zeros(shape, **f32))
        ctx.S = None
        return (None, None, None, None, d_var, None, None) + tuple(grads)


# ----------------------------------------------------------------------------------------------------------------------
# plain SDF value with first-order weight gradients (InitSDFRegLoss inputs, network/renderer.py:591-594)
# ----------------------------------------------------------------------------------------------------------------------
class SDFValue(torch.autograd.Function):
    """sdf(x) for x [n,3] as an autograd node w.r.t. the 2 * (sdf_n_layers + 1) effective SDF weights / biases (no gradient w.r.t. x)."""

    @staticmethod
    def forward(ctx, K, x, *params):
        from .sdf import encode_pe
        n = x.shape[0]
        pe = encode_pe(x.contiguous(), n, 3, K.sdf.n_freq, K.sdf.ld_pe)
        fwd = K.sdf.value_only.forward(pe, pe, n, save=True)
        ctx.K, ctx.pe, ctx.fwd, ctx.n = K, pe, fwd, n
        ctx.shapes = [tuple(p.shape) for p in params]
        return fwd['heads'][K.sdf.last][:n, 0].clone()

    @staticmethod
    def backward(ctx, d_sdf):
        K, pe, fwd, n = ctx.K, ctx.pe, ctx.fwd, ctx.n
        ch = K.sdf.value_only
        rp = row_pad(n)
        dy = torch.zeros((rp, 4), dtype=torch.float32, device=d_sdf.device)
        dy[:n, 0] = d_sdf
        last = K.sdf.last
        bwd = ch.backward(fwd, n, head_dys={last: dy})
        gr = ch.weight_grads(fwd, bwd, n, pe, pe, head_dys={last: dy})
        out = []
        for l in range(last):
            out += [gr[l]['dW'], gr[l]['db']]
        dW8 = torch.zeros(ctx.shapes[2 * last], dtype=torch.float32, device=d_sdf.device)
        db8 = torch.zeros(ctx.shapes[2 * last + 1], dtype=torch.float32, device=d_sdf.device)
        dW8[0:1] = gr[last]['dWh']
        db8[0:1] = gr[last]['dbh']
        out += [dW8, db8]
        return (None, None) + tuple(out)


def secondary_occlusion(K, o, dr, variance, sn0, sn1):
    """get_intersection (network/field.py:454-484) for points o [P,3] inside the unit sphere and unit directions dr [P,3]:
    sum of the sn1-1 section weights of the importance-resampled march = occlusion probability [P]"""
    dev = o.device
    lib, st = L.lib, _st()
    f32 = dict(dtype=torch.float32, device=dev)
    Pn = o.shape[0]
    z = torch.empty((Pn, sn0), **f32)
    L.check(lib.nero_occ_z(_p(o), _p(dr), Pn, sn0, _p(z), st))
    s4 = K.sdf.sdf_from_pe(K.sdf.pe_of_rays(o, dr, z, 0, sn0), Pn * sn0)
    w = torch.empty((Pn, sn0 - 1), **f32)
    L.check(lib.nero_section_weights(_p(z), _p(s4), 4, sn0, _p(variance), Pn, _p(w), _p(None), st))
    z_new = torch.empty((Pn, sn1), **f32)
    L.check(lib.nero_sample_pdf(_p(z), sn0, _p(w), sn0 - 1, sn0, sn1, Pn, _p(z_new), _p(None), st))
    s4b = K.sdf.sdf_from_pe(K.sdf.pe_of_rays(o, dr, z_new, 0, sn1), Pn * sn1)
    gt = torch.empty(Pn, **f32)
    L.check(lib.nero_section_weights(_p(z_new), _p(s4b), 4, sn1, _p(variance), Pn, _p(None), _p(gt), st))
    return gt


INTER_KEYS = (('specular_albedo', 0, 3), ('specular_ref', 3, 6), ('specular_light', 6, 9), ('specular_color', 9, 12),
              ('diffuse_albedo', 12, 15), ('diffuse_light', 15, 18), ('diffuse_color', 18, 21), ('metallic', 21, 22),
              ('roughness', 22, 23), ('occ_prob', 23, 24), ('indirect_light', 24, 27))


def validation_info(K, cfg, shader_cfg, lut, variance, o, d, z_vals, weights, poses):
    """compute_validation_info (network/renderer.py:465-482): expected depth, normal map, shader intermediates at the depth point
    and the marched occlusion probability (sn0=128, sn1=9).  No grad."""
    dev = o.device
    lib, st = L.lib, _st()
    f32 = dict(dtype=torch.float32, device=dev)
    R = o.shape[0]
    depth = torch.sum(weights * z_vals, -1, keepdim=True)
    pts = (depth * d + o).contiguous()
    inner = (torch.norm(pts, dim=-1, keepdim=True) <= 1.0).float()
    rp = row_pad(R)
    x4 = torch.zeros((rp, 4), **f32)
    x4[:R, :3] = pts
    sctx = K.sdf.forward_normal(x4, R)
    grad = sctx['normal']
    out = {'depth': depth, 'normal': ((torch.nn.functional.normalize(grad, dim=-1) + 1.0) * 0.5) * inner}
    idx = torch.arange(R, dtype=torch.int32, device=dev)            # sample k belongs to ray k (T = 1)
    alpha, geo, gerr = torch.empty(rp, **f32), torch.empty((rp, 8), **f32), torch.empty(rp, **f32)
    L.check(lib.nero_sdf_alpha_fwd(_p(sctx['sdf4']), _p(grad), _p(x4), _p(idx), _p(d), 1, _p(variance), C.c_float(0.0), R,
                                   _p(alpha), _p(geo), _p(gerr), st))
    x8 = torch.zeros((rp, 8), **f32)
    x8[:, :3] = x4[:, :3]
    mats = [c.forward(sctx['feat'], x8, R, save=False) for c in K.mat]
    mat = torch.empty((rp, 8), **f32)
    Xo2, Xi, Xo = torch.empty((2 * rp, K.ld_outer), **f32), torch.empty((rp, 128), **f32), torch.empty((rp, 96), **f32)
    L.check(lib.nero_shade_encode(_p(x4), _p(geo), _p(mats[0]['heads'][3]), _p(mats[1]['heads'][3]), _p(mats[2]['heads'][3]), R,
                                  _p(mat), _p(Xo2[:rp]), _p(Xo2[rp:]), _p(Xi), _p(Xo), K.sphere, st))
    if K.pos_freq != 8:                                # shader_config.light_pos_freq != 8: the chains are packed for PE-f positions
        Xi, Xo = K.recode_positions(x4, R, Xi, Xo)     # (as RenderCore.forward does; ADVICE r5: validation fed PE-8 columns)
    Lh2 = K.outer_light.forward(Xo2, None, rp + R, save=False)['heads'][3]
    Li = K.inner_light.forward(Xi, None, R, save=False)['heads'][3]
    Lo = K.inner_weight.forward(Xo, None, R, save=False)['heads'][3]
    Lhum = hmask = None
    if K.human:
        Xh, hmask = torch.empty((rp, 24), **f32), torch.empty(rp, **f32)
        L.check(lib.nero_human_encode(_p(x4), _p(geo), _p(mat), _p(idx), 1, _p(poses), R, _p(Xh), _p(hmask), st))
        Lhum = K.human_light.forward(Xh, None, R, save=False)['heads'][3]
    rec = torch.empty((R, 32), **f32)
    L.check(lib.nero_shade_inter_results(_p(geo), _p(mat), _p(Lh2[:rp]), _p(Lh2[rp:]), _p(Li), _p(Lo), _p(lut),
                                         C.c_float(shader_cfg['light_exp_max']), R, _p(Lhum), _p(hmask), _p(rec), st))
    for k, a, b in INTER_KEYS:
        out[k] = rec[:, a:b] * inner
    if K.human:
        out['human_light'] = rec[:, 27:30] * inner
    # marched occlusion along the reflected ray; points at |p| >= 0.999 report 0 (field.py:464-470)
    inside = torch.norm(pts, dim=-1) < 0.999
    gt = torch.zeros((R, 1), **f32)
    sel = torch.nonzero(inside)[:, 0]
    if sel.numel() > 0:
        gt[sel, 0] = secondary_occlusion(K, pts[sel].contiguous(), geo[sel, 4:7].contiguous(), variance, 128, 9)
    out['occ_prob_gt'] = gt
    return out


class OccL1(torch.autograd.Function):
    """F.l1_loss(occ_prob[cand], gt) (network/renderer.py:546-547) on the fixed-capacity candidate list of nero_occ_select (unused slots: -1),
    mean over the kept count (a device scalar): nero_occ_l1 / nero_occ_l1_backward, two launches where the tensor expression and its autograd
    nodes were about twenty (round 6: a 512-ray drop-in step is dominated by such launches).  Gradient w.r.t. occ_prob only."""

    @staticmethod
    def forward(ctx, occ_prob, cand, counts, gt):
        occ_prob, gt = occ_prob.contiguous(), gt.contiguous().reshape(-1)
        loss = torch.empty(1, dtype=torch.float32, device=occ_prob.device)
        L.check(L.lib.nero_occ_l1(_p(occ_prob), _p(cand), _p(counts), _p(gt), cand.numel(), _p(loss), _st()))
        ctx.save_for_backward(occ_prob, cand, counts, gt)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, d_loss):
        occ_prob, cand, counts, gt = ctx.saved_tensors
        d = d_loss.contiguous().reshape(1).float()
        d_occ = torch.empty_like(occ_prob)
        L.check(L.lib.nero_occ_l1_backward(_p(d), _p(occ_prob), _p(cand), _p(counts), _p(gt), cand.numel(), occ_prob.numel(), _p(d_occ), _st()))
        return d_occ, None, None, None


def occ_loss(S, occ_prob, cfg, variance, occ_keys=None):
    """compute_occ_loss (network/renderer.py:522-548): surface subset -> march the reflected ray to the unit sphere
    (64 uniform + 16 importance z, no grad) -> L1(occ_prob, sum of section weights)."""
    K, n_in, T = S['K'], S['n_in'], S['T']
    dev = occ_prob.device
    lib, st = L.lib, _st()
    f32 = dict(dtype=torch.float32, device=dev)
    flag = torch.empty(n_in, dtype=torch.uint8, device=dev)
    sctx = S['sctx']
    L.check(lib.nero_occ_candidates(_p(S['x4']), _p(sctx['sdf4']), _p(sctx['normal']), _p(S['inner_idx']), _p(S['d']), T,
                                    C.c_float(cfg['occ_sdf_thresh']), n_in, _p(flag), st))
    cap = int(cfg['occ_loss_max_pn'])
    if 1 <= cap <= 4096 and os.environ.get('NERO_OCC_DEVICE', '1') != '0':
        # round 6: the candidate subset chosen ON THE DEVICE (nero_occ_select: what the fused trainer's glue has done since round 4) -- no
        # torch.nonzero, i.e. no host read-back in the middle of the step: at the reference's own batch that stall was 1 ms of a 6 ms
        # drop-in step (scripts/r06/dropin_profile.py).  The march covers the fixed capacity `cap`; unused slots carry index -1 and weight 0;
        # the mean divides by the kept count as a device scalar.  Same subset rule as below -- argsort(keys[:total], stable)[:cap], sorted --
        # with keys drawn for every inner sample instead of for the candidates (not seed-comparable with the tensor path; tests pass keys).
        keys = occ_keys.to(dev).contiguous().float() if occ_keys is not None else torch.rand(n_in, **f32)
        if keys.numel() < n_in:
            keys = torch.cat([keys, torch.full((n_in - keys.numel(),), float('inf'), **f32)])
        cand = torch.empty(cap, dtype=torch.int32, device=dev)
        counts = torch.zeros(2, dtype=torch.int32, device=dev)
        lib.nero_occ_select_workspace.restype = C.c_size_t
        ws = torch.empty(lib.nero_occ_select_workspace(n_in), dtype=torch.uint8, device=dev)
        L.check(lib.nero_occ_select(_p(flag), n_in, _p(keys), cap, _p(cand), _p(counts), C.c_void_p(ws.data_ptr()), C.c_size_t(ws.numel()), st))
        pts, dirs = torch.empty((cap, 3), **f32), torch.empty((cap, 3), **f32)
        L.check(lib.nero_occ_gather(_p(S['x4']), _p(S['geo']), _p(cand), cap, _p(pts), _p(dirs), st))
        gt = secondary_occlusion(K, pts, dirs, variance, 64, 16)
        return OccL1.apply(occ_prob, cand, counts, gt), counts[0]
    cand = torch.nonzero(flag)[:, 0]
    Pn = cand.numel()
    if Pn > cfg['occ_loss_max_pn']:
        keys = occ_keys[:Pn].to(dev) if occ_keys is not None else torch.rand(Pn, device=dev)
        keep = torch.sort(torch.argsort(keys, stable=True)[:cfg['occ_loss_max_pn']])[0]
        cand = cand[keep]
        Pn = cand.numel()
    if Pn == 0:
        return torch.zeros(1, device=dev), 0
    gt = secondary_occlusion(K, S['x4'][cand, :3].contiguous(), S['geo'][cand, 4:7].contiguous(), variance, 64, 16)
    return torch.nn.functional.l1_loss(occ_prob[cand], gt), Pn
