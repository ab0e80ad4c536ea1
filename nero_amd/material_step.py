"""Stage-II (material estimation) step on the HIP library: host-side orchestration of MaterialFeatsNetwork + predictors and of
the Monte-Carlo shader (secondary rays through the BVH tracer, light MLPs on compacted hit / miss rows, microfacet estimator).
Mirrors MCShadingNetwork.forward / shade_mixed / get_lights / material_regularization (network/field.py:856-1087) and
NeROMaterialRenderer.shade / train_step (network/renderer.py:810-848)."""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib as L
from .chain import Chain, Dense, Head, row_pad
from .fields import fibonacci_az_el
from .shape_step import _p, _st, predictor_entries


GEOMETRY_TYPES = {'schlick': 0, 'ggx_smith': 1}      # include/nero_hip.h nero_mc_combine_*: geometry_type


class MaterialKernels:
    def __init__(self, eff, cfg, device='cuda'):
        if cfg['outer_light_version'] not in ('direction', 'sphere_direction'):
            raise NotImplementedError(cfg['outer_light_version'])
        if cfg['geometry_type'] not in GEOMETRY_TYPES:
            raise NotImplementedError(cfg['geometry_type'])
        self.device, self.cfg = device, cfg
        f = eff['feats']
        ent = [(Dense(W, b, L.ACT_RELU, 51 if i == 0 else 256), None) for i, (W, b) in enumerate(f[:4])]
        ent.append((Dense(f[4][0], f[4][1], L.ACT_RELU, 256, 0, 51, 256), None))
        ent += [(Dense(f[5][0], f[5][1], L.ACT_RELU, 256), None), (Dense(f[6][0], f[6][1], L.ACT_RELU, 256), None),
                (Dense(f[7][0], f[7][1], L.ACT_NONE, 256), None)]
        self.feats = Chain(ent, k_init=56, k_aux=56, aux_wide=True, device=device)
        self.mat = [Chain(predictor_entries(eff[k], 256, 3), k_init=256, k_aux=8, device=device)
                    for k in ('metallic', 'roughness', 'albedo')]
        self.sphere = int(cfg['outer_light_version'] == 'sphere_direction')
        kout = 144 if self.sphere else 72
        self.outer_light = Chain(predictor_entries(eff['outer_light'], kout), k_init=kout, device=device)
        self.inner_light = Chain(predictor_entries(eff['inner_light'], 123), k_init=128, device=device)
        self.human_light = Chain(predictor_entries(eff['human'], 24), k_init=24, device=device) if cfg['human_lights'] else None
        dn, sn = cfg['diffuse_sample_num'], cfg['specular_sample_num']

        def table(n):
            az, el = fibonacci_az_el(n)
            return torch.from_numpy(np.stack([az * 0.5 / np.pi, 1 - 2 * el / np.pi], -1).astype(np.float32)).to(device).contiguous()
        self.tab_d, self.tab_s = table(dn), table(sn)

    def pack(self):
        """(re)pack the operand images of all networks into ONE zero-filled flat buffer (kept while its size fits) with the pack jobs
        of every chain batched (as ShapeKernels.pack)"""
        from .chain import run_pack_jobs
        parts = [c for c in [self.feats, self.outer_light, self.inner_light, self.human_light] + self.mat if c is not None]
        sizes = [(c.pack_floats() + 63) // 64 * 64 for c in parts]
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


MAT_NAMES = ('metallic', 'roughness', 'albedo')


def flatten_material_effective(shader):
    """(names, tensors) of the Stage-II shader's effective weights; the weight-normed Linears through ONE batched node (fields.batched_weight_norm)"""
    from .fields import batched_weight_norm
    return batched_weight_norm(lambda: _flatten_material_effective(shader), owner=shader)


def _flatten_material_effective(shader):
    names, ts = [], []

    def add(prefix, wb):
        names.extend([prefix + '.weight', prefix + '.bias'])
        ts.extend(wb)
    for i, wb in enumerate(shader.feats_network.effective()):
        add(f'feats.{i}', wb)
    preds = ['metallic_predictor', 'roughness_predictor', 'albedo_predictor', 'outer_light', 'inner_light']
    if shader.cfg['human_lights']:
        preds.append('human_light')
    for pn in preds:
        for i, wb in enumerate(getattr(shader, pn).effective()):
            add(f'{pn}.{i}', wb)
    return names, ts


def unflatten_material_effective(names, ts):
    d = dict(zip(names, ts))

    def wb(prefix):
        return d[prefix + '.weight'], d[prefix + '.bias']
    eff = {'feats': [wb(f'feats.{i}') for i in range(8)]}
    for short, pn in (('metallic', 'metallic_predictor'), ('roughness', 'roughness_predictor'), ('albedo', 'albedo_predictor'),
                      ('outer_light', 'outer_light'), ('inner_light', 'inner_light'), ('human', 'human_light')):
        if f'{pn}.0.weight' in d:
            eff[short] = [wb(f'{pn}.{i}') for i in range(4)]
    return eff


class PredictMaterials(torch.autograd.Function):
    """raw (pre-sigmoid) metallic / roughness / albedo heads for points x [n,3] -> [n,5]   (predict_materials, field.py:915-922)"""

    @staticmethod
    def forward(ctx, K, names, gv, x, *params):
        """gv: optional {name: destination view} (fused trainer: slices of the flat gradient bucket the GEMMs write in place)"""
        dev = x.device
        n = x.shape[0]
        rp = row_pad(n)
        f32 = dict(dtype=torch.float32, device=dev)
        x = x.contiguous()
        pe = torch.empty((rp, 56), **f32)
        L.check(L.lib.nero_encode_pe(_p(x), x.stride(0), 3, 8, n, _p(pe), 56, _st()))
        x8 = torch.zeros((rp, 8), **f32)
        x8[:n, :3] = x
        ff = K.feats.forward(pe, pe, n)
        feats = ff['saves'][7]
        mf = [c.forward(feats, x8, n) for c in K.mat]
        ctx.K, ctx.names, ctx.n, ctx.pe, ctx.x8, ctx.ff, ctx.mf, ctx.gv = K, names, n, pe, x8, ff, mf, (gv or {})
        ctx.shapes = [tuple(p.shape) for p in params]
        return torch.cat([mf[0]['heads'][3][:n, :1], mf[1]['heads'][3][:n, :1], mf[2]['heads'][3][:n, :3]], -1)

    @staticmethod
    def backward(ctx, d_raw):
        K, n, pe, x8, ff, mf = ctx.K, ctx.n, ctx.pe, ctx.x8, ctx.ff, ctx.mf
        dev = d_raw.device
        rp = row_pad(n)
        f32 = dict(dtype=torch.float32, device=dev)
        ws = torch.empty(L.lib.nero_dw_workspace_floats(max(n, 1)), **f32)
        G = {}
        gv, inplace = ctx.gv, set()

        def outs_of(prefix, idxs):
            o = {i: (gv[f'{prefix}.{i}.weight'], gv[f'{prefix}.{i}.bias']) for i in idxs if f'{prefix}.{i}.weight' in gv}
            inplace.update(f'{prefix}.{i}.{k}' for i in o for k in ('weight', 'bias'))
            return o
        d_feats = torch.empty((rp, 256), **f32)
        feats = ff['saves'][7]
        cols = ((0, 1), (1, 2), (2, 5))
        for j, (c, name) in enumerate(zip(K.mat, ('metallic_predictor', 'roughness_predictor', 'albedo_predictor'))):
            dh = torch.zeros((rp, 4), **f32)
            dh[:n, :cols[j][1] - cols[j][0]] = d_raw[:, cols[j][0]:cols[j][1]]
            mb = c.backward(mf[j], n, head_dys={3: dh}, need_dinit=True, dinit_out=d_feats, accumulate_dinit=(j > 0))
            gr = c.weight_grads(mf[j], mb, n, feats, x8, head_dys={3: dh}, workspace=ws, outs=outs_of(name, range(3)))
            for i in range(3):
                G[f'{name}.{i}.weight'], G[f'{name}.{i}.bias'] = gr[i]['dW'], gr[i]['db']
            G[f'{name}.3.weight'], G[f'{name}.3.bias'] = gr[3]['dWh'], gr[3]['dbh']
        fb = K.feats.backward(ff, n, dy=d_feats)
        fg = K.feats.weight_grads(ff, fb, n, pe, pe, workspace=ws, outs=outs_of('feats', range(8)))
        for i in range(8):
            G[f'feats.{i}.weight'], G[f'feats.{i}.bias'] = fg[i]['dW'], fg[i]['db']
        grads = []
        for nm, shape in zip(ctx.names[:len(ctx.shapes)], ctx.shapes):
            if nm in inplace:
                grads.append(None)                    # written straight into the flat bucket by the weight-gradient GEMM
            else:
                g = G.get(nm)
                grads.append(g if g is not None else (None if nm in gv else torch.zeros(shape, **f32)))
        return (None, None, None, None) + tuple(grads)


class MCShade(torch.autograd.Function):
    """(pts, view, normals, mat5 = [metallic, roughness, albedo]) -> linear radiance rgb [P,3], mean diffuse light [P,3], mean
    weighted specular light [P,3] (no grad).  Gradients: mat5 and the outer / inner light MLP weights."""

    @staticmethod
    def forward(ctx, K, tracer, names, gv, pts, view, normals, mat5, rand_d, rand_s, poses, *params):
        dev = pts.device
        lib, st = L.lib, _st()
        f32 = dict(dtype=torch.float32, device=dev)
        cfg = K.cfg
        Pn = pts.shape[0]
        Dd, Ds = cfg['diffuse_sample_num'], cfg['specular_sample_num']
        D = Dd + Ds
        pts, view, normals, mat5 = (t.detach().contiguous().float() for t in (pts, view, normals, mat5))
        pt = torch.empty((Pn, 32), **f32)
        rd = rand_d.reshape(-1).contiguous() if rand_d is not None else None
        rs = rand_s.reshape(-1).contiguous() if rand_s is not None else None
        L.check(lib.nero_mc_point_setup(_p(pts), _p(view), _p(normals), _p(mat5), _p(rd), _p(rs), Pn, _p(pt), st))
        dirs, orig = torch.empty((Pn * D, 3), **f32), torch.empty((Pn * D, 3), **f32)
        L.check(lib.nero_mc_dirs(_p(pt), _p(K.tab_d), _p(K.tab_s), Pn, Dd, Ds, _p(dirs), _p(orig), st))
        # closest hit, depth >= 10 <=> miss.  A tracer that takes a launch-order hint starts the specular chunks of every point first
        # (nero_bvh_trace_grouped: same outputs); any other RayTracer-shaped object (tests, a reference-side tracer) gets the plain call
        # Round 6: rays whose estimator weight is exactly zero (GGX directions below the shading horizon under the Schlick geometry term:
        # mc_shade.hip, DEAD_SLOT) are flagged, not traversed by a tracer that takes the flags, and left out of both light MLPs
        N = Pn * D
        geom = GEOMETRY_TYPES[cfg['geometry_type']]
        dead = None
        if geom == 0 and os.environ.get('NERO_MC_SKIP_DEAD', '1') != '0':
            dead = torch.empty(N, dtype=torch.uint8, device=dev)
            L.check(lib.nero_mc_dead_rays(_p(pt), _p(dirs), Pn, Dd, Ds, geom, _p(dead), st))
        tg = getattr(tracer, 'trace_grouped', None)
        tm = getattr(tracer, 'trace_masked', None)
        from .stage2 import descending_chunks
        order = descending_chunks(D) if tm is not None else None
        if tm is not None and (dead is not None or order is not None):
            pos, fnrm, depth = tm(orig, dirs, dead, chunk_order=order)
        elif tg is not None and os.environ.get('NERO_TRACE_ORDER', 'descending') == 'grouped':
            pos, fnrm, depth = tg(orig, dirs, D, Dd)
        else:
            pos, fnrm, depth = tracer.trace(orig, dirs)
        # hit / miss split on the device (ordered compaction, nero_mc_split): the index lists torch.nonzero would give + the slot map
        depth = depth.contiguous().reshape(-1)
        i32 = dict(dtype=torch.int32, device=dev)
        slot, miss_idx, hit_idx, counts = torch.empty(N, **i32), torch.empty(N, **i32), torch.empty(N, **i32), torch.empty(3, **i32)
        tmp = torch.empty(lib.nero_mc_split_tmp_ints(N), **i32)
        poses = poses.detach().contiguous().float() if (poses is not None and K.human_light is not None) else None
        if K.human_light is not None and poses is None:
            raise ValueError('shader_cfg.human_lights needs human_poses [P,3,4]')
        # the human-light MLP's output is multiplied by the plane-hit mask of its ray (field.py:829): the miss list is partitioned by that mask
        # (nero_mc_split_classes) and the MLP runs on the miss rows [0, n_hum) only
        hum = None
        if K.human_light is not None and os.environ.get('NERO_MC_SKIP_DEAD', '1') != '0':
            hum = torch.empty(N, dtype=torch.uint8, device=dev)
            L.check(lib.nero_mc_human_flags(_p(pt), _p(dirs), _p(poses), Pn, D, _p(hum), st))
        L.check(lib.nero_mc_split_classes(_p(depth), _p(dead), _p(hum), N, _p(slot), _p(miss_idx), _p(hit_idx), _p(counts), _p(tmp), st))
        n_miss, n_hit, n_hum = (int(v) for v in counts.cpu())        # the step's host synchronisation: sizes of the light-MLP launches
        if hum is None:
            n_hum = n_miss if K.human_light is not None else 0
        from . import chain as CH
        if CH.MASK_CAPTURE is not None:                 # (tests: which rays own a row of which light MLP -- the capture's masks are per ROW)
            CH.MASK_CAPTURE.append({'kind': 'mc_split', 'slot': slot.clone(), 'depth': depth.clone(), 'n_hum': n_hum})
        rpm, rph = row_pad(n_miss), row_pad(n_hit)
        Xm, Xh = torch.empty((max(rpm, 64), 144 if K.sphere else 72), **f32), torch.empty((max(rph, 64), 128), **f32)
        fo = fi = fh = None
        outer_raw = inner_raw = human_raw = hmask = Xhum = None
        if n_miss > 0:
            L.check(lib.nero_mc_encode_miss(_p(dirs), _p(miss_idx), _p(pt), D, K.sphere, n_miss, _p(Xm), st))
            fo = K.outer_light.forward(Xm, None, n_miss)
            outer_raw = fo['heads'][3]
            if K.human_light is not None and n_hum > 0:
                rpu = max(row_pad(n_hum), 64)
                Xhum, hmask = torch.empty((rpu, 24), **f32), torch.empty(rpu, **f32)
                L.check(lib.nero_mc_human_encode(_p(dirs), _p(miss_idx), _p(pt), D, _p(poses), n_hum, _p(Xhum), _p(hmask), st))
                fh = K.human_light.forward(Xhum, None, n_hum)
                human_raw = fh['heads'][3]
        if n_hit > 0:
            L.check(lib.nero_mc_encode_hit(_p(dirs), _p(pos), _p(fnrm), _p(hit_idx), n_hit, _p(Xh), st))
            fi = K.inner_light.forward(Xh, None, n_hit)
            inner_raw = fi['heads'][3]
        rgb, dl, sl, sp = (torch.empty((Pn, 3), **f32) for _ in range(4))
        L.check(lib.nero_mc_combine_fwd_h(_p(pt), _p(dirs), _p(depth), _p(slot), _p(outer_raw), _p(inner_raw), _p(human_raw), _p(hmask), n_hum,
                                        C.c_float(cfg['light_exp_max']), C.c_float(cfg['inner_light_exp_max']), Pn, Dd, Ds,
                                        GEOMETRY_TYPES[cfg['geometry_type']], _p(rgb), _p(dl), _p(sl), _p(sp), st))
        ctx.S = dict(K=K, names=names, P=Pn, pt=pt, dirs=dirs, depth=depth, fnrm=fnrm, slot=slot, Xm=Xm, Xh=Xh, fo=fo, fi=fi,
                     fh=fh, Xhum=Xhum, hmask=hmask, poses=poses, gv=(gv or {}),
                     n_miss=n_miss, n_hit=n_hit, n_hum=n_hum, shapes=[tuple(p.shape) for p in params])
        ctx.mark_non_differentiable(sl, sp)
        return rgb, dl, sl, sp

    @staticmethod
    def backward(ctx, d_rgb, d_dl, _d_sl, _d_sp):
        S = ctx.S
        K, Pn = S['K'], S['P']
        cfg = K.cfg
        dev = d_rgb.device
        lib, st = L.lib, _st()
        f32 = dict(dtype=torch.float32, device=dev)
        Dd, Ds = cfg['diffuse_sample_num'], cfg['specular_sample_num']
        n_miss, n_hit, n_hum = S['n_miss'], S['n_hit'], S['n_hum']
        fo, fi, fh = S['fo'], S['fi'], S['fh']
        d_hr = torch.zeros((max(row_pad(n_hum), 64), 4), **f32) if fh else None
        d_or = torch.zeros((max(row_pad(n_miss), 64), 4), **f32)
        d_ir = torch.zeros((max(row_pad(n_hit), 64), 4), **f32)
        d_mat5 = torch.empty((Pn, 5), **f32)
        d_w = torch.zeros((Pn * Ds, 3), **f32)
        d_rgb_c, d_dl_c = d_rgb.contiguous(), (d_dl.contiguous() if d_dl is not None else None)     # locals: both live across the C call
        L.check(lib.nero_mc_combine_bwd_h(_p(S['pt']), _p(S['dirs']), _p(S['depth']), _p(S['slot']),
                                        _p(fo['heads'][3] if fo else None), _p(fi['heads'][3] if fi else None),
                                        _p(fh['heads'][3] if fh else None), _p(S['hmask']), n_hum,
                                        C.c_float(cfg['light_exp_max']), C.c_float(cfg['inner_light_exp_max']), Pn, Dd, Ds,
                                        GEOMETRY_TYPES[cfg['geometry_type']], _p(d_rgb_c), _p(d_dl_c),
                                        _p(d_or), _p(d_ir), _p(d_hr), _p(d_mat5), _p(d_w), st))
        ws = torch.empty(L.lib.nero_dw_workspace_floats(max(n_miss, n_hit, 1)), **f32)
        G = {}
        gv, inplace = S['gv'], set()

        def outs_of(prefix):
            o = {i: (gv[f'{prefix}.{i}.weight'], gv[f'{prefix}.{i}.bias']) for i in range(3) if f'{prefix}.{i}.weight' in gv}
            inplace.update(f'{prefix}.{i}.{k}' for i in o for k in ('weight', 'bias'))
            return o

        def put(prefix, gr):
            for i in range(3):
                G[f'{prefix}.{i}.weight'], G[f'{prefix}.{i}.bias'] = gr[i]['dW'], gr[i]['db']
            G[f'{prefix}.3.weight'], G[f'{prefix}.3.bias'] = gr[3]['dWh'], gr[3]['dbh']
        dXm = dXh = dXhum = None
        if n_miss > 0:
            ob = K.outer_light.backward(fo, n_miss, head_dys={3: d_or}, need_dinit=True)
            put('outer_light', K.outer_light.weight_grads(fo, ob, n_miss, S['Xm'], None, head_dys={3: d_or}, workspace=ws, outs=outs_of('outer_light')))
            dXm = ob['d_init']
            if fh:
                hb = K.human_light.backward(fh, n_hum, head_dys={3: d_hr}, need_dinit=True)
                put('human_light', K.human_light.weight_grads(fh, hb, n_hum, S['Xhum'], None, head_dys={3: d_hr}, workspace=ws, outs=outs_of('human_light')))
                dXhum = hb['d_init']
        if n_hit > 0:
            ib = K.inner_light.backward(fi, n_hit, head_dys={3: d_ir}, need_dinit=True)
            put('inner_light', K.inner_light.weight_grads(fi, ib, n_hit, S['Xh'], None, head_dys={3: d_ir}, workspace=ws, outs=outs_of('inner_light')))
            dXh = ib['d_init']
        L.check(lib.nero_mc_dir_bwd_h(_p(S['pt']), _p(S['dirs']), _p(S['fnrm']), _p(S['slot']), _p(K.tab_s), _p(dXm), _p(dXh), _p(d_w),
                                      Pn, Dd, Ds, _p(d_mat5), K.sphere, _p(dXhum), _p(S['poses']), n_hum, st))
        grads = []
        for nm, shape in zip(S['names'], S['shapes']):
            if nm in inplace:
                grads.append(None)                    # written straight into the flat bucket by the weight-gradient GEMM
            else:
                g = G.get(nm)
                # (no gradient this step, e.g. no light ray hit the mesh: zeros -- or nothing at all when the leaf's .grad is a
                # view of the pre-zeroed bucket)
                grads.append(g if g is not None else (None if nm in gv else torch.zeros(shape, **f32)))
        ctx.S = None
        return (None, None, None, None, None, None, None, d_mat5, None, None, None) + tuple(grads)
