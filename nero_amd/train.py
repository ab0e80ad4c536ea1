"""Stage-I training step driver: ray pool resident in HBM, render + loss + backward on the HIP path, data-parallel gradient
all-reduce over RCCL, Adam.  Mirrors Trainer.run's inner loop (train/trainer.py:109-140), NeROShapeRenderer.train_step
(network/renderer.py:319-330), the loss assembly (network/loss.py) and the warm-up/cosine LR rule
(train/lr_common_manager.py:20-43)."""
import math

import numpy as np
import gc

import torch
import torch.distributed as dist  # noqa: F401

from .parallel import GradBucket, device_count_weights, parallel_forced, per_rank_occ_cap, rank_slice

from .renderer import NeROShapeRenderer
from .synthetic import perturb_state, synthetic_rays


def warm_up_cos_lr(step, total_step=300000, warm_up_end=5000, learning_rate=5e-4, learning_rate_alpha=0.05):
    if step < warm_up_end:
        f = step / warm_up_end
    else:
        prog = (step - warm_up_end) / (total_step - warm_up_end)
        f = (math.cos(math.pi * prog) + 1.0) * 0.5 * (1 - learning_rate_alpha) + learning_rate_alpha
    return f * learning_rate


def shape_training_loss(net, out, gt, step, eikonal_weight=0.1, eikonal_rank_weight=1.0, occ_rank_weight=1.0):
    """sum of the means of every `loss*` entry the reference's loss objects produce for the shape stage
    (train/trainer.py:127-137; network/loss.py: NeRFRenderLoss, EikonalLoss, OccLoss, InitSDFRegLoss)."""
    loss = net.compute_rgb_loss(out['ray_rgb'], gt).mean() + (out['gradient_error'] * eikonal_weight).mean() * eikonal_rank_weight
    if 'loss_occ' in out:
        loss = loss + out['loss_occ'].mean() * occ_rank_weight
    if step < 1000 and 'sdf_vals' in out:
        norm = torch.norm(out['sdf_pts'], dim=-1)
        sdf = out['sdf_vals']
        w = (math.cos(step / 1000 * math.pi) + 1) / 2
        sm = norm < 0.1
        if int(sm.sum()) > 0:
            sl = torch.mean(torch.clamp(sdf[sm] - (norm[sm] - 0.1), min=0.0))
            loss = loss + sl / ((sl > 1e-5).float() + 1e-3) * w
        lm = norm > 1.05
        if int(lm.sum()) > 0:
            ll = torch.clamp((norm[lm] - 1.05) - sdf[lm], min=0.0)
            loss = loss + torch.sum(ll) / (torch.sum(ll > 1e-5) + 1e-3) * w
    return loss


class FusedOptimizer:
    """Trainer-loop fusion for a model given as a list of Linear modules.  Replaces, per step, the
    torch._weight_norm launches, their backward launches, the per-parameter gradient fills and torch's multi-tensor Adam by
        nero_wn_forward_batch   (1 launch: every effective weight W = g v / ||v|| into persistent buffers the packed operand images
                                 are built from -- the chain descriptors are built ONCE, the pointers never change)
        nero_wn_adam_batch      (2 launches: weight-norm backward fused with Adam for (g, v); plain Adam for biases, plain Linear
                                 weights and extra scalars)
    The render step's autograd node hands dL/dW_eff to leaf tensors whose .grad are views of ONE flat bucket, which is also the
    all-reduce payload (dg, dv are linear in dW_eff, so reducing dW_eff is equivalent to reducing the parameter gradients).

    Adam follows torch.optim.Adam per PARAMETER: a tensor without a gradient in some step (`absent` in step(): the variance while
    step < freeze_inv_s_step) is skipped -- value, moments and its own step counter stay put -- exactly like a `.grad is None`
    parameter under torch (train/trainer.py:105-170 runs torch.optim.Adam)."""

    def __init__(self, lins, extra_plain, device, betas=(0.9, 0.999), eps=1e-8):
        import ctypes as C
        from . import _lib as L
        self.L, self.C, self.device = L, C, device
        self.betas, self.eps, self.t = betas, eps, 0
        f32 = dict(dtype=torch.float32, device=device)
        self.names, self.eff, leaves = [], [], []
        self._wn, self._plain = [], []                # (lin, w_eff, inv_norm) / plain parameters
        for name, lin in lins:
            if hasattr(lin, 'weight_g'):
                w = torch.empty(lin.weight_v.shape, **f32).requires_grad_(True)
                self._wn.append((lin, w, torch.empty(lin.weight_v.shape[0], **f32)))
                leaves.append(w)
            else:
                w = lin.weight
                self._plain.append(w)
                leaves.append(w)
            self._plain.append(lin.bias)
            leaves.append(lin.bias)
            self.names += [name + '.weight', name + '.bias']
            self.eff += [w, lin.bias]
        assert len(self._wn) <= L.MAX_WN_JOBS
        for p_ in extra_plain:
            self._plain.append(p_)
            leaves.append(p_)
        self._plain_steps = [0] * len(self._plain)    # torch.optim.Adam keeps one step counter per parameter
        self.bucket = GradBucket(leaves)              # .grad of every leaf = view of one flat buffer
        # Adam moments: (m, v) for weight_v / weight_g of every weight-normed Linear and for every plain tensor
        n_state = sum(l.weight_v.numel() + l.weight_g.numel() for l, _, _ in self._wn) + sum(p.numel() for p in self._plain)
        self.m, self.v = torch.zeros(n_state, **f32), torch.zeros(n_state, **f32)
        off = 0
        self._wn_jobs = (L.WnJob * max(1, len(self._wn)))()
        for j, (lin, w, inv) in zip(self._wn_jobs, self._wn):
            rows, cols = lin.weight_v.shape
            assert lin.weight_v.is_contiguous() and lin.weight_g.is_contiguous()
            j.v = j.v_rw = lin.weight_v.data_ptr()
            j.g = j.g_rw = lin.weight_g.data_ptr()
            j.w_eff, j.inv_norm, j.dW, j.rows, j.cols = w.data_ptr(), inv.data_ptr(), w.grad.data_ptr(), rows, cols
            j.m_v, j.v_v = self.m[off:].data_ptr(), self.v[off:].data_ptr()
            off += rows * cols
            j.m_g, j.v_g = self.m[off:].data_ptr(), self.v[off:].data_ptr()
            off += rows
        self._plain_jobs = (L.AdamJob * len(self._plain))()
        for j, p_ in zip(self._plain_jobs, self._plain):
            assert p_.is_contiguous()
            j.p, j.grad, j.m, j.v, j.n = p_.data_ptr(), p_.grad.data_ptr(), self.m[off:].data_ptr(), self.v[off:].data_ptr(), p_.numel()
            off += p_.numel()
        self.reparametrise()
        self.grad_views = {n: t.grad for n, t in zip(self.names, self.eff)}     # where the weight-gradient GEMMs write

    def reparametrise(self):
        """effective weights of every weight-normed Linear from the current (g, v): one launch"""
        L = self.L
        L.check(L.lib.nero_wn_forward_batch(self._wn_jobs, len(self._wn), L.stream_ptr()))

    def kernels(self):
        """(names, effective leaves, packed chains), repacked from the current parameters"""
        self.reparametrise()
        return self.names, self.eff, self.K.pack()

    def zero_grad(self):
        self.bucket.zero()

    def step(self, lr, world=1, absent=()):
        """all-reduce (mean) of the flat dL/dW_eff bucket, then weight-norm backward + Adam.  `absent`: parameters that received NO
        gradient this step (torch: .grad is None) -- skipped, their Adam step counters do not advance."""
        L, C = self.L, self.C
        self.bucket.all_reduce_mean(world)
        self.t += 1
        skip = {id(p_) for p_ in absent}
        for i, (j, p_) in enumerate(zip(self._plain_jobs, self._plain)):
            if id(p_) in skip:
                j.step = -1
            else:
                self._plain_steps[i] += 1
                j.step = self._plain_steps[i]
        L.check(L.lib.nero_wn_adam_batch(self._wn_jobs, len(self._wn), self._plain_jobs, len(self._plain), C.c_float(lr),
                                         C.c_float(self.betas[0]), C.c_float(self.betas[1]), C.c_float(self.eps), self.t, L.stream_ptr()))
        self._after_step()

    def _after_step(self):
        pass


class FusedShapeOptimizer(FusedOptimizer):
    """FusedOptimizer over the ten networks of NeROShapeRenderer (+ deviation_network.variance) with their packed chains."""

    def __init__(self, net, device, betas=(0.9, 0.999), eps=1e-8):
        from .shape_step import ShapeKernels, unflatten_effective
        self.net = net
        self.variance = net.deviation_network.variance
        super().__init__(self._linears(net), [self.variance], device, betas, eps)
        # the sdf last layer is consumed as [Dense(W[1:]) | Head(W[0:1])] by the kernels: views of the same leaf
        self.K = ShapeKernels(unflatten_effective(self.names, [t.detach() for t in self.eff]), net.color_network.cfg, device)

    @staticmethod
    def _linears(net):
        out = [(f'sdf.{l}', getattr(net.sdf_network, f'lin{l}')) for l in range(net.sdf_network.n_lin)]
        nf = net.outer_nerf
        out += [(f'nerf.pts.{i}', lin) for i, lin in enumerate(nf.pts_linears)]
        out += [('nerf.views', nf.views_linears[0]), ('nerf.feature', nf.feature_linear), ('nerf.alpha', nf.alpha_linear), ('nerf.rgb', nf.rgb_linear)]
        cn = net.color_network
        preds = ['metallic_predictor', 'roughness_predictor', 'albedo_predictor', 'outer_light', 'inner_light', 'inner_weight']
        if cn.cfg['human_light']:
            preds.append('human_light_predictor')
        for pn in preds:
            out += [(f'{pn}.{i}', getattr(cn, pn)[k]) for i, k in enumerate((0, 2, 4, 6))]
        return out

    def _after_step(self):
        # the parameters were updated by raw kernels: torch's version counters did not move, so the renderer's no-grad cache of
        # packed operand images (NeROShapeRenderer._kernels) must be dropped explicitly
        self.net._param_epoch = getattr(self.net, '_param_epoch', 0) + 1
        self.net._kern_cache = None


class FusedMaterialOptimizer(FusedOptimizer):
    """FusedOptimizer over MCShadingNetwork (feats network, three material predictors, outer / inner / human light MLPs):
    Stage II's ~500 torch weight-norm launches and its multi-tensor Adam become 3 launches (profiles/r02_stage2_kernel_stats.csv)."""

    def __init__(self, net, device, betas=(0.9, 0.999), eps=1e-8):
        from .material_step import MaterialKernels, unflatten_material_effective
        self.net = net
        super().__init__(self._linears(net.shader_network), [], device, betas, eps)
        self.K = MaterialKernels(unflatten_material_effective(self.names, [t.detach() for t in self.eff]), net.shader_network.cfg, device)

    @staticmethod
    def _linears(sn):
        out = [(f'feats.{i}', sn.feats_network.module0[k]) for i, k in enumerate((0, 2, 4, 6))]
        out += [(f'feats.{4 + i}', sn.feats_network.module1[k]) for i, k in enumerate((0, 2, 4, 6))]
        preds = ['metallic_predictor', 'roughness_predictor', 'albedo_predictor', 'outer_light', 'inner_light']
        if sn.cfg['human_lights']:
            preds.append('human_light')
        for pn in preds:
            out += [(f'{pn}.{i}', getattr(sn, pn)[k]) for i, k in enumerate((0, 2, 4, 6))]
        return out

    def _after_step(self):
        # same reason as FusedShapeOptimizer._after_step: NeROMaterialRenderer._kernels keys its no-grad cache on torch's version
        # counters, which raw-kernel updates do not move
        self.net._param_epoch = getattr(self.net, '_param_epoch', 0) + 1
        self.net._kern_cache = None


class ShapeTrainStep:
    """one process = one GPU.  Every rank holds the same weights and a disjoint slice of each global ray batch
    (rank-strided); gradients are summed with ONE flat all-reduce per step and divided by world size.
    CONSTRUCTION IS COLLECTIVE when world > 1: the priming passes run full forward + backward passes whose loss weights come from an
    all-reduce of the per-rank sample counts (`device_count_weights`), so every rank must construct its ShapeTrainStep with the same
    arguments at the same point of the program (what `bench.py` and the 2-rank tests do), exactly like the steps themselves."""

    def __init__(self, cfg, rays_per_rank=4096, pool_rays=262144, device='cuda', seed=6033, variance=None, eikonal_weight=0.1,
                 rank=0, world=1, prime_fraction=0.35, fused=None, prime_passes=4):
        self.device, self.rank, self.world, self.R = device, rank, world, rays_per_rank
        torch.manual_seed(seed)
        if world > 1:                  # global occlusion-loss candidate budget = the single-process cap
            cfg = {**cfg, 'occ_loss_max_pn': per_rank_occ_cap({**NeROShapeRenderer.default_cfg, **cfg}['occ_loss_max_pn'], world)}
        self.net = NeROShapeRenderer(cfg, training=False)
        if variance is not None:
            perturb_state(self.net, variance)
        self.net = self.net.to(device)
        self.params = [p for p in self.net.parameters()]
        # fused trainer loop (weight-norm + Adam kernels, persistent effective-weight buffers) on the GPU; `fused=False` keeps the
        # torch path (torch._weight_norm autograd + torch.optim.Adam), which is also what the drop-in renderer runs under an
        # external optimiser
        self.fused = (device != 'cpu') if fused is None else fused
        self.drv = None
        self._glue_obj = None
        if self.fused:
            self.fopt = FusedShapeOptimizer(self.net, device)
            self.bucket = self.fopt.bucket
            # the C-level step driver (nero_stage1_*: sampler, render forward and backward as one C call each); NERO_STEP_DRIVER=py
            # keeps the Python-sequenced launches of nero_amd/shape_step.py (identical kernels, identical results)
            import os
            from . import stage1
            if os.environ.get('NERO_STEP_DRIVER', 'c') != 'py' and stage1.supported(self.net.cfg, self.net.color_network.cfg):
                self.drv = stage1.Stage1Driver(self.net.cfg, self.net.color_network.cfg, device)
        else:
            self.bucket = GradBucket(self.params)                # p.grad = views of one flat buffer, for the whole run
            self.opt = torch.optim.Adam(self.params, lr=1e-3, fused=(device != 'cpu'))
        self.eik_w = eikonal_weight
        o, d, poses, gt = synthetic_rays(pool_rays, seed=1)
        self.pool = {'o': o.to(device), 'd': d.to(device), 'gt': gt.to(device)}
        self.human = bool(self.net.color_network.cfg['human_light'])
        if self.human:                 # per-ray human frame of the ray's image (get_human_coordinate_poses, network/renderer.py:240-256)
            self.pool['hp'] = self.net.get_human_coordinate_poses(poses.to(device))
        self.pool_n = pool_rays
        self.cursor = 0
        if device != 'cpu':
            if prime_fraction > 0:
                self.prime_allocator(prime_fraction)
            self._lazy_init()
            # size the caching allocator for the real batches: `prime_passes` full-size forward + backward passes over the first
            # batches of the pool (no optimiser step: the weights and Adam moments are untouched; the pool cursor is rewound).
            # The per-step workspaces are ~30 GB of variously sized blocks whose sizes follow the per-batch sample counts; without
            # this the first optimisation steps pay hipMalloc / block-splitting for them.
            if prime_passes > 0 and pool_rays >= rays_per_rank * world:
                for i in range(prime_passes):
                    self.forward_backward(25000 + i)
                self.bucket.zero()
                self.cursor = 0
                torch.cuda.synchronize()
            # everything built so far (modules, packed-weight caches, ray pool) is long-lived: keep it out of the cyclic GC's
            # generations, so that a periodic full collection does not stall a step walking it
            gc.collect()
            gc.freeze()

    def _lazy_init(self):
        """one 64-ray render so that one-time costs (library load, hipFuncSetAttribute, IDE table upload, kernel code objects)
        are paid at construction, not inside the first training step"""
        o, d = self.pool['o'][:64], self.pool['d'][:64]
        near, far = self.net.near_far_from_sphere(o, d)
        out = self.net.render(o, d, near, far, self.pool['hp'][:64] if self.human else None, -1, 0.5, is_train=True, step=25000,
                              **{k: v for k, v in self._render_args(25000).items() if k != '_grad_views'})
        shape_training_loss(self.net, out, self.pool['gt'][:64], 25000).backward()
        self.bucket.zero()
        torch.cuda.synchronize()

    def prime_allocator(self, fraction=0.35, cap_bytes=64 << 30):
        """Reserve one large HBM segment up front (35 % of the free memory, at most 64 GB: a 4096-ray step peaks at ~30 GB)
        and hand it to torch's caching allocator: every per-step activation / gradient workspace is then carved out of it
        instead of triggering hipMalloc (hundreds of ms for multi-GB segments) while the per-step sample counts fluctuate."""
        free, _ = torch.cuda.mem_get_info(self.device)
        n = min(int(free * fraction), cap_bytes) // 4
        t = torch.empty(n, dtype=torch.float32, device=self.device)
        del t

    def _batch(self):
        G = self.R * self.world
        if self.cursor + G > self.pool_n:
            self.cursor = 0
        s = rank_slice(self.cursor, self.R, self.rank)
        self.cursor += G
        self._hp = self.pool['hp'][s] if self.human else None
        return self.pool['o'][s], self.pool['d'][s], self.pool['gt'][s]

    def _render_args(self, step):
        """fused trainer: the effective-weight leaves, their packed operand images (C driver: nero_stage1_pack; otherwise the Python
        chains) and the bucket views the weight-gradient GEMMs write into"""
        if not self.fused:
            return {}
        if self.drv is not None and step >= 1000 and self.drv.matches_current_modes():
            self.fopt.reparametrise()
            self.drv.pack([t.detach() for t in self.fopt.eff])
            return dict(_kern=(self.fopt.names, self.fopt.eff, None), _grad_views=self.fopt.grad_views, _driver=self.drv)
        return dict(_kern=self.fopt.kernels(), _grad_views=self.fopt.grad_views)

    def forward_only(self, step):
        """one inference render of the next ray batch: the reference's is_train=False path (sampler + render forward + the
        validation extras of compute_validation_info), no loss / backward / optimiser; `step` only sets the cosine anneal"""
        net = self.net
        o, d, _ = self._batch()
        near, far = net.near_far_from_sphere(o, d)
        with torch.no_grad():
            # (a schedule step below occ_loss_step: inference does not evaluate the occlusion loss)
            out = net.render(o, d, near, far, self._hp, 0, net.get_anneal_val(step), is_train=False,
                             step=min(step, net.cfg['occ_loss_step'] - 1))
        return out['ray_rgb']

    def _glue(self, step):
        """the HIP-glued step (nero_amd.stage1.ShapeStepGlue) when the C driver runs the render: default engines, schedule step >= 1000,
        std_act 'exp', a known rgb_loss; NERO_STEP_GLUE=torch keeps the tensor glue (net.render + shape_training_loss + autograd)"""
        import os
        from . import stage1
        if not (self.fused and self.drv is not None and step >= 1000 and self.drv.matches_current_modes()
                and stage1.ShapeStepGlue.supported(self.net) and os.environ.get('NERO_STEP_GLUE', 'hip') != 'torch'):
            return None
        if self._glue_obj is None:
            self._glue_obj = stage1.ShapeStepGlue(self.net, self.drv, self.fopt.grad_views, self.fopt.names)
        return self._glue_obj

    def forward_backward(self, step, rands=None):
        """render + loss + backward of this rank's slice of the next global batch; gradients land in the flat bucket.
        rands = (rand1 [R,1], rand_bg [R,n_bg], occ_keys) injects the random draws (tests)."""
        net = self.net
        self.bucket.zero()
        o, d, gt = self._batch()
        glue = self._glue(step)
        if glue is not None:
            self.fopt.reparametrise()
            self.drv.pack([t.detach() for t in self.fopt.eff])
            c = net.cfg
            frozen = c['freeze_inv_s_step'] is not None and step < c['freeze_inv_s_step']
            # data parallel: the count weights of the eikonal / occlusion means from device-side counts, no host stall
            wfn = (lambda n_in, counts: device_count_weights([n_in, counts[0] if counts is not None else 0], self.world, self.device)) \
                if (self.world > 1 or parallel_forced()) else None
            r = glue.forward_backward(o, d, gt, self._hp, step, net.deviation_network.variance, self.eik_w, frozen, wfn, rands)
            # ('loss' is a copy: the glue's loss buffer is rewritten by the next step; 'loss_terms' -- total, rgb, eikonal, occlusion -- is that buffer)
            return {'loss': r['loss'][0].clone(), 'n_in': r['n_in'], 'n_out': r['n_out'], 'loss_terms': r['loss']}
        near, far = net.near_far_from_sphere(o, d)
        extra = {} if rands is None else dict(rand1=rands[0], rand_bg=rands[1], occ_keys=rands[2])
        out = net.render(o, d, near, far, self._hp, -1, net.get_anneal_val(step), is_train=True, step=step, **self._render_args(step), **extra)
        # data parallel: the eikonal mean runs over each rank's own inner samples and the occlusion loss over its own candidate
        # set -> weight both by their global counts so that N ranks reproduce the single-process means
        w_eik, w_occ = device_count_weights([out['_state']['n_in'], out.get('_occ_count', 0)], self.world, self.device)
        loss = shape_training_loss(net, out, gt, step, self.eik_w, w_eik, w_occ)
        loss.backward()
        st = out['_state']
        return {'loss': loss.detach(), 'n_in': st['n_in'], 'n_out': st['n_out']}

    def step(self, step):
        lr = warm_up_cos_lr(step)
        info = self.forward_backward(step)
        # deviation_network.variance receives no gradient while inv_s is frozen (network/renderer.py:494-495): torch.optim.Adam in the
        # reference skips it (.grad is None) -- value, moments and its step counter stay put until the unfreeze step
        c = self.net.cfg
        var = self.net.deviation_network.variance
        frozen = c['freeze_inv_s_step'] is not None and step < c['freeze_inv_s_step']
        if self.fused:
            self.fopt.step(lr, self.world, absent=[var] if frozen else ())     # flat all-reduce + weight-norm backward + Adam
        else:
            for g in self.opt.param_groups:
                g['lr'] = lr
            self.bucket.all_reduce_mean(self.world)
            keep = var.grad
            if frozen:
                var.grad = None
            self.opt.step()
            var.grad = keep
        return info


# ----------------------------------------------------------------------------------------------------------------------
# Stage II: material estimation (NeROMaterialRenderer.train_step, network/renderer.py:829-844; MaterialRegLoss, network/loss.py:45-55)
# ----------------------------------------------------------------------------------------------------------------------
def material_lr(step, total_step=100000, warm_up_end=1000, learning_rate=5e-4, learning_rate_alpha=0.05):
    """WarmUpCosLR with the material YAMLs' lr_cfg (configs/material/*/ *.yaml: end_warm 1000, end_iter 100000)"""
    return warm_up_cos_lr(step, total_step, warm_up_end, learning_rate, learning_rate_alpha)


def material_training_loss(shader_cfg, out, step, world=1):
    """sum of the means of the `loss*` entries (train/trainer.py:134-137): loss_rgb, loss_mat_reg, loss_diffuse_light.
    Data parallel: all three are means over a rank's P surface points -- equal shards, so the rank average of the
    gradients IS the big-batch gradient -- except the `reg_min_max` hinge of the first 2000 steps, which the reference adds as a SUM
    over the batch's points (network/field.py:1079-1084): a rank's share of that sum is weighted by `world` so that the average over
    ranks restores the global sum."""
    loss = out['loss_rgb'].mean()
    if 'loss_mat_reg' in out:
        loss = loss + out['loss_mat_reg'].mean()
    if 'loss_diffuse_light' in out:
        loss = loss + out['loss_diffuse_light'].mean()
    if world > 1 and 'loss_mat_reg' in out:
        from .renderer import material_hinge
        hinge = material_hinge(shader_cfg, out['roughness'], out['metallic'], step)
        if hinge is not None:
            loss = loss + (world - 1) * hinge
    return loss


def synthetic_surface_pool(net, n_points, device, seed=5, window=110, n_images=8):
    """Stage-II pool stand-in (no dataset in the container): camera rays of the synthetic rig traced through the renderer's mesh with
    the product tracer, hits kept -- what NeROMaterialRenderer.set_ray_pool builds from a database (network/renderer.py:756-802).
    -> dict(pts, view, normals, rgb [n,3], img_idx [n]) on `device`, poses_img [n_images,3,4]"""
    pts, view, nrm, rgb, idx = [], [], [], [], []
    got, off = 0, 0
    while got < n_points:
        o, d, poses, gt = synthetic_rays(4 * n_points, seed=seed, window=window, offset=off, n_images=n_images)
        off += 4 * n_points
        o, d = o.to(device), d.to(device)
        inters, normals, depth, hit = net.trace(o, d)
        sel = torch.nonzero(hit)[:, 0]
        pts.append(inters[sel]); view.append(-d[sel]); nrm.append(normals[sel]); rgb.append(gt.to(device)[sel])
        idx.append((torch.arange(o.shape[0], device=device) % n_images)[sel])
        got += sel.numel()
        if off > 64 * n_points:
            raise RuntimeError('synthetic_surface_pool: the camera rig hardly hits the mesh')
    _, _, poses, _ = synthetic_rays(n_images, seed=seed, n_images=n_images)
    cat = lambda xs: torch.cat(xs, 0)[:n_points].contiguous()
    return {'pts': cat(pts), 'view': cat(view), 'normals': cat(nrm), 'rgb': cat(rgb), 'img_idx': cat(idx)}, poses.to(device)


class MaterialTrainStep:
    """Stage-II data parallelism (BASELINE configs[4]): one process = one GPU; every rank holds the same weights, the
    same BVH and the same (identically shuffled) pool of surface points and takes the rank-strided slice of each global batch;
    gradients of all 104 / 112 tensors travel as ONE flat all-reduce (5.6 / 6.3 MB), then the fused weight-norm + Adam kernels."""

    def __init__(self, cfg, mesh, points_per_rank=4096, pool_points=None, device='cuda', seed=6033, rank=0, world=1, fused=None,
                 pool=None, fused_glue=None):
        from .renderer import NeROMaterialRenderer
        self.device, self.rank, self.world, self.P = device, rank, world, points_per_rank
        torch.manual_seed(seed)
        self.net = NeROMaterialRenderer(cfg, mesh=mesh)
        perturb_state(self.net, None)
        self.net = self.net.to(device)
        self.params = [p for p in self.net.parameters()]
        self.fused = (device != 'cpu') if fused is None else fused
        self.drv = None
        import os
        # NERO_LOSS_GLUE=torch keeps the tensor-op glue (NeROMaterialRenderer.shade_train) around the C-level calls
        self.fused_glue = (os.environ.get('NERO_LOSS_GLUE', 'hip') != 'torch') if fused_glue is None else fused_glue
        if self.fused:
            self.fopt = FusedMaterialOptimizer(self.net, device)
            self.bucket = self.fopt.bucket
            import os
            from . import stage2
            if os.environ.get('NERO_STEP_DRIVER', 'c') != 'py' and stage2.supported():      # the C-level shading driver (nero_stage2_*)
                self.drv = stage2.Stage2Driver(self.net.shader_network.cfg, device)
        else:
            self.bucket = GradBucket(self.params)
            self.opt = torch.optim.Adam(self.params, lr=1e-3, fused=(device != 'cpu'))
        pool_points = pool_points or 4 * points_per_rank * world
        if pool is None:
            pool, poses_img = synthetic_surface_pool(self.net, pool_points, device)
        else:
            pool, poses_img = pool
        self.pool, self.pool_n = pool, pool['pts'].shape[0]
        self.human_img = self.net.get_human_coordinate_poses(poses_img)
        self.cursor = 0

    def _batch(self):
        G = self.P * self.world
        if self.cursor + G > self.pool_n:
            self.cursor = 0
        s = rank_slice(self.cursor, self.P, self.rank)
        self.cursor += G
        return {k: v[s] for k, v in self.pool.items()}

    def forward_backward(self, step, rands=None):
        """shade + losses + backward of this rank's slice of the next global batch; gradients land in the flat bucket.
        rands: optional dict(rand_d, rand_s, reg_ang, reg_eps) for this slice (tests); default: drawn on the device"""
        net = self.net
        self.bucket.zero()
        b = self._batch()
        hp = self.human_img[b['img_idx']] if net.shader_network.cfg['human_lights'] else None
        if self.fused and self.fused_glue and self.drv is not None and self.drv.matches_current_modes():
            return self._forward_backward_fused_glue(b, hp, step, rands or {})
        if self.fused:                                # effective-weight leaves + packed chains of the fused optimiser, and the
            if self.drv is not None and self.drv.matches_current_modes():                        # bucket views the GEMMs write into
                self.fopt.reparametrise()
                self.drv.pack([t.detach() for t in self.fopt.eff])
                net._kern_override, net._grad_views, net._driver = (self.fopt.names, self.fopt.eff, None), self.fopt.grad_views, self.drv
            else:
                net._kern_override, net._grad_views = self.fopt.kernels(), self.fopt.grad_views
        try:
            out = net.shade_train(b['pts'], b['view'], b['normals'], hp, b['rgb'], step, **(rands or {}))
        finally:
            net._kern_override = net._grad_views = net._driver = None
        loss = material_training_loss(net.shader_network.cfg, out, step, self.world)
        loss.backward()
        return {'loss': loss.detach(), 'out': out}

    def _forward_backward_fused_glue(self, b, hp, step, rands):
        """the same step with the tensor glue between the C-level calls as single launches (nero_amd/csrc/mat_loss.hip): perturbed points,
        sigmoid heads, the three losses and their gradients -- ~25 launches of glue per step instead of ~240 (the reference's
        visualisation-only outputs of shade(): specular / diffuse colour, approximate light, are not formed in a training step).
        Random numbers are drawn in the order the tensor path draws them (reg_ang, reg_eps, rand_d, rand_s)."""
        from . import stage2 as S2
        net, drv = self.net, self.drv
        cfg, scfg = net.cfg, net.shader_network.cfg
        pts, view, nrm, gt = b['pts'], b['view'], b['normals'], b['rgb']
        P, dev = pts.shape[0], pts.device
        self.fopt.reparametrise()
        drv.pack([t.detach() for t in self.fopt.eff])
        names, eff, gv = self.fopt.names, self.fopt.eff, self.fopt.grad_views
        x = pts
        if cfg['reg_mat'] and scfg['reg_change']:
            ang = rands['reg_ang'] if rands.get('reg_ang') is not None else torch.rand(P, 1, device=dev)
            if scfg['change_type'] == 'constant':
                eps = scfg['change_eps']
            elif scfg['change_type'] == 'gaussian':
                eps = rands['reg_eps'] if rands.get('reg_eps') is not None else torch.normal(mean=0.0, std=scfg['change_eps'], size=[P, 1], device=dev)
            else:
                raise NotImplementedError(scfg['change_type'])
            x = S2.reg_points(pts, nrm, ang, eps)
        raw = S2.PredictMaterialsC.apply(drv, names[:40], gv, P, x, *eff[:40])
        mat = S2.MaterialHeadC.apply(raw)
        rand_d = rand_s = None
        if scfg['random_azimuth']:
            rand_d = rands['rand_d'] if rands.get('rand_d') is not None else torch.rand(P, 1, 1, device=dev)
            rand_s = rands['rand_s'] if rands.get('rand_s') is not None else torch.rand(P, 1, 1, device=dev)
        rgb_lin, dl, _sl, _sp = S2.MCShadeC.apply(drv, net.ray_tracer, names[40:], gv, pts, view, nrm, mat[:P], rand_d, rand_s, hp, *eff[40:])
        terms, rgb_pr = S2.MaterialLossC.apply(S2.loss_cfg(cfg, scfg, step, self.world), P, mat, rgb_lin, dl, gt)
        loss = terms[0]
        loss.backward()
        m = mat.detach()
        return {'loss': loss.detach(), 'loss_terms': terms.detach(),
                'out': {'rgb_pr': rgb_pr, 'rgb_gt': gt, 'metallic': m[:P, 0:1], 'roughness': m[:P, 1:2], 'albedo': m[:P, 2:5]}}

    def step(self, step):
        lr = material_lr(step)
        info = self.forward_backward(step)
        if self.fused:
            self.fopt.step(lr, self.world)
        else:
            for g in self.opt.param_groups:
                g['lr'] = lr
            self.bucket.all_reduce_mean(self.world)
            self.opt.step()
        return info
