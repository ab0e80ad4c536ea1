"""Stage-I training step driver: ray pool resident in HBM, render + loss + backward on the HIP path, data-parallel gradient
all-reduce over RCCL, Adam.  Mirrors Trainer.run's inner loop (train/trainer.py:109-140), NeROShapeRenderer.train_step
(network/renderer.py:319-330), the loss assembly (network/loss.py) and the warm-up/cosine LR rule
(train/lr_common_manager.py:20-43)."""
import math

import numpy as np
import gc

import torch
import torch.distributed as dist  # noqa: F401

from .parallel import GradBucket, global_count_weights, per_rank_occ_cap, rank_slice

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


class ShapeTrainStep:
    """one process = one GPU.  Every rank holds the same weights and a disjoint slice of each global ray batch
    (rank-strided); gradients are summed with ONE flat all-reduce per step and divided by world size."""

    def __init__(self, cfg, rays_per_rank=4096, pool_rays=262144, device='cuda', seed=6033, variance=None, eikonal_weight=0.1,
                 rank=0, world=1, prime_fraction=0.35):
        self.device, self.rank, self.world, self.R = device, rank, world, rays_per_rank
        torch.manual_seed(seed)
        if world > 1:                  # global occlusion-loss candidate budget = the single-process cap
            cfg = {**cfg, 'occ_loss_max_pn': per_rank_occ_cap({**NeROShapeRenderer.default_cfg, **cfg}['occ_loss_max_pn'], world)}
        self.net = NeROShapeRenderer(cfg, training=False)
        if variance is not None:
            perturb_state(self.net, variance)
        self.net = self.net.to(device)
        self.params = [p for p in self.net.parameters()]
        self.bucket = GradBucket(self.params)                    # p.grad = views of one flat buffer, for the whole run
        self.opt = torch.optim.Adam(self.params, lr=1e-3, fused=(device != 'cpu'))
        self.eik_w = eikonal_weight
        o, d, poses, gt = synthetic_rays(pool_rays, seed=1)
        self.pool = {'o': o.to(device), 'd': d.to(device), 'gt': gt.to(device)}
        self.pool_n = pool_rays
        self.cursor = 0
        if device != 'cpu':
            if prime_fraction > 0:
                self.prime_allocator(prime_fraction)
            self._lazy_init()
            # everything built so far (modules, packed-weight caches, ray pool) is long-lived: keep it out of the cyclic GC's
            # generations, so that a periodic full collection does not stall a step walking it
            gc.collect()
            gc.freeze()

    def _lazy_init(self):
        """one 64-ray render so that one-time costs (library load, hipFuncSetAttribute, IDE table upload, kernel code objects)
        are paid at construction, not inside the first training step"""
        o, d = self.pool['o'][:64], self.pool['d'][:64]
        near, far = self.net.near_far_from_sphere(o, d)
        out = self.net.render(o, d, near, far, None, -1, 0.5, is_train=True, step=25000)
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
        return self.pool['o'][s], self.pool['d'][s], self.pool['gt'][s]

    def forward_only(self, step):
        """one inference render of the next ray batch: the reference's is_train=False path (sampler + render forward + the
        validation extras of compute_validation_info), no loss / backward / optimiser; `step` only sets the cosine anneal"""
        net = self.net
        o, d, _ = self._batch()
        near, far = net.near_far_from_sphere(o, d)
        with torch.no_grad():
            # (a schedule step below occ_loss_step: inference does not evaluate the occlusion loss)
            out = net.render(o, d, near, far, None, 0, net.get_anneal_val(step), is_train=False,
                             step=min(step, net.cfg['occ_loss_step'] - 1))
        return out['ray_rgb']

    def forward_backward(self, step):
        """render + loss + backward of this rank's slice of the next global batch; gradients land in the flat bucket"""
        net = self.net
        self.bucket.zero()
        o, d, gt = self._batch()
        near, far = net.near_far_from_sphere(o, d)
        out = net.render(o, d, near, far, None, -1, net.get_anneal_val(step), is_train=True, step=step)
        # data parallel: the eikonal mean runs over each rank's own inner samples and the occlusion loss over its own candidate
        # set -> weight both by their global counts so that N ranks reproduce the single-process means
        w_eik, w_occ = global_count_weights([out['_state']['n_in'], out.get('_occ_count', 0)], self.world, self.device)
        loss = shape_training_loss(net, out, gt, step, self.eik_w, w_eik, w_occ)
        loss.backward()
        st = out['_state']
        return {'loss': loss.detach(), 'n_in': st['n_in'], 'n_out': st['n_out']}

    def step(self, step):
        lr = warm_up_cos_lr(step)
        for g in self.opt.param_groups:
            g['lr'] = lr
        info = self.forward_backward(step)
        self.bucket.all_reduce_mean(self.world)
        self.opt.step()
        return info
