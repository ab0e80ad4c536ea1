"""Batched weight normalisation for the drop-in renderers (round 6).

The reference reparametrises every Linear with nn.utils.weight_norm (network/field.py:118-119, 323-331): per step, 37 torch._weight_norm
forwards and as many backward nodes -- about 300 tiny launches and twice as many autograd-node dispatches, which at the reference's own
batch (512 rays, configs/shape/syn/bell.yaml:31) made the drop-in trainer path 35-48 % slower than the fused trainer of nero_amd.train.
Here every weight-normed Linear of a model goes through ONE autograd node: nero_wn_forward_batch (one launch, all effective weights into
one flat buffer) and nero_wn_backward_batch (one launch, all dv / dg).  torch still owns the parameters (weight_g / weight_v, the
reference's state_dict names) and the optimiser; it just sees two launches where it used to schedule hundreds.

Numerics: W = g v / ||v||_row with the row norm as a wave-level fp32 sum; torch._weight_norm reduces the same products in another order,
so the two agree to fp32 rounding (~1e-7 relative), not bit for bit -- tests/test_trainer_fusion.py holds both the values and the
gradients against torch's."""
import ctypes as C

import torch

from . import _lib as L


class _WeightNormBatch(torch.autograd.Function):
    @staticmethod
    def forward(ctx, shapes, *vg):
        n = len(shapes)
        vs, gs = vg[:n], vg[n:]
        dev = vs[0].device
        sizes = [r * c for r, c in shapes]
        w_flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        inv_flat = torch.empty(sum(r for r, _ in shapes), dtype=torch.float32, device=dev)
        jobs = (L.WnJob * n)()
        outs, ow, oi = [], 0, 0
        keep = []
        for i, ((r, c), v, g) in enumerate(zip(shapes, vs, gs)):
            v, g = v.detach().contiguous(), g.detach().contiguous().view(-1)
            keep += [v, g]
            w = w_flat[ow:ow + r * c].view(r, c)
            j = jobs[i]
            j.v, j.g, j.w_eff, j.inv_norm, j.rows, j.cols = v.data_ptr(), g.data_ptr(), w.data_ptr(), inv_flat.data_ptr() + 4 * oi, r, c
            outs.append(w)
            ow += r * c
            oi += r
        L.check(L.lib.nero_wn_forward_batch(jobs, n, L.stream_ptr()))
        ctx.shapes, ctx.inv_flat = shapes, inv_flat
        ctx.save_for_backward(*keep)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dW):
        shapes, n = ctx.shapes, len(ctx.shapes)
        saved = ctx.saved_tensors
        dev = saved[0].device
        dv_flat = torch.empty(sum(r * c for r, c in shapes), dtype=torch.float32, device=dev)      # (fresh every backward: torch may keep
        dg_flat = torch.empty(sum(r for r, _ in shapes), dtype=torch.float32, device=dev)          #  the views as the parameters' .grad)
        jobs = (L.WnGradJob * n)()
        dvs, dgs, ow, oi, keep = [], [], 0, 0, []
        for i, (r, c) in enumerate(shapes):
            v, g = saved[2 * i], saved[2 * i + 1]
            d = dW[i]
            d = torch.zeros((r, c), dtype=torch.float32, device=dev) if d is None else d.contiguous()
            keep.append(d)
            j = jobs[i]
            j.v, j.g, j.inv_norm, j.dW = v.data_ptr(), g.data_ptr(), ctx.inv_flat.data_ptr() + 4 * oi, d.data_ptr()
            j.dv, j.dg, j.rows, j.cols = dv_flat.data_ptr() + 4 * ow, dg_flat.data_ptr() + 4 * oi, r, c
            dvs.append(dv_flat[ow:ow + r * c].view(r, c))
            dgs.append(dg_flat[oi:oi + r].view(r, 1))
            ow += r * c
            oi += r
        L.check(L.lib.nero_wn_backward_batch(jobs, n, L.stream_ptr()))
        return (None,) + tuple(dvs) + tuple(dgs)


def weight_norm_batch(lins):
    """effective weights g v / ||v|| (dim 0, as nn.utils.weight_norm) of the weight-normed Linear modules `lins`, all in one autograd node
    -> list of [n_out, n_in] tensors (views of one flat buffer)"""
    if len(lins) > L.MAX_WN_JOBS:
        out = []
        for i in range(0, len(lins), L.MAX_WN_JOBS):
            out += weight_norm_batch(lins[i:i + L.MAX_WN_JOBS])
        return out
    shapes = tuple((int(l.weight_v.shape[0]), int(l.weight_v.shape[1])) for l in lins)
    for l in lins:
        assert l.weight_g.shape == (l.weight_v.shape[0], 1), 'weight_norm(dim=0) layout expected'
    return list(_WeightNormBatch.apply(shapes, *[l.weight_v for l in lins], *[l.weight_g for l in lins]))


def usable(lins):
    """the batched node needs fp32 CUDA parameters and the HIP library (the product path has no fallback arithmetic of its own: torch's
    per-Linear weight-norm -- the reference's own formulation -- stays in charge of CPU modules, e.g. state_dict round trips in tests)"""
    return bool(lins) and all(l.weight_v.is_cuda and l.weight_v.dtype == torch.float32 for l in lins)
