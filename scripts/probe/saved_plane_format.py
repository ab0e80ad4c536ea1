"""CPU study for a possible next format of the SAVED activations: what happens to the gradients if every hidden activation
is kept as the two block-scaled fp16 planes the GEMMs already use (hi + lo, 22 bits under a per-row power-of-two scale) instead of fp32 --
so that the weight-gradient GEMM and the reverse walk could read operand-ready planes (no conversion, no transposition) at the same 4
bytes per element.  The oracle (torch, CPU) runs one Stage-I step three times on the same rays: fp64, fp32, and fp32 with every
activation replaced by its two-plane value (straight-through in the backward, which over-states the effect: the forward moves too).
Prints, per parameter tensor, the error against fp64 of the two fp32 runs.   usage: python scripts/probe/saved_plane_format.py [rays]"""
import os
import sys
import types

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import nero_oracle as O                                     # noqa: E402  (a study script: test infrastructure, not the product)
from tests.test_parity_at_size import BELL, _shape_case                 # noqa: E402
from nero_amd.synthetic import synthetic_rays                           # noqa: E402


class TwoPlanes(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        if x.dtype != torch.float32:
            return x
        m = x.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)
        s = torch.exp2(torch.ceil(torch.log2(m)))
        y = x / s
        hi = y.half().float()
        lo = ((y - hi) * 2048.0).half().float() / 2048.0
        return (hi + lo) * s

    @staticmethod
    def backward(ctx, g):
        return g


def run(net, cfg, rays, dtype, quant):
    o, d, poses, gt = rays
    f = lambda a: a.to(dtype)
    sd = {k: v for k, v in net.to(dtype).named_parameters()}
    sd.update({k: v for k, v in net.named_buffers()})
    P = O.effective_params(sd)
    c = {**O.DEFAULT_CFG, **cfg}
    sp, rl = O.softplus100, F.relu
    if quant:
        O.softplus100 = lambda x: TwoPlanes.apply(sp(x))
        O.F = types.SimpleNamespace(**{k: getattr(F, k) for k in dir(F) if not k.startswith('__')})
        O.F.relu = lambda x: TwoPlanes.apply(rl(x))
    try:
        near, far = O.near_far_from_sphere(f(o), f(d))
        g = torch.Generator().manual_seed(0)
        R = o.shape[0]
        with torch.no_grad():
            z = O.sample_ray(P, c, f(o), f(d), near, far, rand1=torch.rand(R, 1, generator=g).to(dtype), rand_bg=torch.rand(R, 32, generator=g).to(dtype))
        return z
    finally:
        O.softplus100, O.F = sp, F


def grads(net, cfg, rays, z, dtype, quant):
    o, d, poses, gt = rays
    f = lambda a: a.to(dtype)
    net = net.to(dtype)
    for p in net.parameters():
        p.grad = None
    sd = {k: v for k, v in net.named_parameters()}
    sd.update({k: v for k, v in net.named_buffers()})
    P = O.effective_params(sd)
    c = {**O.DEFAULT_CFG, **cfg}
    sp, rl = O.softplus100, F.relu
    if quant:
        O.softplus100 = lambda x: TwoPlanes.apply(sp(x))
        O.F = types.SimpleNamespace(**{k: getattr(F, k) for k in dir(F) if not k.startswith('__')})
        O.F.relu = lambda x: TwoPlanes.apply(rl(x))
    try:
        step = 25000
        oo = O.render_core(P, c, f(o), f(d), z.to(dtype), f(poses), O.anneal(c, step), step, torch.rand(o.shape[0] * 160, generator=torch.Generator().manual_seed(1)))
        loss = O.training_loss(c, oo, f(gt), step)
        loss.backward()
    finally:
        O.softplus100, O.F = sp, F
    return float(loss), {k: p.grad.detach().double().clone() for k, p in net.named_parameters() if p.grad is not None}


def main():
    torch.set_num_threads(16)
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    net = _shape_case(BELL, 0.5)
    rays = synthetic_rays(R, seed=1)
    z = run(net, BELL, rays, torch.float32, False)                     # one set of samples for all three runs
    l64, g64 = grads(net, BELL, rays, z, torch.float64, False)
    net = _shape_case(BELL, 0.5)
    l32, g32 = grads(net, BELL, rays, z, torch.float32, False)
    net = _shape_case(BELL, 0.5)
    l32q, g32q = grads(net, BELL, rays, z, torch.float32, True)
    print(f'rays {R}: loss fp64 {l64:.9f}  fp32 {l32:.9f} ({abs(l32 - l64):.1e})  fp32 + two-plane activations {l32q:.9f} ({abs(l32q - l64):.1e})')
    rows = []
    for k in g64:
        ref = g64[k]
        sc = float(ref.abs().max())
        if sc < 1e-12:
            continue
        e32 = float((g32[k] - ref).abs().max()) / sc
        eq = float((g32q[k] - ref).abs().max()) / sc
        rows.append((k, e32, eq))
    rows.sort(key=lambda r: -r[2])
    import statistics
    print(f'{len(rows)} tensors: error against fp64 (max |dg| / max |g|): fp32 median {statistics.median(r[1] for r in rows):.2e} worst {max(r[1] for r in rows):.2e};'
          f' two-plane activations median {statistics.median(r[2] for r in rows):.2e} worst {max(r[2] for r in rows):.2e};'
          f' ratio median {statistics.median(r[2] / max(r[1], 1e-12) for r in rows):.2f} worst {max(r[2] / max(r[1], 1e-12) for r in rows):.2f}')
    for k, a, b in rows[:8]:
        print(f'   {k:48s} fp32 {a:.2e}   two-plane {b:.2e}')


if __name__ == '__main__':
    main()
