"""Bit-reproducibility of ONE forward chain launch: records the NeRF++ head chain (256 -> 256 -> 128 (+27 aux) -> 3, ~390 k rows) of a
4096-ray render, takes the 512-thread engine's result as the reference and replays the chain 60 times on the engine named in
the forward engine, counting the launches whose first saved activation differs.
It was the reproducer of the round-3 fault of the two-workgroups-per-CU kernel (packed fp32 beside MFMAs) and stays as a determinism probe.  usage: python scripts/replay_fwd_chain.py"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from nero_amd import chain as CH, _lib as L
from nero_amd.renderer import NeROShapeRenderer
from nero_amd.synthetic import perturb_state, synthetic_rays
torch.manual_seed(5)
net = NeROShapeRenderer({'apply_occ_loss': True, 'occ_loss_step': 20000}, training=False)
perturb_state(net, 0.4)
net = net.cuda()
R = 4096
o, d, poses, gt = synthetic_rays(R, seed=1)
o, d = o.cuda(), d.cuda()
near, far = net.near_far_from_sphere(o, d)
rec = []
orig = CH.Chain.forward
def fwd(self, init, aux, n_rows, save=True):
    out = orig(self, init, aux, n_rows, save)
    rec.append((self, None if init is None else init.clone(), None if aux is None else aux.clone(), n_rows, save))
    return out
with torch.no_grad():
    kern = net._kernels()
    z = net.sample_ray(o, d, near, far, 0, None, None, kern[2])
    CH.Chain.forward = fwd
    net.render_core(o, d, z, None, 0.0, step=19999, is_train=True, _kern=kern)
    CH.Chain.forward = orig
    c, init, aux, n, save = rec[1]
    CH.GEMM_MODE['fwd'] = L.GEMM_F16X3
    o_ = orig(c, init, aux, n, save)
    good = o_['saves'][0][:n].clone()
    import os
    eng = 'f16x3'
    CH.GEMM_MODE['fwd'] = L.GEMM_F16X3
    nbad, total_bad_elems, col_hist, row_hist = 0, 0, {}, {}
    N = 60
    for k in range(N):
        o_ = orig(c, init, aux, n, save)
        dd = (o_['saves'][0][:n] - good).abs()
        if float(dd.max()) > 0:
            nbad += 1
            idx = torch.nonzero(dd > 0)
            total_bad_elems += idx.shape[0]
            for r_, c_ in idx.tolist()[:64]:
                row_hist[r_ % 64] = row_hist.get(r_ % 64, 0) + 1
                col_hist[c_ % 32] = col_hist.get(c_ % 32, 0) + 1
    print(eng, 'launches differing from the f16x3 result:', nbad, 'of', N, 'bad elements', total_bad_elems, 'rows mod 64', sorted(row_hist.items()), 'cols mod 32', sorted(col_hist.items()))
