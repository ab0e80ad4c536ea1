"""what IS the wrong value of a corrupted fwd_p_kernel element?  For corrupted (row, column) pairs of the first layer
(y = x W^T + b, 16 k-steps of 16): compare wrong - right with every k-step's contribution of that row and of the rows 16 / 32 above."""
import os, sys
sys.path.insert(0, '.')
os.environ['NERO_REPLAY_ENGINE'] = 'f16x3p'
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
    good = orig(c, init, aux, n, save)['saves'][0][:n].clone()
    CH.GEMM_MODE['fwd'] = L.GEMM_F16X3P
    d0 = c.entries[0][0]
    W = d0.W.detach().double()[:, d0.main_c0:d0.main_c0 + d0.k_main] * d0.scale
    b = d0.b.detach().double()
    X = init[:n, :d0.k_main].double()
    print('check layer 0 in fp64 against the good run: max err', float(((X[:4096] @ W.t() + b) - good[:4096].double()).abs().max()), 'act', d0.act)
    events = 0
    for k in range(80):
        out = orig(c, init, aux, n, save)['saves'][0][:n]
        dd = out - good
        idx = torch.nonzero(dd != 0)
        if idx.shape[0] == 0:
            continue
        r_, c_ = idx.tolist()[0]
        wrong = out[r_, c_]
        t0 = (r_ // 64) * 64
        same = torch.unique(out[idx[:, 0], idx[:, 1]])
        tile = good[t0:t0 + 64]
        hits = torch.nonzero(tile == wrong).tolist()
        near = torch.nonzero((good[max(0, t0 - 128):t0 + 192] == wrong)).tolist()
        print(f'run {k}: {idx.shape[0]} elements, rows {sorted(set((idx[:, 0] % 64).tolist()))[:3]}.., cols {sorted(set(idx[:, 1].tolist()))}, distinct wrong values {same.numel()} ({float(wrong):+.7f}); '
              f'same bits inside the good tile at (row, col): {hits[:6]}; in the 5 tiles around: {len(near)}; bias[c] {float(b[c_]):+.7f}')
        # the same column in the OTHER resident tiles?  (any row of the whole matrix with this exact value in column c_)
        col_hits = torch.nonzero(good[:, c_] == wrong)[:, 0].tolist()
        print('      rows of the whole matrix holding that value in the same column:', col_hits[:8], ' -> mod 64:', [x % 64 for x in col_hits[:8]], 'tile distance:', [(x // 64) - (r_ // 64) for x in col_hits[:8]])
        events += 1
        if events >= 5:
            break
