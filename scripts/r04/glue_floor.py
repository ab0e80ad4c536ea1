"""What would a step cost without the torch glue between the C-driver calls?  Times ShapeTrainStep.step against the same step with the loss
assembly / occlusion-loss selection / near-far / autograd replaced by NOTHING (constant seeds: a lower bound for a fused-glue step).
usage: python scripts/r04/glue_floor.py [rays] [steps]"""
import ctypes as C
import sys, time
import torch
sys.path.insert(0, '.')
from nero_amd import _lib as L
from nero_amd import stage1
from nero_amd.stage1 import Grads, _p
import nero_amd.train as TR
from nero_amd.train import ShapeTrainStep
TR.warm_up_cos_lr = lambda step, **kw: 0.0          # learning rate 0 in BOTH arms: the weights (and with them the inner / outer sample counts) stay put
warm_up_cos_lr = TR.warm_up_cos_lr

R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
cfg = {}
ts = ShapeTrainStep(cfg, rays_per_rank=R, device='cuda', variance=0.5)
lib = stage1._lib


def timed(fn, n):
    for i in range(5):
        fn(25000 + i)
    torch.cuda.synchronize()
    t = time.time()
    for i in range(n):
        fn(25010 + i)
    torch.cuda.synchronize()
    return (time.time() - t) / n * 1e3


import os
MODE = os.environ.get('GLUE_MODE', 'both')
full = timed(ts.step, N) if MODE in ('both', 'full') else float('nan')
net, drv, fopt = ts.net, ts.drv, ts.fopt
dev = 'cuda'
f32 = dict(dtype=torch.float32, device=dev)
T = drv.T
d_rgb = torch.full((R, 3), 1e-3, **f32)
d_g = torch.full((R * T,), 1e-6, **f32)
d_o = torch.zeros(R * T, **f32)
rgb, gerr, occ = torch.empty((R, 3), **f32), torch.empty(R * T, **f32), torch.empty(R * T, **f32)
dsum = torch.zeros(1, **f32)
G = Grads()
for i in range(drv.n_lin):
    G.lin[i].W, G.lin[i].b = fopt.grad_views[fopt.names[2 * i]].data_ptr(), fopt.grad_views[fopt.names[2 * i + 1]].data_ptr()
near = torch.empty((R, 1), **f32)
far = torch.empty((R, 1), **f32)


def bare(step):
    lr = warm_up_cos_lr(step)
    ts.bucket.zero()
    o, d, gt = ts._batch()
    near_, far_ = net.near_far_from_sphere(o, d) if step < 25005 else (near, far)      # (keeps valid values in near / far)
    if step < 25005:
        near.copy_(near_); far.copy_(far_)
    fopt.reparametrise()
    drv.pack([t.detach() for t in fopt.eff])
    var = net.deviation_network.variance.detach()
    rand1, rand_bg = torch.rand((R, 1), **f32), torch.rand((R, 32), **f32)
    z = drv.sample(o, d, near, far, var, rand1, rand_bg)
    ws = drv.workspace(R)
    n_in, n_out = C.c_int(0), C.c_int(0)
    L.check(lib.nero_stage1_render_fwd(drv.h, R, _p(o), _p(d), _p(z), _p(var), _p(net.color_network.FG_LUT), None, 0.5, _p(rgb), _p(gerr), _p(occ),
                                       C.byref(n_in), C.byref(n_out), ws.data_ptr(), ws.numel(), L.stream_ptr()))
    L.check(lib.nero_stage1_render_bwd(drv.h, _p(d_rgb), _p(d_g), _p(d_o), C.byref(G), _p(dsum), L.stream_ptr()))
    fopt.step(lr, 1, absent=[net.deviation_network.variance])


b = timed(bare, N) if MODE in ('both', 'bare') else float('nan')
print(f'rays {R}: full step {full:.3f} ms, driver calls + optimiser only {b:.3f} ms (no occlusion-loss march, no loss glue): the glue costs <= {full - b:.3f} ms')
