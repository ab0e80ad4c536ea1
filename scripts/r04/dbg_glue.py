import os, sys, torch
sys.path.insert(0, '/root/repo')
from nero_amd.train import ShapeTrainStep
from nero_amd.renderer import NeROShapeRenderer
from nero_amd import _lib as L
from nero_amd import stage1
cfg, step = ({'shader_config': {'human_light': True}, 'rgb_loss': 'l1', 'occ_loss_max_pn': 333}, 25000)
R = 384
ts = ShapeTrainStep(cfg, rays_per_rank=R, pool_rays=4 * R, device='cuda', variance=0.5, prime_fraction=0.0, prime_passes=0)
c = ts.net.cfg
g = torch.Generator().manual_seed(11)
rands = (torch.rand(R, 1, generator=g).cuda(), torch.rand(R, c['n_bg_samples'], generator=g).cuda(), torch.rand(R * 160, generator=g).cuda())
rands = rands + tuple(ts.net.near_far_from_sphere(ts.pool["o"][:R], ts.pool["d"][:R]))
o, d = ts.pool['o'][:R].contiguous(), ts.pool['d'][:R].contiguous()
n1, f1 = NeROShapeRenderer.near_far_from_sphere(o, d)
n2, f2 = torch.empty_like(n1), torch.empty_like(f1)
L.check(stage1._lib.nero_near_far_sphere(o.data_ptr(), d.data_ptr(), R, n2.data_ptr(), f2.data_ptr(), L.stream_ptr()))
print('near bit-equal', bool((n1 == n2).all()), 'far', bool((f1 == f2).all()), float((n1 - n2).abs().max()), float((f1 - f2).abs().max()))
def run(mode):
    os.environ['NERO_STEP_GLUE'] = mode
    ts.cursor = 0
    info = ts.forward_backward(step, rands)
    torch.cuda.synchronize()
    return float(info['loss']), ts.bucket.flat.clone(), info
l_t, g_t, i_t = run('torch')
l_t2, g_t2, _ = run('torch')
l_h, g_h, i_h = run('hip')
print('loss', l_t, l_h, 'torch rerun diff', float((g_t - g_t2).abs().max()))
print('terms', i_h['loss_terms'].tolist(), 'counts', i_h.get('n_in'))
rows = []
for name, view in ts.fopt.grad_views.items():
    off = (view.data_ptr() - ts.bucket.flat.data_ptr()) // 4
    a, b = g_t[off: off + view.numel()], g_h[off: off + view.numel()]
    rows.append((float((a - b).abs().max()) / (float(a.abs().max()) + 1e-30), name, float(a.abs().max())))
for r in sorted(rows, reverse=True)[:12]:
    print('%.2e  %-40s scale %.3e' % r)
