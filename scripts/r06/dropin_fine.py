"""host-side time of the drop-in training step (512 rays) by part, including the autograd nodes' backward bodies (cProfile does not see the
engine's device thread): wrappers with perf_counter accumulators around the Python bodies and around the C calls inside them"""
import sys, time, os
sys.path.insert(0, '.')
import numpy as np, torch
import bench as B
from nero_amd import stage1, wn_fused, shape_step, renderer as RR, fields, _lib as L
from nero_amd.renderer import NeROShapeRenderer
from nero_amd.synthetic import look_at_pose, perturb_state
from nero_amd.train import warm_up_cos_lr
T, ON = {}, [False]
def timed(name, f):
    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            if ON[0]: T[name] = T.get(name, 0.0) + time.perf_counter() - t
    return g
WRAP = {'nero_stage1_render_fwd', 'nero_stage1_render_bwd', 'nero_stage1_sample', 'nero_stage1_pack', 'nero_wn_forward_batch', 'nero_wn_backward_batch',
        'nero_occ_select', 'nero_occ_gather', 'nero_stage1_sdf_from_pe', 'nero_ray_points_pe', 'nero_stage1_get_state'}
class LibProxy:
    def __init__(self, lib): self._lib = lib; self._c = {}
    def __getattr__(self, n):
        if n not in WRAP:
            return getattr(self._lib, n)
        if n not in self._c:
            self._c[n] = timed('C:' + n, getattr(self._lib, n))
        return self._c[n]
stage1._lib = LibProxy(stage1._lib)
L.lib = LibProxy(L.lib)
for cls, nm in ((stage1.RenderCoreC, 'RenderCoreC'), (wn_fused._WeightNormBatch, 'WnBatch')):
    cls.forward = staticmethod(timed(nm + '.forward', cls.forward))
    cls.backward = staticmethod(timed(nm + '.backward', cls.backward))
for mod, names in ((shape_step, ('occ_loss', 'secondary_occlusion', '_flatten_effective')), (wn_fused, ('weight_norm_batch',)), (fields, ('batched_weight_norm',))):
    for n in names:
        setattr(mod, n, timed(n, getattr(mod, n)))
for n in ('_process_ray_batch', '_train_driver', 'render', 'render_core', 'compute_rgb_loss', 'sample_ray'):
    if hasattr(NeROShapeRenderer, n): setattr(NeROShapeRenderer, n, timed(n, getattr(NeROShapeRenderer, n)))
stage1.Stage1Driver.pack = timed('drv.pack', stage1.Stage1Driver.pack)
stage1.Stage1Driver.sample = timed('drv.sample', stage1.Stage1Driver.sample)
dev = 'cuda:0'
rays = int(sys.argv[1]) if len(sys.argv) > 1 else 512
torch.manual_seed(6033)
net = NeROShapeRenderer({**B.BELL, 'train_ray_num': rays}, training=False)
perturb_state(net, B.VARIANCE)
net = net.to(dev)
rg = np.random.default_rng(0)
n_img, res = 8, 512
az, el = rg.uniform(0, 2 * np.pi, n_img), rg.uniform(0.15, 1.2, n_img)
cams = np.stack([np.cos(az) * np.cos(el), np.sin(az) * np.cos(el), np.sin(el)], -1) * 3.0
poses = torch.from_numpy(np.stack([look_at_pose(c) for c in cams], 0))
K = torch.tensor([[700.0, 0, res / 2], [0, 700.0, res / 2], [0, 0, 1]]).repeat(n_img, 1, 1)
imgs = torch.from_numpy(np.random.default_rng(2).uniform(0, 1, (n_img, res, res, 3)).astype(np.float32))
net.set_ray_pool(imgs, K, poses, device=dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)
def one(i):
    st = 25000 + i
    t = [time.perf_counter()]
    for g in opt.param_groups: g['lr'] = warm_up_cos_lr(st)
    opt.zero_grad(); t.append(time.perf_counter())
    out = net({'step': st}); t.append(time.perf_counter())
    loss = out['loss_rgb'].mean() + (out['gradient_error'] * 0.1).mean()
    if 'loss_occ' in out: loss = loss + out['loss_occ'].mean()
    t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    if ON[0]:
        for k, a, b in zip(('zero_grad', 'forward', 'loss', 'backward', 'opt.step'), t[:-1], t[1:]): T['top:' + k] = T.get('top:' + k, 0.0) + (b - a)
for i in range(8): one(i)
torch.cuda.synchronize(); ON[0] = True; t0 = time.time(); N = 40
for i in range(N): one(8 + i)
torch.cuda.synchronize(); dt = (time.time() - t0) / N
print(f'{rays} rays: {dt*1e3:.3f} ms/step wall (with the timers on)')
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print(f'  {k:36s} {v / N * 1e3:7.3f} ms')
