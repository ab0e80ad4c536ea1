"""cProfile of the drop-in trainer path at 512 rays (host side): where do the 2 ms over the fused trainer go?"""
import cProfile, pstats, sys, time, io, os
sys.path.insert(0, '.')
import numpy as np, torch
import bench as B
from nero_amd.renderer import NeROShapeRenderer
from nero_amd.synthetic import look_at_pose, perturb_state
from nero_amd.train import warm_up_cos_lr
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
T = {}
def one(i, timed=False):
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
    if timed:
        for k, a, b in zip(('zero_grad', 'forward', 'loss', 'backward', 'opt.step'), t[:-1], t[1:]): T[k] = T.get(k, 0.0) + (b - a)
for i in range(8): one(i)
torch.cuda.synchronize(); t0 = time.time()
for i in range(20): one(8 + i, True)
torch.cuda.synchronize(); dt = (time.time() - t0) / 20
print(f'{rays} rays: {dt*1e3:.3f} ms/step wall; host time per call (ms):', {k: round(v / 20 * 1e3, 3) for k, v in T.items()}, 'host sum', round(sum(T.values()) / 20 * 1e3, 3))
if os.environ.get('NO_CPROFILE'): sys.exit(0)
pr = cProfile.Profile(); pr.enable()
for i in range(20): one(40 + i)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45); print(s.getvalue()[:9000])
