"""how many inner samples of the benchmarked Stage-I step carry a volume-rendering weight of exactly 0.0 (T underflowed behind an opaque stretch):
their colour-network rows contribute nothing to ray_rgb or to any gradient through the colour path.  python scripts/r06/zero_weight_samples.py [rays]"""
import sys
sys.path.insert(0, '.')
import torch
import bench as B
from nero_amd.renderer import NeROShapeRenderer
from nero_amd.synthetic import perturb_state, synthetic_rays
rays = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = 'cuda:0'
torch.manual_seed(6033)
net = NeROShapeRenderer({**B.BELL, 'train_ray_num': rays}, training=False)
perturb_state(net, B.VARIANCE)
net = net.to(dev)
o, d, poses_img, gt = synthetic_rays(rays, seed=1)
o, d = o.to(dev), d.to(dev)
near, far = net.near_far_from_sphere(o, d)
with torch.no_grad():
    out = net.render(o, d, near, far, None, -1, 1.0, is_train=True, step=25000)
S = out['_state']
w = S['weights']                    # [R, T]
n_in = S['n_in']
idx = S['inner_idx'][:n_in].long()
wi = w.reshape(-1)[idx]
T = w.shape[1]
print(f'rays {rays}, samples per ray {T}, inner samples {n_in} ({n_in / rays:.1f} per ray)')
for thr in (0.0, 1e-30, 1e-20, 1e-12, 1e-9, 1e-7):
    print(f'  inner samples with weight <= {thr:g}: {float((wi <= thr).float().mean()):.4f}')
print('  all samples with weight == 0:', float((w == 0).float().mean()))
occ = out['_occ_prob'].reshape(-1)
print(f'  occlusion probability of the specular query (clamped to [0, 1] before it blends indirect and direct light, field.py:572-575): '
      f'<= 0 (indirect-light MLP row exactly dead): {float((occ <= 0).float().mean()):.4f}; >= 1 (direct-light row of the specular query dead): {float((occ >= 1).float().mean()):.4f}')
