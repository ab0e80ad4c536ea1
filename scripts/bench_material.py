"""Stage-II step timing on synthetic geometry (run on the GPU box): P surface points x (Dd + Ds) light directions."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from nero_amd.renderer import NeROMaterialRenderer
from nero_amd.synthetic import icosphere, synthetic_rays
P_ = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
Dd = int(sys.argv[2]) if len(sys.argv) > 2 else 128
Ds = int(sys.argv[3]) if len(sys.argv) > 3 else 128
sub = int(sys.argv[4]) if len(sys.argv) > 4 else 7
t = time.time()
v, f = icosphere(sub, 0.5, 0.2)
f = np.ascontiguousarray(f[:, ::-1])
print(f'mesh {len(f)} tris built in {time.time()-t:.1f}s')
torch.manual_seed(6033)
t = time.time()
net = NeROMaterialRenderer({'shader_cfg': dict(diffuse_sample_num=Dd, specular_sample_num=Ds, human_lights=False, outer_light_version='direction'),
                            'database_name': 'syn/bell'}, mesh=(v, f)).cuda()
print(f'renderer + BVH in {time.time()-t:.1f}s')
o, d, _, gt = synthetic_rays(8 * P_, seed=5, window=110)
o, d, gt = o.cuda(), d.cuda(), gt.cuda()
inters, normals, depth, hit = net.trace(o, d)
sel = torch.nonzero(hit)[:P_, 0]
assert sel.numel() == P_, sel.numel()
pts, view, nrm, gt = inters[sel].contiguous(), -d[sel].contiguous(), normals[sel].contiguous(), gt[sel]
opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True)
free, _ = torch.cuda.mem_get_info(); tmp = torch.empty(int(free * 0.4) // 4, device='cuda'); del tmp
def step(i):
    opt.zero_grad(set_to_none=True)
    out = net.shade_train(pts, view, nrm, None, gt, 5000 + i)
    loss = out['loss_rgb'].mean() + out['loss_mat_reg'].mean() + out['loss_diffuse_light'].mean()
    loss.backward(); opt.step()
    return float(loss) if i < 0 else None
for i in range(3): step(i)
torch.cuda.synchronize(); t = time.time(); n = 5
for i in range(n): step(i)
torch.cuda.synchronize(); dt = (time.time() - t) / n
C_mat, C_outer, C_inner = 1078272, 150272, 163328
D = Dd + Ds
flop = 2 * 3 * (2 * C_mat + D * C_outer) * P_
print(f'P={P_} D={Dd}+{Ds}: {dt*1e3:.1f} ms/step, {P_/dt:.0f} pts/s, {P_*D/dt/1e6:.1f} M light-rays/s, ~{flop/dt/1e12:.1f} TFLOP/s (hit-fraction~0 model)')
import gc
gc.collect(); gc.disable(); torch.cuda.synchronize(); m0 = torch.cuda.memory_allocated(); step(0); torch.cuda.synchronize()
print(f'live memory before / after one more step without cyclic GC: {m0/2**30:.3f} / {torch.cuda.memory_allocated()/2**30:.3f} GB')
