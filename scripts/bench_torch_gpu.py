"""The oracle (torch port of the reference's PyTorch path: per-layer ATen GEMMs, boolean-mask gathers, autograd double backward)
run on the MI355X itself -- the 'reference single-GPU PyTorch' denominator of BASELINE.json's 10x target.  Documentation only."""
import sys, time
sys.path.insert(0, '.')
import torch
from oracle import nero_oracle as O
from nero_amd.renderer import NeROShapeRenderer
from nero_amd.synthetic import perturb_state, synthetic_rays
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
step = 5000
cfg = {'freeze_inv_s_step': 15000, 'apply_occ_loss': True, 'occ_loss_step': 20000}
torch.manual_seed(6033)
net = NeROShapeRenderer(cfg, training=False)
perturb_state(net, 0.5)
net = net.cuda()
opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True)
o, d, _, gt = synthetic_rays(R * 4, seed=1)
o, d, gt = o.cuda(), d.cuda(), gt.cuda()
c = {**O.DEFAULT_CFG, **cfg}
torch.set_default_device('cuda')
def one(i):
    s = slice((i % 4) * R, (i % 4 + 1) * R)
    opt.zero_grad(set_to_none=True)
    sd = {k: v for k, v in net.named_parameters()}; sd.update({k: v for k, v in net.named_buffers()})
    P = O.effective_params(sd)
    near, far = O.near_far_from_sphere(o[s], d[s])
    out = O.render(P, c, o[s], d[s], near, far, torch.zeros(R, 3, 4), step, O.anneal(c, step), torch.rand(R, 1), torch.rand(R, 32))
    loss = O.training_loss(c, out, gt[s], step)
    loss.backward(); opt.step()
for i in range(2): one(i)
torch.cuda.synchronize(); t = time.time(); n = 5
for i in range(n): one(i)
torch.cuda.synchronize(); dt = (time.time() - t) / n
print(f'torch-port on MI355X: {R} rays/step, {dt*1e3:.1f} ms/step, {R/dt:.0f} rays/s, peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB')
