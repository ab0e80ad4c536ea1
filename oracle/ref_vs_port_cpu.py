"""Build container only (needs /root/reference): the shimmed, unmodified reference against the oracle port on the SAME CPU and the
same workload -- the ratio that turns bench.py's `cpu_baseline` (kind: port, the only one that can travel to the GPU box) into an
estimate of the reference's own CPU rate.    python oracle/ref_vs_port_cpu.py [rays] [steps]
TEST INFRASTRUCTURE ONLY."""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import ref_shim  # noqa: E402
from oracle import nero_oracle as O  # noqa: E402
from oracle.golden_util import perturb_state, synthetic_rays  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = {'freeze_inv_s_step': 15000, 'apply_occ_loss': True, 'occ_loss_step': 20000}
step = 25000
renderer, field = ref_shim.load_reference()
torch.manual_seed(6033)
net = renderer.NeROShapeRenderer(cfg, training=False)
perturb_state(net, 0.5)
net.train()
o, d, poses_img, gt = synthetic_rays(R, seed=1)
near, far = net.near_far_from_sphere(o, d)
hp = torch.zeros(R, 3, 4)
anneal = float(net.get_anneal_val(step))


def ref_step():
    net.zero_grad()
    out = net.render(o, d, near, far, hp, -1, anneal, is_train=True, step=step)
    loss = net.compute_rgb_loss(out['ray_rgb'], gt).mean() + (out['gradient_error'] * 0.1).mean() + out['loss_occ'].mean()
    loss.backward()


sd = {k: v for k, v in net.named_parameters()}
sd.update({k: v for k, v in net.named_buffers()})
c = {**O.DEFAULT_CFG, **cfg}
g = torch.Generator().manual_seed(3)
rand1, rand_bg, keys = torch.rand(R, 1, generator=g), torch.rand(R, 32, generator=g), torch.rand(R * 160, generator=g)


def port_step():
    for p in sd.values():
        p.grad = None
    P = O.effective_params(sd)
    out = O.render(P, c, o, d, near, far, hp, step, anneal, rand1, rand_bg, keys)
    O.training_loss(c, out, gt, step).backward()


def timeit(f):
    f()
    t = time.time()
    for _ in range(steps):
        f()
    return (time.time() - t) / steps


tr, tp = timeit(ref_step), timeit(port_step)
print(f'threads {torch.get_num_threads()}  rays {R}: reference {R / tr:.1f} rays/s ({tr:.2f} s/step), oracle port {R / tp:.1f} rays/s ({tp:.2f} s/step), port / reference = {tr / tp:.2f}')
