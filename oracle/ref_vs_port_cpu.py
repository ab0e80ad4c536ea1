"""Build container only (needs /root/reference): the shimmed, unmodified reference against the oracle port on the SAME CPU and the
same workload -- the ratio that turns bench.py's `cpu_baseline` (kind: port, the only one that can travel to the GPU box) into an
estimate of the reference's own CPU rate.    python oracle/ref_vs_port_cpu.py [rays] [steps] [ns ni nb] [out.json]
Default = BASELINE.md section 3's protocol: config C1 (bell Stage I, 512 rays, n_samples = n_importance = n_bg_samples = 32, schedule step
25000), 1 warm-up + 3 timed steps, median rays/s, core count stated.  The record is committed as profiles/rNN_ref_vs_port_cpu.json and
bench.py attaches it to its `cpu_baseline` (the reference itself cannot travel to the GPU box).
TEST INFRASTRUCTURE ONLY."""
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import ref_shim  # noqa: E402
from oracle import nero_oracle as O  # noqa: E402
from oracle.golden_util import perturb_state, synthetic_rays  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ns, ni, nb = (int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (32, 32, 32)
out_json = sys.argv[6] if len(sys.argv) > 6 else None
torch.set_num_threads(os.cpu_count())
cfg = {'freeze_inv_s_step': 15000, 'apply_occ_loss': True, 'occ_loss_step': 20000, 'n_samples': ns, 'n_importance': ni, 'n_bg_samples': nb}
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
rand1, rand_bg, keys = torch.rand(R, 1, generator=g), torch.rand(R, nb, generator=g), torch.rand(R * (ns + ni + nb), generator=g)


def port_step():
    for p in sd.values():
        p.grad = None
    P = O.effective_params(sd)
    out = O.render(P, c, o, d, near, far, hp, step, anneal, rand1, rand_bg, keys)
    O.training_loss(c, out, gt, step).backward()


def timeit(f):
    f()                                            # 1 warm-up
    ts = []
    for _ in range(steps):
        t = time.time()
        f()
        ts.append(time.time() - t)
    return sorted(ts)


tr, tp = timeit(ref_step), timeit(port_step)
mr, mp = tr[len(tr) // 2], tp[len(tp) // 2]
print(f'threads {torch.get_num_threads()}  rays {R} x ({ns}+{ni}+{nb}): reference {R / mr:.1f} rays/s (median {mr:.2f} s/step, min {tr[0]:.2f}, '
      f'max {tr[-1]:.2f}), oracle port {R / mp:.1f} rays/s ({mp:.2f} s/step), port / reference = {mr / mp:.2f}')
if out_json:
    rec = {'what': 'unmodified reference (NeROShapeRenderer.render + loss + backward, network/renderer.py:608-627, under oracle/ref_shim.py) '
                   'against the oracle port (oracle/nero_oracle.py) on the same host cores and workload',
           'where': 'build container (no GPU): the reference cannot travel to the GPU box', 'cores': torch.get_num_threads(),
           'workload': f'bell Stage I, {R} rays x ({ns}+{ni}+{nb}) samples, schedule step {step} (occlusion loss on), synthetic rays seed 1',
           'protocol': f'1 warm-up + {steps} timed steps, median',
           'reference': {'rays_per_s': round(R / mr, 2), 's_per_step_median': round(mr, 3), 's_per_step_min': round(tr[0], 3), 's_per_step_max': round(tr[-1], 3)},
           'port': {'rays_per_s': round(R / mp, 2), 's_per_step_median': round(mp, 3), 's_per_step_min': round(tp[0], 3), 's_per_step_max': round(tp[-1], 3)},
           'port_over_reference': round(mr / mp, 3), 'torch': torch.__version__}
    with open(out_json, 'w') as f:
        json.dump(rec, f, indent=1)
    print('wrote', out_json)
