"""How much of the shading network's work could live-sample compaction remove on the BENCHMARKED workload?
A sample may be dropped from the colour networks only when its compositing weight w_i = alpha_i T_i AND its transmittance T_i (the factor
of d colour / d alpha_i) are provably negligible: T_i <= eps.  CPU, oracle, the bench's synthetic rays and weights (bench.py: VARIANCE 0.5).
    python scripts/r05/cull_fraction.py [rays]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import nero_oracle as O  # noqa: E402
from nero_amd.renderer import NeROShapeRenderer  # noqa: E402
from nero_amd.synthetic import perturb_state, synthetic_rays  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = {'freeze_inv_s_step': 15000, 'apply_occ_loss': False, 'occ_loss_step': 20000}
torch.manual_seed(6033)
net = NeROShapeRenderer(cfg, training=False)
perturb_state(net, 0.5)
o, d, _, gt = synthetic_rays(R, seed=1)
sd = {k: v.detach() for k, v in net.state_dict().items()}
P = O.effective_params(sd)
c = {**O.DEFAULT_CFG, **cfg}
g = torch.Generator().manual_seed(3)
near, far = O.near_far_from_sphere(o, d)
with torch.no_grad():
    out = O.render(P, c, o, d, near, far, torch.zeros(R, 3, 4), 25000, O.anneal(c, 25000), torch.rand(R, 1, generator=g), torch.rand(R, 32, generator=g))
w = out['weights']                                        # [R, T]
T_ = w.shape[1]
alpha = None
# transmittance in front of every sample: T_i = w_i / alpha_i is ill-defined where alpha = 0; recompute from the weights' suffix sum
trans = 1.0 - torch.cumsum(w, -1) + w                     # T_i ~ 1 - sum_{j<i} w_j  (up to the 1e-7 clamp)
inner = out.get('inner_mask')
if inner is None:
    pts = o[:, None] + d[:, None] * out['z_mid'][..., None] if 'z_mid' in out else None
n_in = out['gradient_error'].shape[0]
hit = (w[:, :T_ - 32].sum(-1) > 0.5)
for eps in (1e-6, 1e-5, 1e-4):
    dead = (trans <= eps)[:, :T_ - 32]
    print(f'eps {eps:g}: samples of the 128 inner-range z with T_i <= eps: {float(dead.float().mean()):.4f} of all rays x 128; rays whose inner '
          f'weights sum > 0.5: {float(hit.float().mean()):.4f}; inner samples (|p| <= 1): {n_in / R:.1f} per ray')
