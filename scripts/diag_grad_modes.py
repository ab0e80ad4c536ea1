"""Diagnostic (GPU box): parameter-gradient error of the HIP step against an fp64 oracle run in each GEMM arithmetic (f32 MFMA,
bf16x6, f16x3), next to fp32-torch's own error -- separates arithmetic precision from ReLU-gate flips between fp32 evaluation
orders.  usage: python scripts/diag_grad_modes.py [bear|bell] [rays]"""
import json, os, sys
sys.path.insert(0, '.')
import numpy as np, torch
import tests.test_parity_at_size as TP
from tests.helpers import named_grads, rel_err, _mlp_of
from oracle import nero_oracle as O
from nero_amd import chain as CH
from nero_amd.synthetic import synthetic_rays
from nero_amd.train import shape_training_loss

which = sys.argv[1] if len(sys.argv) > 1 else 'bear'
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cfg = {**TP.BELL, **({'shader_config': {'human_light': True}} if which == 'bear' else {})}
step = 25000
o, d, poses, gt = synthetic_rays(R, seed=1)
ref = TP._shape_case(cfg, 0.5, R)
c = {**O.DEFAULT_CFG, **cfg}
hp = ref.get_human_coordinate_poses(poses)
g = torch.Generator().manual_seed(3)
rand1, rand_bg, keys = torch.rand(R, 1, generator=g), torch.rand(R, c['n_bg_samples'], generator=g), torch.rand(R * 160, generator=g)
near, far = O.near_far_from_sphere(o, d)
with torch.no_grad():
    z_vals = O.sample_ray(O.effective_params({k: v.detach() for k, v in ref.state_dict().items()}), c, o, d, near, far, rand1, rand_bg)
TP._oracle_step(ref, cfg, o, d, z_vals, hp, gt, step, keys, torch.float32)
ref64 = TP._shape_case(cfg, 0.5, R, dtype=torch.float64)
TP._oracle_step(ref64, cfg, o, d, z_vals, hp, gt, step, keys, torch.float64)
g32, g64 = named_grads(ref), named_grads(ref64)
rows = {k: {'floor': rel_err(g32[k], g64[k]), 'gmax': float(g64[k].abs().max())} for k in g64 if float(g64[k].abs().max()) > 1e-12}
cu = lambda a: a.cuda()
for mode in ('f32', 'bf16x6', 'f16x3'):
    CH.set_gemm_mode(mode)
    net = TP._shape_case(cfg, 0.5, R, device='cuda')
    out = net.render(cu(o), cu(d), cu(near), cu(far), cu(hp), -1, O.anneal(c, step), is_train=True, step=step, z_vals=cu(z_vals), occ_keys=keys)
    shape_training_loss(net, out, cu(gt), step).backward()
    gh = named_grads(net)
    for k in rows:
        rows[k][mode] = rel_err(gh[k], g64[k])
os.makedirs('gpurun_out/diag', exist_ok=True)
json.dump(rows, open(f'gpurun_out/diag/grad_modes_{which}_{R}.json', 'w'), indent=0)
bad = sorted(rows.items(), key=lambda kv: -max(kv[1]['f16x3'], kv[1]['f32']))[:25]
print(f'{"tensor":52s} {"|g|max":>9s} {"torch32":>9s} {"hip f32":>9s} {"bf16x6":>9s} {"f16x3":>9s}')
for k, r in bad:
    print(f'{k:52s} {r["gmax"]:9.2e} {r["floor"]:9.2e} {r["f32"]:9.2e} {r["bf16x6"]:9.2e} {r["f16x3"]:9.2e}')
for m in ('floor', 'f32', 'bf16x6', 'f16x3'):
    v = np.array([r[m] for r in rows.values()])
    print(m, 'median', np.median(v), 'p90', np.quantile(v, 0.9), 'max', v.max(), '#>1e-4', int((v > 1e-4).sum()), 'of', len(v))
