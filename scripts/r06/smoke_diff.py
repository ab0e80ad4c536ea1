"""Where does smoke()'s end-to-end error come from?  (1.57e-5 in rounds 3-4, 3.93e-5 in round 5, same golden, same oracle.)

Runs the smoke case (tests/golden/bell_s25000: 48 rays x (16+16+8)) through the HIP sampler + render and through the CPU oracle (fp32 and
fp64) on the same draws, and reports per ray: the first upsampling round whose searchsorted indices differ from the oracle's, how close the
deciding sample u sat to the cdf boundary, the z_vals difference that follows, and the ray's share of the ray_rgb error.  Run it under two
library builds (NERO_HIP_LIB=...: the default one-accumulator format and -DF16_TWO_ACC) to see which sampler decision moved.
usage (GPU box): [NERO_HIP_LIB=...] python scripts/r06/smoke_diff.py [golden-name]"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from oracle import nero_oracle as O
from tests.helpers import T, build_case_model, load_golden
from nero_amd import shape_step as SS

name = sys.argv[1] if len(sys.argv) > 1 else 'bell_s25000'
z, meta = load_golden(name)
net = build_case_model(meta).cuda()
ref = build_case_model(meta)
sd = {k: v.detach() for k, v in ref.state_dict().items()}
cfg = {**O.DEFAULT_CFG, **meta['cfg'], 'apply_occ_loss': False}


def oracle(dtype):
    P = O.effective_params({k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()})
    c = lambda k: T(z, k).to(dtype)
    tr = []
    with torch.no_grad():
        zv = O.sample_ray(P, cfg, c('o'), c('d'), c('near'), c('far'), c('rand1'), c('rand_bg'), trace=tr)
        oo = O.render_core(P, cfg, c('o'), c('d'), zv, c('human_poses'), meta['anneal'], meta['step'])
    return zv, oo['ray_rgb'], tr


z32, rgb32, tr32 = oracle(torch.float32)
z64, rgb64, tr64 = oracle(torch.float64)

names, eff, K = net._kernels()
cu = lambda k: T(z, k, 'cuda')
trace = []
with torch.no_grad():
    zg = net.sample_ray(cu('o'), cu('d'), cu('near'), cu('far'), 1.0, cu('rand1'), cu('rand_bg'), trace=trace)
out = net.render(cu('o'), cu('d'), cu('near'), cu('far'), cu('human_poses'), -1, meta['anneal'], is_train=True, step=meta['step'],
                 rand1=cu('rand1'), rand_bg=cu('rand_bg'))
rgb = out['ray_rgb'].detach().cpu()
zg = zg.cpu()

scale = float(rgb32.abs().max())
err32 = (rgb - rgb32).abs().max(dim=-1).values / scale
err64 = (rgb.double() - rgb64).abs().max(dim=-1).values / scale
o32_64 = (rgb32.double() - rgb64).abs().max(dim=-1).values / scale
print(f'case {name}: rel err of ray_rgb  HIP vs oracle-fp32 {float(err32.max()):.3e} (ray {int(err32.argmax())})   HIP vs oracle-fp64 {float(err64.max()):.3e} '
      f'(ray {int(err64.argmax())})   oracle-fp32 vs oracle-fp64 {float(o32_64.max()):.3e} (ray {int(o32_64.argmax())})')
nb = cfg['n_bg_samples']
dz32 = (zg[:, :-nb] - z32[:, :-nb]).abs().max(dim=-1).values
dz64 = (zg[:, :-nb].double() - z64[:, :-nb]).abs().max(dim=-1).values
d3264 = (z32[:, :-nb].double() - z64[:, :-nb]).abs().max(dim=-1).values
print('rays whose z_vals differ by more than 1e-5:  HIP vs fp32', [int(i) for i in torch.nonzero(dz32 > 1e-5)[:, 0]],
      ' HIP vs fp64', [int(i) for i in torch.nonzero(dz64 > 1e-5)[:, 0]], ' fp32 vs fp64', [int(i) for i in torch.nonzero(d3264 > 1e-5)[:, 0]])


def first_flip(tr_a, tr_b, key='inds'):
    """per ray: (round, sample) of the first differing searchsorted index between two traces"""
    R = tr_a[0][key].shape[0]
    res = {}
    for rnd, (a, b) in enumerate(zip(tr_a, tr_b)):
        ia, ib = a[key].cpu().long(), b[key].cpu().long()
        for r in range(R):
            if r in res:
                continue
            bad = torch.nonzero(ia[r] != ib[r])
            if len(bad):
                res[r] = (rnd, int(bad[0, 0]), int(ia[r, bad[0, 0]]), int(ib[r, bad[0, 0]]))
    return res


def margin(tr, rnd, r, j):
    """distance of the deterministic sample u_j from the nearest cdf entry of ray r in round rnd (oracle trace, fp64)"""
    w = tr[rnd]['weights'][r].double() + 1e-5
    pdf = w / w.sum()
    cdf = torch.cat([torch.zeros(1, dtype=torch.float64), torch.cumsum(pdf, 0)])
    n = tr[rnd]['z_new'].shape[1]
    u = torch.linspace(0.5 / n, 1 - 0.5 / n, n, dtype=torch.float64)[j]
    return float((cdf - u).abs().min())


if trace:
    for label, tr in (('oracle-fp32', tr32), ('oracle-fp64', tr64)):
        fl = first_flip(trace, tr)
        print(f'first searchsorted flips HIP vs {label}:', {r: v for r, v in sorted(fl.items())})
        for r, (rnd, j, a, b) in sorted(fl.items()):
            print(f'   ray {r}: round {rnd} sample {j}: HIP index {a}, {label} {b}; |u - nearest cdf| = {margin(tr64, rnd, r, j):.2e}; '
                  f'max|dz| HIP-fp32 {float(dz32[r]):.2e} HIP-fp64 {float(dz64[r]):.2e}; rgb err vs fp32 {float(err32[r]):.2e} vs fp64 {float(err64[r]):.2e}')
fl = first_flip(tr32, tr64)
print('first searchsorted flips oracle-fp32 vs oracle-fp64:', {r: v for r, v in sorted(fl.items())})
top = torch.argsort(err32, descending=True)[:5]
print('five largest per-ray errors vs oracle-fp32:', [(int(r), f'{float(err32[r]):.2e}', f'dz {float(dz32[r]):.1e}') for r in top])

# ---- the worst ray, round by round: where does its z difference enter?
r = int(err32.argmax())
print(f'--- ray {r}, round by round (HIP vs oracle-fp64):')
n = tr64[0]['z_new'].shape[1]
u = torch.linspace(0.5 / n, 1 - 0.5 / n, n, dtype=torch.float64)
for rnd in range(len(tr64)):
    a, b = trace[rnd], tr64[rnd]
    dzin = float((a['z'][r].cpu().double() - b['z'][r]).abs().max())
    dsdf = (a['sdf'][r].cpu().double() - b['sdf'][r]).abs()
    dw = (a['weights'][r].cpu().double() - b['weights'][r]).abs()
    dzn = (a['z_new'][r].cpu().double() - b['z_new'][r]).abs()
    j = int(dzn.argmax())
    w = b['weights'][r].double() + 1e-5
    cdf = torch.cat([torch.zeros(1, dtype=torch.float64), torch.cumsum(w / w.sum(), 0)])
    k = int(b['inds'][r][j])            # searchsorted(cdf, u, right=True)
    below, above = max(k - 1, 0), min(k, cdf.numel() - 1)
    denom = float(cdf[above] - cdf[below])
    print(f'   round {rnd}: max|dz_in| {dzin:.2e}  max|dsdf| {float(dsdf.max()):.2e} (|sdf| there {float(b["sdf"][r][int(dsdf.argmax())].abs()):.3e})  '
          f'max|dweight| {float(dw.max()):.2e} (weights sum {float(b["weights"][r].sum()):.3e})  max|dz_new| {float(dzn.max()):.2e} at sample {j}: '
          f'cdf interval [{below},{above}] width {denom:.3e} (reference threshold 1e-5), bin width {float(b["z"][r][above] - b["z"][r][below]):.3e}, '
          f'inv_s {b["inv_s"]:.0f}')
