import math, sys
import torch, torch.nn.functional as F
sys.path.insert(0, '.')
from nero_amd import _lib as L
from nero_amd import chain as CH
from nero_amd.chain import Chain, Dense, Head, row_pad
def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))
g = torch.Generator().manual_seed(0)
def mk(n_out, n_in, s=2.0):
    return (torch.randn(n_out, n_in, generator=g) * s / math.sqrt(n_in)).cuda(), (torch.randn(n_out, generator=g) * 0.1).cuda()
n_rows, k_in = 64, 123
W0, b0 = mk(256, k_in); W1, b1 = mk(256, 256); W2, b2 = mk(256, 256); W3, b3 = mk(3, 256)
rp = row_pad(n_rows); kp = 128
X = torch.zeros(rp, kp, device='cuda'); X[:n_rows, :k_in] = torch.randn(n_rows, k_in, generator=g).cuda()
dy = torch.zeros(rp, 4, device='cuda'); dy[:n_rows, :3] = torch.randn(n_rows, 3, generator=g).cuda()
res = {}
for mode in ('f32', 'bf16x6'):
    CH.set_gemm_mode(mode)
    ch = Chain([(Dense(W0, b0, L.ACT_RELU, k_in), None), (Dense(W1, b1, L.ACT_RELU, 256), None),
                (Dense(W2, b2, L.ACT_RELU, 256), None), (None, Head(W3, b3))], k_init=kp).pack()
    fwd = ch.forward(X, None, n_rows)
    bwd = ch.backward(fwd, n_rows, head_dys={3: dy}, need_dinit=True)
    res[mode] = (fwd, bwd)
f0, b0_ = res['f32']; f1, b1_ = res['bf16x6']
for i in range(3):
    print('save', i, rel(f1['saves'][i][:n_rows], f0['saves'][i][:n_rows]))
for i in (2, 1, 0):
    d0, d1 = b0_['deltas'][i][:n_rows], b1_['deltas'][i][:n_rows]
    print('delta', i, rel(d1, d0), 'per-tile max err:', [round(float((d1[:, 32*t:32*t+32]-d0[:, 32*t:32*t+32]).abs().max()), 4) for t in range(8)])
d0, d1 = b0_['d_init'][:n_rows], b1_['d_init'][:n_rows]
print('d_init', rel(d1, d0), [round(float((d1[:, 32*t:32*t+32]-d0[:, 32*t:32*t+32]).abs().max()), 4) for t in range(4)], 'rows0-31 vs 32-63:', float((d1[:32]-d0[:32]).abs().max()), float((d1[32:]-d0[32:]).abs().max()))
d0, d1 = b0_['deltas'][1][:n_rows], b1_['deltas'][1][:n_rows]
err = (d1 - d0).abs().cpu()
bad = (err > 1e-4).nonzero()
print('bad count', len(bad), 'rows', sorted(set(bad[:, 0].tolist())), 'cols', sorted(set(bad[:, 1].tolist())))
a1 = f1['saves'][1][:n_rows].cpu(); 
for r, c in bad[:10].tolist():
    print(r, c, 'f32', float(d0[r, c]), 'split', float(d1[r, c]), 'a_prev', float(a1[r, c]))
