"""row-owner forward kernel (mlp_f16r.hip) vs the default organisation on the SDF value-only chain at the sampler's launch sizes and on the
synthetic 8 x 256 chain: ms per launch, us per 64-row layer-tile per CU.  python scripts/r06/bench_rowowner.py"""
import sys, time, math
sys.path.insert(0, '.')
import torch
from nero_amd import chain as CH, _lib as L
from nero_amd.chain import Chain, Dense, row_pad
from nero_amd.sdf import SDFField
g = torch.Generator().manual_seed(4)
dims = [39] + [256] * 8 + [257]
eff = []
for l in range(9):
    n_out = dims[l + 1] - (39 if l + 1 == 4 else 0)
    eff.append(((torch.randn(n_out, dims[l], generator=g) * 1.2 / math.sqrt(dims[l])).cuda(), (torch.randn(n_out, generator=g) * 0.05).cuda()))
f = SDFField(eff).pack()
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n
for n in (4096 * 16, 4096 * 64, 4096 * 128):
    pe = torch.randn(row_pad(n), 40, device='cuda'); pe[:, 39] = 0
    res = {}
    for name, ro, pa in (('512-thread', 0, 0), ('paired', 0, 3), ('row-owner', 3, 0)):
        CH.f16_rowowner(ro); CH.f16_paired(pa)
        res[name] = timeit(lambda: f.sdf_from_pe(pe, n))
    lt = row_pad(n) / 64 * 8.4 / 256          # 64-row layer-tiles per CU (8 full layers + the 39-wide first ~ 8.4)
    print(f'SDF value chain {n:7d} rows: ' + '  '.join(f'{k} {v*1e3:.3f} ms ({v*1e6/lt:.2f} us/layer-tile)' for k, v in res.items()))
N = 524288
x = torch.randn(row_pad(N), 256, device='cuda') * 0.1
def mk(n_out, n_in, s=1.0): return ((torch.randn(n_out, n_in, generator=g) * s / math.sqrt(n_in)).cuda(), (torch.randn(n_out, generator=g) * 0.01).cuda())
Ws = [mk(256, 256, 1.4) for _ in range(8)]
for name, act in (('relu', L.ACT_RELU), ('softplus', L.ACT_SOFTPLUS100)):
    ch = Chain([(Dense(W, b, act, 256), None) for W, b in Ws[:7]] + [(Dense(Ws[7][0], Ws[7][1], L.ACT_NONE, 256), None)], k_init=256).pack()
    res = {}
    for nm, ro, pa in (('512-thread', 0, 0), ('paired', 0, 3), ('row-owner', 3, 0)):
        CH.f16_rowowner(ro); CH.f16_paired(pa)
        res[nm] = timeit(lambda: ch.forward(x, None, N, save=False), 5)
    print(f'8 x 256 {name:8s} chain, {N} rows, no save: ' + '  '.join(f'{k} {v*1e3:.3f} ms ({v*1e6/(N/64*8/256):.2f} us/layer-tile)' for k, v in res.items()))
# the same chain with ALL-ZERO weights, biases and inputs (no operand toggling in the matrix pipe): how much of the time is the data?
Wz = [(torch.zeros(256, 256, device='cuda'), torch.zeros(256, device='cuda')) for _ in range(8)]
xz = torch.zeros_like(x)
ch = Chain([(Dense(W, b, L.ACT_RELU, 256), None) for W, b in Wz[:7]] + [(Dense(Wz[7][0], Wz[7][1], L.ACT_NONE, 256), None)], k_init=256).pack()
res = {}
for nm, ro, pa in (('512-thread', 0, 0), ('paired', 0, 3), ('row-owner', 3, 0)):
    CH.f16_rowowner(ro); CH.f16_paired(pa)
    res[nm] = timeit(lambda: ch.forward(xz, None, N, save=False), 5)
print(f'8 x 256 relu chain, ALL ZERO operands, {N} rows: ' + '  '.join(f'{k} {v*1e3:.3f} ms ({v*1e6/(N/64*8/256):.2f} us/layer-tile)' for k, v in res.items()))
