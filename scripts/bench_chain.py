"""Chain-kernel engines side by side on the shapes of the training step (run on the GPU box): 8x(256->256) chains with / without
saves, the value-only SDF chain of the sampler, the full SDF forward, a 4-layer predictor with a head.  Prints times, fp32-equivalent
TFLOP/s and the agreement of the outputs with the exact-f32 MFMA engine.  python scripts/bench_chain.py [rows] [modes]"""
import math, sys, time
import torch
sys.path.insert(0, '.')
from nero_amd import _lib as L
from nero_amd import chain as CH
from nero_amd.chain import Chain, Dense, Head, row_pad
from nero_amd.sdf import SDFField, encode_pe

g = torch.Generator().manual_seed(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
modes = sys.argv[2].split(',') if len(sys.argv) > 2 else ['f32', 'bf16x6', 'f16x3']
rp = row_pad(N)
x = torch.randn(rp, 256, device='cuda') * 0.1
def mk(n_out, n_in, s=1.0): return ((torch.randn(n_out, n_in, generator=g) * s / math.sqrt(n_in)).cuda(), (torch.randn(n_out, generator=g) * 0.01).cuda())
Ws = [mk(256, 256) for _ in range(8)]
def timeit(f, n=5):
    f(); torch.cuda.synchronize(); t = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time() - t) / n
def rel(a, b): return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))
flop = 2 * 8 * 256 * 256 * N
ref = {}
# SDF weights (geometric-init-like scale), PE input
sdf_eff = [mk(256, 39, 1.4)] + [mk(256, 256, 1.4) for _ in range(2)] + [mk(217, 256, 1.4)] + [mk(256, 256, 1.4) for _ in range(4)] + [mk(257, 256, 1.0)]
pts = (torch.rand(rp, 3, device='cuda') - 0.5) * 1.6
pred = [mk(256, 259, 1.4), mk(256, 256, 1.4), mk(256, 256, 1.4), mk(3, 256)]
x8 = torch.zeros(rp, 8, device='cuda'); x8[:, :3] = pts
for mode in modes:
    CH.set_gemm_mode(mode)
    for name, act in (('relu', L.ACT_RELU), ('softplus', L.ACT_SOFTPLUS100)):
        ch = Chain([(Dense(W, b, act, 256), None) for W, b in Ws[:7]] + [(Dense(Ws[7][0], Ws[7][1], L.ACT_NONE, 256), None)], k_init=256).pack()
        for save in (False, True):
            t = timeit(lambda: ch.forward(x, None, N, save=save))
            out = ch.forward(x, None, N, save=save)['saves'][7][:N]
            key = ('chain', name)
            if mode == 'f32': ref[key] = out.clone()
            err = rel(out, ref[key]) if key in ref else float('nan')
            print(f'{mode:7s} 8x256 {name:8s} save={int(save)}: {t*1e3:7.3f} ms {flop/t/1e12:6.1f} TF   err vs f32 {err:.2e}', flush=True)
    sdf = SDFField(sdf_eff).pack()
    pe = encode_pe(pts, N, 3, 6, 40)
    t = timeit(lambda: sdf.sdf_from_pe(pe, N))
    out = sdf.sdf_from_pe(pe, N)[:N, 0]
    if mode == 'f32': ref['sdf'] = out.clone()
    fl = 2 * 524544 * N
    print(f'{mode:7s} sdf value-only          : {t*1e3:7.3f} ms {fl/t/1e12:6.1f} TF   err vs f32 {rel(out, ref["sdf"]) if "sdf" in ref else float("nan"):.2e}', flush=True)
    t = timeit(lambda: sdf.full.forward(pe, pe, N, save=True))
    f = sdf.full.forward(pe, pe, N, save=True)
    outs = torch.cat([f['saves'][8][:N], f['heads'][8][:N, :1], f['saves'][3][:N]], 1)
    if mode == 'f32': ref['sdff'] = outs.clone()
    print(f'{mode:7s} sdf full fwd + saves    : {t*1e3:7.3f} ms {fl/t/1e12:6.1f} TF   err vs f32 {rel(outs, ref["sdff"]) if "sdff" in ref else float("nan"):.2e}', flush=True)
    pc = Chain([(Dense(pred[0][0], pred[0][1], L.ACT_RELU, 256, 0, 3, 256), None), (Dense(*pred[1], L.ACT_RELU, 256), None),
                (Dense(*pred[2], L.ACT_RELU, 256), None), (None, Head(*pred[3]))], k_init=256, k_aux=8).pack()
    t = timeit(lambda: pc.forward(x, x8, N, save=True))
    f = pc.forward(x, x8, N, save=True)
    outs = torch.cat([f['heads'][3][:N, :3], f['saves'][2][:N]], 1)
    if mode == 'f32': ref['pred'] = outs.clone()
    fl = 2 * (259 * 256 + 2 * 65536 + 768) * N
    print(f'{mode:7s} predictor fwd + saves   : {t*1e3:7.3f} ms {fl/t/1e12:6.1f} TF   err vs f32 {rel(outs, ref["pred"]) if "pred" in ref else float("nan"):.2e}', flush=True)
