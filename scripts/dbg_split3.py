"""8x(256->256) chain: f32 vs bf16x6 engines, all four passes (run on the GPU box)"""
import math, os, sys, time
import torch
sys.path.insert(0, '.')
from nero_amd import _lib as L
from nero_amd import chain as CH
from nero_amd.chain import Chain, Dense, row_pad
g = torch.Generator().manual_seed(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
rp = row_pad(N)
x = torch.randn(rp, 256, device='cuda') * 0.1
dy = torch.randn(rp, 256, device='cuda')
Ws = [((torch.randn(256, 256, generator=g) / 16).cuda(), (torch.randn(256, generator=g) * 0.01).cuda()) for _ in range(8)]
ws = torch.empty(L.lib.nero_dw_workspace_floats(N), device='cuda')
def timeit(f, n=5):
    f(); torch.cuda.synchronize(); t = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time() - t) / n
flop = 2 * 8 * 256 * 256 * N
for mode in ('f32', 'bf16x6', 'f16x3'):
    CH.set_gemm_mode(mode)
    for name, act in (('relu', L.ACT_RELU), ('softplus', L.ACT_SOFTPLUS100)):
        ch = Chain([(Dense(W, b, act, 256), None) for W, b in Ws[:7]] + [(Dense(Ws[7][0], Ws[7][1], L.ACT_NONE, 256), None)], k_init=256).pack()
        t = timeit(lambda: ch.forward(x, None, N, save=True))
        fwd = ch.forward(x, None, N)
        t2 = timeit(lambda: ch.backward(fwd, N, dy=dy))
        print(f'{mode:7s} {name:8s} fwd+save {t*1e3:6.2f} ms {flop/t/1e12:6.1f} TF | bwd {t2*1e3:6.2f} ms {flop*7/8/t2/1e12:6.1f} TF')
    bwd = ch.backward(fwd, N, dy=dy)
    t = timeit(lambda: ch.weight_grads(fwd, bwd, N, x, None, workspace=ws))
    print(f'{mode:7s} dW {t*1e3:6.2f} ms {flop/t/1e12:6.1f} TF')
