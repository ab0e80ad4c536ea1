import math, os, sys, time
import torch
sys.path.insert(0, '.')
from nero_amd import _lib as L
from nero_amd import chain as CH
from nero_amd.chain import Chain, Dense, row_pad
g = torch.Generator().manual_seed(0)
N = 524288
x = torch.randn(row_pad(N), 256, device='cuda') * 0.1
Ws = [((torch.randn(256, 256, generator=g) / 16).cuda(), (torch.randn(256, generator=g) * 0.01).cuda()) for _ in range(8)]
def timeit(f, n=5):
    f(); torch.cuda.synchronize(); t = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time() - t) / n
flop = 2 * 8 * 256 * 256 * N
CH.GEMM_MODE['fwd'] = CH._MODE_NAMES['bf16x6']
for dbg in ('0', '1'):
    os.environ['NERO_SPLIT_DEBUG'] = dbg
    for name, act in (('relu', L.ACT_RELU), ('softplus', L.ACT_SOFTPLUS100)):
        ch = Chain([(Dense(W, b, act, 256), None) for W, b in Ws[:7]] + [(Dense(Ws[7][0], Ws[7][1], L.ACT_NONE, 256), None)], k_init=256).pack()
        for save in (False, True):
            t = timeit(lambda: ch.forward(x, None, N, save=save))
            print(f'dbg={dbg} fwd act={name:8s} save={save}: {t*1e3:6.2f} ms {flop/t/1e12:6.1f} TF  ({t/8/(N/64/256)*2.4e9:.0f} nominal cycles per layer-tile)')
