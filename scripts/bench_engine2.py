"""engine ceiling experiments: 8x(256->256) chain with different epilogues (run on the GPU box)"""
import math, sys, time
import torch
sys.path.insert(0, '.')
from nero_amd import _lib as L
from nero_amd.chain import Chain, Dense, Head, row_pad
g = torch.Generator().manual_seed(0)
N = 524288
rp = row_pad(N)
x = torch.randn(rp, 256, device='cuda') * 0.1
def mk(): return (torch.randn(256, 256, generator=g) / 16).cuda(), (torch.randn(256, generator=g) * 0.01).cuda()
Ws = [mk() for _ in range(8)]
def timeit(f, n=5):
    f(); torch.cuda.synchronize(); t = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time() - t) / n
flop = 2 * 8 * 256 * 256 * N
for name, act in (('none', L.ACT_NONE), ('relu', L.ACT_RELU), ('softplus', L.ACT_SOFTPLUS100)):
    ch = Chain([(Dense(W, b, act, 256), None) for W, b in Ws[:7]] + [(Dense(Ws[7][0], Ws[7][1], L.ACT_NONE, 256), None)], k_init=256).pack()
    for save in (False, True):
        t = timeit(lambda: ch.forward(x, None, N, save=save))
        print(f'fwd act={name:8s} save={save}: {t*1e3:6.2f} ms {flop/t/1e12:6.1f} TF')
ch = Chain([(Dense(W, b, L.ACT_RELU, 256), None) for W, b in Ws[:7]] + [(Dense(Ws[7][0], Ws[7][1], L.ACT_NONE, 256), None)], k_init=256).pack()
fwd = ch.forward(x, None, N)
dy = torch.randn(rp, 256, device='cuda')
t = timeit(lambda: ch.backward(fwd, N, dy=dy))
print(f'bwd relu: {t*1e3:6.2f} ms {2*7*256*256*N/t/1e12:6.1f} TF')
bwd = ch.backward(fwd, N, dy=dy)
ws = torch.empty(L.lib.nero_dw_workspace_floats(N), device='cuda')
t = timeit(lambda: ch.weight_grads(fwd, bwd, N, x, None, workspace=ws))
print(f'dW: {t*1e3:6.2f} ms {flop/t/1e12:6.1f} TF')
