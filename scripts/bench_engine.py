"""micro-benchmark of the chain kernels on the SDF-shaped network (run on the GPU box)."""
import math, sys, time
import torch
sys.path.insert(0, '.')
from nero_amd import _lib as L
from nero_amd.chain import Chain, Dense, Head, row_pad

def mk(n_out, n_in, g):
    return (torch.randn(n_out, n_in, generator=g) / math.sqrt(n_in)).cuda(), (torch.randn(n_out, generator=g) * 0.1).cuda()

g = torch.Generator().manual_seed(0)
dims = [(256, 39), (256, 256), (256, 256), (217, 256), (256, 256), (256, 256), (256, 256), (256, 256), (257, 256)]
Ws = [mk(*d, g) for d in dims]
ent = []
for l, (W, b) in enumerate(Ws):
    if l == 0: ent.append((Dense(W, b, L.ACT_SOFTPLUS100, 39), None))
    elif l == 4: ent.append((Dense(W, b, L.ACT_SOFTPLUS100, 217, 0, 39, 217, 0.7071), None))
    elif l == 8: ent.append((Dense(W[1:], b[1:], L.ACT_NONE, 256), Head(W[0:1], b[0:1])))
    else: ent.append((Dense(W, b, L.ACT_SOFTPLUS100, 256), None))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
rp = row_pad(N)
pe = torch.randn(rp, 40, device='cuda') * 0.5
ch = Chain(ent, k_init=40, k_aux=40).pack()
macs = sum(a * b for a, b in dims)
def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.time() - t) / n
t = timeit(lambda: ch.forward(pe, pe, N, save=False))
print(f'fwd nosave  {t*1e3:.2f} ms  {2*macs*N/t/1e12:.1f} TFLOP/s')
t = timeit(lambda: ch.forward(pe, pe, N, save=True))
print(f'fwd save    {t*1e3:.2f} ms  {2*macs*N/t/1e12:.1f} TFLOP/s')
fwd = ch.forward(pe, pe, N)
dfeat = torch.randn(rp, 256, device='cuda'); dsdf = torch.randn(rp, 4, device='cuda')
t = timeit(lambda: ch.backward(fwd, N, dy=dfeat, head_dys={8: dsdf}))
print(f'bwd         {t*1e3:.2f} ms  {2*macs*N/t/1e12:.1f} TFLOP/s')
bwd = ch.backward(fwd, N, dy=dfeat, head_dys={8: dsdf})
ws = torch.empty(L.lib.nero_dw_workspace_floats(N), device='cuda')
t = timeit(lambda: ch.weight_grads(fwd, bwd, N, pe, pe, head_dys={8: dsdf}, workspace=ws))
print(f'dW          {t*1e3:.2f} ms  {2*macs*N/t/1e12:.1f} TFLOP/s')
