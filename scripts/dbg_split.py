"""split-bf16 engine vs f32 engine vs fp64 torch: accuracy and speed of the forward chain (run on the GPU box)"""
import math, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from nero_amd import _lib as L
from nero_amd import chain as CH
from nero_amd.chain import Chain, Dense, Head, row_pad
g = torch.Generator().manual_seed(0)

def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))

def mk(n_out, n_in, s=2.0):
    return (torch.randn(n_out, n_in, generator=g) * s / math.sqrt(n_in)).cuda(), (torch.randn(n_out, generator=g) * 0.1).cuda()

# ---- accuracy: predictor-like chain with an odd input width, skip-like aux input and a head -----------------------------
n_rows, k_in, k_aux = 1000, 123, 39
rp = row_pad(n_rows)
W0, b0 = mk(256, k_in); W1, b1 = mk(256, 256 + k_aux); W2, b2 = mk(256, 256); W3, b3 = mk(3, 256)
X = torch.zeros(rp, 128, device='cuda'); X[:n_rows, :k_in] = torch.randn(n_rows, k_in, generator=g).cuda()
A = torch.zeros(rp, 40, device='cuda'); A[:n_rows, :k_aux] = torch.randn(n_rows, k_aux, generator=g).cuda()
for act, name in ((L.ACT_RELU, 'relu'), (L.ACT_SOFTPLUS100, 'softplus')):
    ref = None
    for mode in ('f32', 'bf16x6', 'f16x3'):
        CH.GEMM_MODE['fwd'] = CH._MODE_NAMES[mode]
        ch = Chain([(Dense(W0, b0, act, k_in), None), (Dense(W1, b1, act, 256, 0, k_aux, 256), None),
                    (Dense(W2, b2, act, 256), None), (None, Head(W3, b3))], k_init=128, k_aux=40).pack()
        fwd = ch.forward(X, A, n_rows)
        if ref is None:
            a = lambda t: F.relu(t) if act == L.ACT_RELU else F.softplus(t, beta=100)
            x, ax = X[:n_rows, :k_in].double().cpu(), A[:n_rows, :k_aux].double().cpu()
            h0 = a(F.linear(x, W0.double().cpu(), b0.double().cpu()))
            h1 = a(F.linear(torch.cat([h0, ax], 1), W1.double().cpu(), b1.double().cpu()))
            h2 = a(F.linear(h1, W2.double().cpu(), b2.double().cpu()))
            y = F.linear(h2, W3.double().cpu(), b3.double().cpu())
            ref = (h0, h1, h2, y)
        print(f'{name:8s} {mode:7s} rel err vs fp64: h0 {rel(fwd["saves"][0][:n_rows], ref[0]):.2e}  h1 {rel(fwd["saves"][1][:n_rows], ref[1]):.2e}  '
              f'h2 {rel(fwd["saves"][2][:n_rows], ref[2]):.2e}  head {rel(fwd["heads"][3][:n_rows, :3], ref[3]):.2e}')

# ---- speed: 8 x (256 -> 256) ----------------------------------------------------------------------------------------------
N = 524288
rpN = row_pad(N)
x = torch.randn(rpN, 256, device='cuda') * 0.1
Ws = [((torch.randn(256, 256, generator=g) / 16).cuda(), (torch.randn(256, generator=g) * 0.01).cuda()) for _ in range(8)]
def timeit(f, n=5):
    f(); torch.cuda.synchronize(); t = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time() - t) / n
flop = 2 * 8 * 256 * 256 * N
for mode in ('f32', 'bf16x6', 'f16x3'):
    CH.GEMM_MODE['fwd'] = CH._MODE_NAMES[mode]
    for name, act in (('relu', L.ACT_RELU), ('softplus', L.ACT_SOFTPLUS100)):
        ch = Chain([(Dense(W, b, act, 256), None) for W, b in Ws[:7]] + [(Dense(Ws[7][0], Ws[7][1], L.ACT_NONE, 256), None)], k_init=256).pack()
        for save in (False, True):
            t = timeit(lambda: ch.forward(x, None, N, save=save))
            print(f'{mode:7s} fwd act={name:8s} save={save}: {t*1e3:6.2f} ms {flop/t/1e12:6.1f} TF')
