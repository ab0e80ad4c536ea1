"""Per-phase shader-clock breakdown of the fp16 tile engine's forward / reverse chain kernels (run on the GPU box with a library
built with -DF16_PHASE_TIMING, see scripts/f16_variants.sh):  NERO_HIP_LIB=build/variants/libf16_phase.so python scripts/phase_timing.py
Prints cycles per (64-row tile, layer) of wave 0 per phase, and the wall time of each launch."""
import ctypes as C, math, sys, time
import torch
sys.path.insert(0, '.')
from nero_amd import _lib as L
from nero_amd import chain as CH
from nero_amd.chain import Chain, Dense, row_pad

N = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
g = torch.Generator().manual_seed(0)
rp = row_pad(N)
x = torch.randn(rp, 256, device='cuda') * 0.1
dy = torch.randn(rp, 256, device='cuda') * 1e-3
def mk(n_out, n_in, s=1.0): return ((torch.randn(n_out, n_in, generator=g) * s / math.sqrt(n_in)).cuda(), (torch.randn(n_out, generator=g) * 0.01).cuda())
Ws = [mk(256, 256, 1.4) for _ in range(8)]
MODE = sys.argv[2] if len(sys.argv) > 2 else 'f16x3'
CH.set_gemm_mode(MODE)
fn = 'nero_debug_phases'
has_ph = hasattr(L.lib, fn)
buf = (C.c_ulonglong * 16)()
def phases(reset=True):
    if not has_ph: return None
    getattr(L.lib, fn)(buf, int(reset))
    return [int(v) for v in buf[:8]]
def timeit(f, n=5):
    f(); torch.cuda.synchronize(); phases()
    t = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize()
    dt = (time.time() - t) / n
    return dt, phases()
tiles = rp // 64
FW = ('init', 'pre-gemm', 'gemm', 'act', 'save+mask', 'rowmax+bar1', 'planes', 'bar2')
BW = ('-', 'pre-gemm', 'gemm', 'gq', 'values', 'delta st', 'commit', '-')
QUICK = len(sys.argv) > 3 and sys.argv[3] == 'quick'
for name, act in ((('relu', L.ACT_RELU),) if QUICK else (('relu', L.ACT_RELU), ('softplus', L.ACT_SOFTPLUS100))):
    ch = Chain([(Dense(W, b, act, 256), None) for W, b in Ws[:7]] + [(Dense(Ws[7][0], Ws[7][1], L.ACT_NONE, 256), None)], k_init=256).pack()
    for save in (False, True):
        t, ph = timeit(lambda: ch.forward(x, None, N, save=save))
        fl = 2 * 8 * 65536 * N
        print(f'fwd {name:8s} save={int(save)}: {t*1e3:7.3f} ms {fl/t/1e12:6.1f} TF  ({t*1e6/ (tiles/256*8):.2f} us per layer-tile per CU)')
        if ph:
            tot = sum(ph)
            print('      ' + '  '.join(f'{n}={p/(5*tiles*8):.0f}' for n, p in zip(FW, ph)) + f'   total={tot/(5*tiles*8):.0f} cycles/layer-tile')
    fwd = ch.forward(x, None, N, save=True)
    t, ph = timeit(lambda: ch.backward(fwd, N, dy=dy, need_dinit=True))
    print(f'bwd {name:8s}       : {t*1e3:7.3f} ms {fl/t/1e12:6.1f} TF')
    if ph:
        tot = sum(ph)
        print('      ' + '  '.join(f'{n}={p/(5*tiles*8):.0f}' for n, p in zip(BW, ph)) + f'   total={tot/(5*tiles*8):.0f} cycles/layer-tile')
    injs = {i: torch.randn(rp, 256, device='cuda') * 1e-4 for i in range(7)}
    t, ph = timeit(lambda: ch.backward(fwd, N, dy=dy, need_dinit=True, injs=injs))
    print(f'bwd {name:8s} +inj  : {t*1e3:7.3f} ms {fl/t/1e12:6.1f} TF')
    if ph:
        print('      ' + '  '.join(f'{n}={p/(5*tiles*8):.0f}' for n, p in zip(BW, ph)))
    del injs, fwd
