"""Where does a LONE wave per SIMD lose the matrix pipe in the chain k-loop?  
The two-workgroups-per-CU forward kernel (mlp_f16p.hip, 4 waves per workgroup) run on the synthetic 8 x 256 ReLU chain (no saves) with ONE
workgroup per CU (-DP_LDS_EXTRA=4096: one wave per SIMD) and the compile-time switches of mlp_f16_util.h that take one ingredient of the k-step
away each: F16_NO_WSTREAM (every k-step re-reads the weight fragments of step 0: the L2 round trip of the weight ring becomes an L1 hit),
F16_NO_MFMA (no matrix instructions: what is left is request issue + waits), both.  Per (64-row tile, layer): shader-clock cycles of wave 0
in the GEMM phase (two feature tiles = 32 k-steps = 192 MFMAs = 6144 cycles at 32 per MFMA) and the launch's wall time.
usage (GPU box): NERO_HIP_LIB=build/variants/libph_<variant>.so python scripts/r06/lone_wave_probe.py"""
import ctypes as C, math, sys, time
import torch
sys.path.insert(0, '.')
from nero_amd import _lib as L
from nero_amd import chain as CH
from nero_amd.chain import Chain, Dense, row_pad
N = 524288
g = torch.Generator().manual_seed(0)
rp = row_pad(N)
x = torch.randn(rp, 256, device='cuda') * 0.1
def mk(n_out, n_in, s=1.0): return ((torch.randn(n_out, n_in, generator=g) * s / math.sqrt(n_in)).cuda(), (torch.randn(n_out, generator=g) * 0.01).cuda())
Ws = [mk(256, 256, 1.4) for _ in range(8)]
ch = Chain([(Dense(W, b, L.ACT_RELU, 256), None) for W, b in Ws[:7]] + [(Dense(Ws[7][0], Ws[7][1], L.ACT_NONE, 256), None)], k_init=256).pack()
CH.f16_paired(9)                                   # forward on the paired kernel whatever the size
buf = (C.c_ulonglong * 16)()
def phases(reset=True):
    L.lib.nero_debug_phases_p(buf, int(reset))
    return [int(v) for v in buf[:8]]
f = lambda: ch.forward(x, None, N, save=False)
f(); torch.cuda.synchronize(); phases()
t = time.time()
for _ in range(5): f()
torch.cuda.synchronize()
dt = (time.time() - t) / 5
ph = phases()
tiles = rp // 64
per = [p / (5 * tiles * 8) for p in ph]
names = ('init', 'pre-gemm', 'gemm (2 feature tiles)', 'act', 'save+mask+publish', 'commit (2 barriers)', '-', 'residence')
print(f'{dt*1e3:7.3f} ms  {dt*1e6/(tiles/256*8):6.2f} us per layer-tile per CU   ' + '  '.join(f'{n}={p:.0f}' for n, p in zip(names, per) if n != '-'))
