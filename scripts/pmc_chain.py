"""workload for scripts/prof_chain_pmc.sh: the 8 x (256 -> 256) ReLU chain forward (no saves) on the two fp16 chain engines, 3 launches each"""
import math, sys, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from nero_amd import _lib as L
from nero_amd import chain as CH
from nero_amd.chain import Chain, Dense, row_pad
N = 524288
g = torch.Generator().manual_seed(0)
x = torch.randn(row_pad(N), 256, device='cuda') * 0.1
Ws = [((torch.randn(256, 256, generator=g) * 1.4 / 16).cuda(), (torch.randn(256, generator=g) * 0.01).cuda()) for _ in range(8)]
for mode in (sys.argv[1].split(',') if len(sys.argv) > 1 else ['f16x3']):
    CH.set_gemm_mode(mode)
    ch = Chain([(Dense(W, b, L.ACT_RELU, 256), None) for W, b in Ws[:7]] + [(Dense(Ws[7][0], Ws[7][1], L.ACT_NONE, 256), None)], k_init=256).pack()
    for _ in range(3):
        ch.forward(x, None, N, save=False)
    torch.cuda.synchronize()
