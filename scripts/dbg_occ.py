import sys, torch
sys.path.insert(0, '.')
from nero_amd import _lib as L
from nero_amd import chain as CH
from nero_amd.chain import Chain, Dense, row_pad
import math
CH.set_gemm_mode('f16x3p')
N=65536
x = torch.randn(row_pad(N), 256, device='cuda')
W = torch.randn(256,256).cuda()/16; b=torch.zeros(256).cuda()
ch = Chain([(Dense(W, b, L.ACT_RELU, 256), None)]*2, k_init=256).pack()
ch.forward(x, None, N, save=False); torch.cuda.synchronize()
p = torch.cuda.get_device_properties(0)
print(p.name, p.multi_processor_count, getattr(p,'shared_memory_per_block',None), getattr(p,'shared_memory_per_multiprocessor',None), getattr(p, 'max_threads_per_multi_processor', None))
