"""head weight gradient (nero_head_dw: dWh = dy^T a over n rows) at the row counts of the training steps: time per call (kernel + its
reduction), HBM rate, error against fp64; with and without the `extra` operand."""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, '.')
from nero_amd import _lib as L

for n in (297000, 131072, 850000):
    dy = torch.randn(n, 4, device='cuda')
    a = torch.randn(n, 256, device='cuda')
    ex = torch.randn(n, 256, device='cuda')
    dWh = torch.empty(4, 256, device='cuda')
    dbh = torch.empty(4, device='cuda')
    ws = torch.empty(L.lib.nero_dw_workspace_floats(n), device='cuda')
    for extra in (None, ex):
        f = lambda: L.check(L.lib.nero_head_dw(C.c_void_p(dy.data_ptr()), C.c_void_p(a.data_ptr()), C.c_void_p(extra.data_ptr()) if extra is not None else None,
                                               3, n, C.c_void_p(dWh.data_ptr()), C.c_void_p(dbh.data_ptr()), C.c_void_p(ws.data_ptr()), 0, L.stream_ptr()))
        f()
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(20):
            f()
        torch.cuda.synchronize()
        dt = (time.time() - t) / 20
        ref = dy[:, :3].double().t() @ a.double()
        if extra is not None:
            ref[0] += extra.double().sum(0)
        nbytes = n * (1040 + (1024 if extra is not None else 0))
        print(f'n {n:7d} extra {extra is not None!s:5}: {dt * 1e6:7.1f} us  {nbytes / dt / 1e12:.2f} TB/s  err {float((dWh[:3].double() - ref).abs().max() / ref.abs().max()):.1e}')
