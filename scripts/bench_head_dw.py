import sys, os, time, ctypes as C, torch
sys.path.insert(0, '.')
from nero_amd import _lib as L
n = 297000
dy = torch.randn(n, 4, device='cuda'); a = torch.randn(n, 256, device='cuda')
dWh = torch.empty(4, 256, device='cuda'); dbh = torch.empty(4, device='cuda')
ws = torch.empty(L.lib.nero_dw_workspace_floats(n), device='cuda')
f = lambda: L.check(L.lib.nero_head_dw(C.c_void_p(dy.data_ptr()), C.c_void_p(a.data_ptr()), None, 3, n, C.c_void_p(dWh.data_ptr()), C.c_void_p(dbh.data_ptr()), C.c_void_p(ws.data_ptr()), 0, L.stream_ptr()))
f(); torch.cuda.synchronize(); t = time.time()
for _ in range(20): f()
torch.cuda.synchronize(); dt = (time.time() - t) / 20
ref = dy[:, :3].double().t() @ a.double()
print(f'{dt*1e6:.1f} us  {n*1040/dt/1e12:.2f} TB/s  err {float((dWh[:3].double()-ref).abs().max()/ref.abs().max()):.1e}')
