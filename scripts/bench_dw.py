"""Weight-gradient GEMM: engines side by side on 256 x 256 jobs (run on the GPU box):  python scripts/bench_dw.py [rows]
Operands with heavy-tailed row magnitudes (per-row log-normal factors spanning ~2^40) and an outlier block; error = max |dW - ref|
/ max |ref| against an fp64 matmul.  The fp32 torch matmul is listed for the noise level of plain fp32 accumulation."""
import ctypes as C, sys, time
import torch
sys.path.insert(0, '.')
from nero_amd import _lib as L
N = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
g = torch.Generator(device='cuda').manual_seed(0)
def operands(kind):
    D = torch.randn(N, 256, device='cuda', generator=g)
    B = torch.randn(N, 256, device='cuda', generator=g)
    if kind == 'tails':
        D *= torch.exp(torch.randn(N, 1, device='cuda', generator=g) * 4.0) * 1e-4
        B *= torch.exp(torch.randn(N, 1, device='cuda', generator=g) * 2.0)
    elif kind == 'outlier-first':
        D *= 1e-6
        D[:16] *= 1e9
    elif kind == 'relu':
        B = torch.relu(B)
        D *= (torch.rand(N, 1, device='cuda', generator=g) < 0.3)
    return D.contiguous(), B.contiguous()
ws = torch.empty(L.lib.nero_dw_workspace_floats(N), dtype=torch.float32, device='cuda')
def run(mode, D, B, D1=None, B1=None, n_out=256, k=256):
    dW = torch.empty(n_out, k, device='cuda'); db = torch.empty(n_out, device='cuda')
    job = L.DwJob()
    job.d0, job.ldd0, job.b0, job.ldb0 = D.data_ptr(), D.stride(0), B.data_ptr(), B.stride(0)
    if D1 is not None:
        job.d1, job.ldd1, job.b1, job.ldb1 = D1.data_ptr(), D1.stride(0), B1.data_ptr(), B1.stride(0)
    job.n_out, job.k_cols, job.dW, job.ldw, job.col0, job.db = n_out, k, dW.data_ptr(), dW.stride(0), 0, db.data_ptr()
    job.scale, job.accumulate, job.gemm_mode = 1.0, 0, mode
    f = lambda: L.check(L.lib.nero_dw_gemm(C.byref(job), N, C.c_void_p(ws.data_ptr()), L.stream_ptr()))
    f(); torch.cuda.synchronize(); t = time.time()
    for _ in range(5): f()
    torch.cuda.synchronize()
    if mode == L.GEMM_F16X3 and hasattr(L.lib, 'nero_debug_phases_dw'):
        buf = (C.c_ulonglong * 8)()
        L.lib.nero_debug_phases_dw(buf, 1)
        chunks = 6 * (N // 16)                     # 6 launches x chunks (wave 0 of every slice)
        print('      phases (cycles per chunk): ' + '  '.join(f'{n}={int(v)/chunks:.0f}' for n, v in zip(('fetch', 'frags', 'convert', 'mfma', 'tail', 'barrier'), buf[:6])))
    return dW, db, (time.time() - t) / 5
rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
QUICK = len(sys.argv) > 2 and sys.argv[2] == 'quick'
for kind in (('tails',) if QUICK else ('plain', 'tails', 'outlier-first', 'relu')):
    D, B = operands(kind)
    ref = D.double().t() @ B.double()
    refb = D.double().sum(0)
    print(f'{kind}: torch fp32 matmul err {rel(D.t() @ B, ref):.2e}')
    for name, mode in (('f32', L.GEMM_F32), ('bf16x6', L.GEMM_BF16X6), ('f16x3', L.GEMM_F16X3)):
        dW, db, t = run(mode, D, B)
        print(f'  {name:7s} {t*1e3:7.3f} ms  {2*65536*N/t/1e12:6.1f} TF  {2*N*1024/t/1e12:5.2f} TB/s   err {rel(dW, ref):.2e}   db err {rel(db, refb):.2e}')
    del ref
if QUICK: sys.exit(0)
D, B = operands('tails'); D1, B1 = operands('plain')
ref = D.double().t() @ B.double() + D1.double().t() @ B1.double()
for name, mode in (('bf16x6', L.GEMM_BF16X6), ('f16x3', L.GEMM_F16X3)):
    dW, db, t = run(mode, D, B, D1, B1)
    print(f'two pairs {name:7s} {t*1e3:7.3f} ms  {4*65536*N/t/1e12:6.1f} TF   err {rel(dW, ref):.2e}')
for n_out, k in ((256, 48), (217, 256), (3, 256), (256, 96)):
    D, B = operands('tails')
    ref = D[:, :n_out].double().t() @ B[:, :k].double()
    for name, mode in (('bf16x6', L.GEMM_BF16X6), ('f16x3', L.GEMM_F16X3)):
        dW, db, t = run(mode, D, B, n_out=n_out, k=k)
        print(f'{n_out}x{k} {name:7s} {t*1e3:7.3f} ms   err {rel(dW, ref):.2e}')
