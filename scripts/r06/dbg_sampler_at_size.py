import sys, json, numpy as np, torch, ctypes as C
sys.path.insert(0,'.')
from nero_amd import _lib as L
z=np.load('tests/golden/at_size_bell_1024.npz'); meta=json.loads(str(z['meta']))
nt=meta['n_trace']; P=C.c_void_p; st=L.stream_ptr()
t=lambda k: torch.from_numpy(np.asarray(z[k]))
for i in range(4):
    zc=t(f'tr/z{i}').cuda().contiguous(); wt=t(f'tr/weights{i}').cuda().contiguous(); ref_new=t(f'tr/z_new{i}')
    n=zc.shape[1]; m=ref_new.shape[1]
    out=torch.empty(nt,m,device='cuda'); inds2=torch.empty(nt,m,dtype=torch.int32,device='cuda')
    L.check(L.lib.nero_sample_pdf(P(zc.data_ptr()), n, P(wt.data_ptr()), n-1, n, m, nt, P(out.data_ptr()), P(inds2.data_ptr()), st))
    dz=(out.cpu()-ref_new).abs()
    r,j=np.unravel_index(int(dz.argmax()), dz.shape)
    w=t(f'tr/weights{i}')[r].double()+1e-5; pdf=w/w.sum(); cdf=torch.cat([torch.zeros(1,dtype=torch.float64), torch.cumsum(pdf,0)])
    k=int(t(f'tr/inds{i}')[r][j]); below=max(k-1,0); above=min(k,n-1)
    print(f'round {i}: n {n} max|dz| {float(dz.max()):.3e} at ray {r} sample {j}; frac>2e-6 {float((dz>2e-6).float().mean()):.2e}; frac>1e-6 {float((dz>1e-6).float().mean()):.2e}; cdf interval {float(cdf[above]-cdf[below]):.3e}; bin width {float(zc[r,above]-zc[r,below]):.3e}; idx equal {bool(torch.equal(inds2.cpu(), t(f"tr/inds{i}").int()))}')
