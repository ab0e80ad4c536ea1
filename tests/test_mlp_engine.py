"""GPU tier: the fused MLP-chain kernels (forward / reverse / weight-gradient GEMM) against plain fp64 torch math
of the same layers.  Tolerances: 2e-5 rel (fp32 accumulation-order noise over K<=340)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu




@pytest.fixture(params=['f32', 'bf16x6', 'f16x3'], autouse=True)
def gemm_mode(request):
    """every engine test runs on all dense-layer arithmetics (include/nero_hip.h NERO_GEMM_*): the exact fp32 MFMA, the
    3-plane bf16 split and the 2-plane block-scaled fp16 split, against the same fp64 reference and the same tolerance."""
    from nero_amd import chain
    old = dict(chain.GEMM_MODE)
    chain.set_gemm_mode(request.param)          # ('f16x3' keeps the weight-gradient GEMM on bf16x6: the shipped default)
    yield request.param
    chain.GEMM_MODE.update(old)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _mk(n_out, n_in, g, scale=1.0):
    W = (torch.randn(n_out, n_in, generator=g) * scale / math.sqrt(n_in)).cuda()
    b = (torch.randn(n_out, generator=g) * 0.1).cuda()
    return W, b


@pytest.mark.parametrize('n_rows', [64, 1000])
def test_predictor_chain_fwd_bwd(n_rows):
    from nero_amd import _lib as L
    from nero_amd.chain import Chain, Dense, Head, row_pad
    g = torch.Generator().manual_seed(0)
    k_in = 123
    W0, b0 = _mk(256, k_in, g, 2.0)
    W1, b1 = _mk(256, 256, g, 2.0)
    W2, b2 = _mk(256, 256, g, 2.0)
    W3, b3 = _mk(3, 256, g, 2.0)
    rp = row_pad(n_rows)
    kp = 128
    X = torch.zeros(rp, kp, device='cuda')
    X[:n_rows, :k_in] = torch.randn(n_rows, k_in, generator=g).cuda()
    ch = Chain([(Dense(W0, b0, L.ACT_RELU, k_in), None), (Dense(W1, b1, L.ACT_RELU, 256), None),
                (Dense(W2, b2, L.ACT_RELU, 256), None), (None, Head(W3, b3))], k_init=kp).pack()
    fwd = ch.forward(X, None, n_rows)
    Ws = [w.double().cpu() for w in (W0, W1, W2, W3)]
    bs = [b.double().cpu() for b in (b0, b1, b2, b3)]
    x = X[:n_rows, :k_in].double().cpu()
    hs = [x]
    for i in range(3):
        hs.append(F.relu(F.linear(hs[-1], Ws[i], bs[i])))
    y = F.linear(hs[3], Ws[3], bs[3])
    assert rel(fwd['heads'][3][:n_rows, :3], y) < 2e-5
    assert rel(fwd['saves'][2][:n_rows], hs[3]) < 2e-5
    dy = torch.zeros(rp, 4, device='cuda')
    dy[:n_rows, :3] = torch.randn(n_rows, 3, generator=g).cuda()
    # fp64 reverse pass with the ReLU masks of the kernel's OWN saved activations: a pre-activation within fp32 noise of 0
    # may legitimately land on either side (the two arithmetics and fp64 differ in the last bits), which flips one whole
    # gradient entry and says nothing about the GEMMs
    masks = [(fwd['saves'][i][:n_rows] > 0).double().cpu() for i in range(3)]
    dyd = dy[:n_rows, :3].double().cpu()
    gW, gb = [None] * 4, [None] * 4
    gW[3], gb[3] = dyd.t() @ hs[3], dyd.sum(0)
    dh = dyd @ Ws[3]
    for i in (2, 1, 0):
        dz = dh * masks[i]
        gW[i], gb[i] = dz.t() @ hs[i], dz.sum(0)
        dh = dz @ Ws[i]
    bwd = ch.backward(fwd, n_rows, head_dys={3: dy}, need_dinit=True)
    assert rel(bwd['d_init'][:n_rows, :k_in], dh) < 2e-5
    gr = ch.weight_grads(fwd, bwd, n_rows, X, None, head_dys={3: dy})
    for i in range(3):
        assert rel(gr[i]['dW'], gW[i]) < 2e-5, i
        assert rel(gr[i]['db'], gb[i]) < 2e-5, i
    assert rel(gr[3]['dWh'], gW[3]) < 2e-5
    assert rel(gr[3]['dbh'], gb[3]) < 2e-5


def sdf_entries(P):
    from nero_amd import _lib as L
    from nero_amd.chain import Dense, Head
    e = []
    for l in range(9):
        W, b = P[f'sdf_network.lin{l}.weight'], P[f'sdf_network.lin{l}.bias']
        if l == 0:
            e.append((Dense(W, b, L.ACT_SOFTPLUS100, 39), None))
        elif l == 4:
            e.append((Dense(W, b, L.ACT_SOFTPLUS100, 217, 0, 39, 217, 1.0 / math.sqrt(2)), None))
        elif l == 8:
            e.append((Dense(W[1:], b[1:], L.ACT_NONE, 256), Head(W[0:1], b[0:1])))
        else:
            e.append((Dense(W, b, L.ACT_SOFTPLUS100, 256), None))
    return e


def test_sdf_chain_forward_and_first_order_backward():
    from nero_amd.chain import Chain, row_pad
    from oracle import nero_oracle as O
    from tests.helpers import build_case_model, load_golden
    _, meta = load_golden('bell_s25000')
    net = build_case_model(meta).cuda()
    P = O.effective_params({k: v.detach() for k, v in net.state_dict().items()})
    n = 777
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(n, 3, generator=g) * 1.6 - 0.8)
    rp = row_pad(n)
    pe = torch.zeros(rp, 40, device='cuda')
    pe[:n, :39] = O.pos_enc(x, 6).cuda()
    ch = Chain(sdf_entries(P), k_init=40, k_aux=40).pack()
    fwd = ch.forward(pe, pe, n)
    Pd = {k: v.double().cpu() for k, v in P.items()}
    y = O.sdf_network(Pd, x.double())
    assert rel(fwd['heads'][8][:n, 0], y[:, 0]) < 2e-5
    assert rel(fwd['saves'][8][:n], y[:, 1:]) < 2e-5
    dfeat = torch.zeros(rp, 256, device='cuda')
    dfeat[:n] = torch.randn(n, 256, generator=g).cuda() * 0.1
    dsdf = torch.zeros(rp, 4, device='cuda')
    dsdf[:n, 0] = torch.randn(n, generator=g).cuda()
    Wl = {k: v.clone().requires_grad_(True) for k, v in Pd.items() if k.startswith('sdf_network')}
    y2 = O.sdf_network(Wl, x.double())
    ((y2[:, 0] * dsdf[:n, 0].double().cpu()).sum() + (y2[:, 1:] * dfeat[:n].double().cpu()).sum()).backward()
    bwd = ch.backward(fwd, n, dy=dfeat, head_dys={8: dsdf})
    gr = ch.weight_grads(fwd, bwd, n, pe, pe, head_dys={8: dsdf})
    for l in range(9):
        ref_w = Wl[f'sdf_network.lin{l}.weight'].grad
        ref_b = Wl[f'sdf_network.lin{l}.bias'].grad
        if l == 8:
            assert rel(gr[l]['dW'], ref_w[1:]) < 5e-5
            assert rel(gr[l]['dWh'], ref_w[0:1]) < 5e-5
            assert rel(gr[l]['db'], ref_b[1:]) < 5e-5
        else:
            assert rel(gr[l]['dW'], ref_w) < 5e-5, l
            assert rel(gr[l]['db'], ref_b) < 5e-5, l


def test_sdf_normal_and_second_order_weight_grads():
    """value/feature/normal and the full second-order weight gradient vs torch double-backward in fp64."""
    from nero_amd.chain import row_pad
    from nero_amd.sdf import SDFField
    from oracle import nero_oracle as O
    from tests.helpers import build_case_model, load_golden
    _, meta = load_golden('bell_s25000')
    net = build_case_model(meta).cuda()
    eff = [(w.detach(), b.detach()) for w, b in net.sdf_network.effective()]
    field = SDFField(eff).pack()
    n = 500
    g = torch.Generator().manual_seed(2)
    x = (torch.rand(n, 3, generator=g) * 1.4 - 0.7)
    xc = x.cuda().contiguous()
    ctx = field.forward_normal(xc, n)
    P = {f'sdf_network.lin{l}.weight': eff[l][0].double().cpu().requires_grad_(True) for l in range(9)}
    P.update({f'sdf_network.lin{l}.bias': eff[l][1].double().cpu().requires_grad_(True) for l in range(9)})
    xr = x.double().requires_grad_(True)
    y = O.sdf_network(P, xr)
    (nrm,) = torch.autograd.grad(y[:, 0].sum(), xr, create_graph=True)
    assert rel(ctx['sdf4'][:n, 0], y[:, 0]) < 2e-5
    assert rel(ctx['normal'], nrm) < 5e-5
    rp = row_pad(n)
    d_sdf4 = torch.zeros(rp, 4, device='cuda'); d_sdf4[:n, 0] = torch.randn(n, generator=g).cuda()
    d_feat = torch.zeros(rp, 256, device='cuda'); d_feat[:n] = torch.randn(n, 256, generator=g).cuda() * 0.1
    d_n = torch.randn(n, 3, generator=g).cuda()
    loss = (y[:, 0] * d_sdf4[:n, 0].double().cpu()).sum() + (y[:, 1:] * d_feat[:n].double().cpu()).sum() + (nrm * d_n.double().cpu()).sum()
    loss.backward()
    grads = field.backward(ctx, d_sdf4, d_feat, d_n)
    for l in range(9):
        assert rel(grads[l][0], P[f'sdf_network.lin{l}.weight'].grad) < 1e-4, l
        assert rel(grads[l][1], P[f'sdf_network.lin{l}.bias'].grad) < 1e-4, l
    # sdf-only evaluation path used by the sampler
    assert rel(field.sdf(xc)[:, 0], y[:, 0]) < 2e-5


@pytest.mark.parametrize('transpose', [0, 1])
def test_split_operand_planes_are_exact_and_in_fragment_order(transpose):
    """nero_pack_weight_split / nero_pack_batch (include/nero_hip.h): the three bf16 planes of every element sum EXACTLY
    (bit for bit, in fp32) to W*scale -- the property the fp32-grade claim of NERO_GEMM_BF16X6 rests on -- and sit at
    out[(((t*(kpad/16) + c)*3 + p)*64 + lane)*8 + j] = plane_p(A[32t + (lane&31)][16c + 8(lane>>5) + j]), zero outside."""
    import ctypes as C
    import numpy as np
    from nero_amd import _lib as L
    g = torch.Generator().manual_seed(3)
    nrows, ld, col0, ncols, scale = 70, 150, 5, 123, 0.70710678
    # wide dynamic range incl. tiny values, zeros and negative numbers
    W = (torch.randn(nrows, ld, generator=g) * torch.exp(torch.randn(nrows, ld, generator=g) * 6)).cuda()
    W[3, 7] = 0.0
    M, K = (nrows, ncols) if not transpose else (ncols, nrows)
    kpad, nt = (K + 15) // 16 * 16, (M + 31) // 32
    n_u16 = nt * (kpad // 16) * 3 * 64 * 8
    outs = []
    for batched in (False, True):
        out = torch.zeros(n_u16 // 2, dtype=torch.int32, device='cuda')
        if not batched:
            L.check(L.lib.nero_pack_weight_split(C.c_void_p(W.data_ptr()), nrows, ld, col0, ncols, transpose, C.c_float(scale),
                                                 kpad, nt, C.c_void_p(out.data_ptr()), L.stream_ptr()))
        else:
            j = L.PackJob()
            j.W, j.out, j.kind, j.nrows, j.ld, j.col0, j.ncols = W.data_ptr(), out.data_ptr(), 0, nrows, ld, col0, ncols
            j.transpose, j.kpad, j.nt_count, j.scale = transpose, kpad, nt, scale
            arr = (L.PackJob * 1)(j)
            L.check(L.lib.nero_pack_batch(arr, 1, L.stream_ptr()))
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy().view(np.uint16).reshape(nt, kpad // 16, 3, 64, 8))
    assert np.array_equal(outs[0], outs[1])
    img = outs[0]
    planes = (img.astype(np.uint32) << 16).view(np.float32)                  # bf16 -> fp32, exact
    total = (planes[:, :, 0] + planes[:, :, 1]) + planes[:, :, 2]            # [nt, steps, 64, 8]; each partial sum is exact in fp32
    Wn = (W.cpu().numpy()[:, col0:col0 + ncols] * np.float32(scale)).astype(np.float32)
    A = Wn if not transpose else Wn.T
    want = np.zeros((nt * 32, kpad), np.float32)
    want[:M, :K] = A
    lane = np.arange(64)
    for t in range(nt):
        for c in range(kpad // 16):
            rows = 32 * t + (lane & 31)
            cols = 16 * c + 8 * (lane >> 5)
            blk = np.stack([want[rows, cols + j] for j in range(8)], -1)      # [64, 8]
            assert np.array_equal(total[t, c].view(np.uint32), blk.view(np.uint32)), (t, c)
    # plane magnitudes: |x1| <= 2^-8 |x0|, |x2| <= 2^-8 |x1| wherever x0 != 0
    p0, p1, p2 = np.abs(planes[:, :, 0]), np.abs(planes[:, :, 1]), np.abs(planes[:, :, 2])
    nz = p0 > 0
    assert (p1[nz] <= p0[nz] * 2.0 ** -8).all() and (p2[nz] <= p0[nz] * 2.0 ** -16).all()


@pytest.mark.parametrize('transpose', [0, 1])
def test_f16x3_operand_planes_reconstruct_to_fp32_precision(transpose):
    """nero_pack_batch kind 3 (NERO_GEMM_F16X3 operand; round 5: one accumulator, mlp_f16_util.h): header float[0] = 2^e with
    max|A| * 2^-e in [2^14, 2^15) -- the top of fp16's range -- and the plane pair (h + l) * 2^e, l the remainder at its TRUE scale,
    reproduces A to <= 2^-24 of the block maximum for every element, and to <= 2^-22 relative for every element within 2^13 of the
    maximum -- the bound the fp32-grade claim of the engine rests on."""
    import ctypes as C
    import numpy as np
    from nero_amd import _lib as L
    g = torch.Generator().manual_seed(5)
    nrows, ld, col0, ncols, scale = 70, 150, 5, 123, 0.70710678
    W = (torch.randn(nrows, ld, generator=g) * torch.exp(torch.randn(nrows, ld, generator=g) * 3)).cuda()
    W[3, 7] = 0.0
    M, K = (nrows, ncols) if not transpose else (ncols, nrows)
    kpad, nt = (K + 15) // 16 * 16, (M + 31) // 32
    n_bytes = 256 + nt * (kpad // 16) * 2 * 64 * 16
    out = torch.zeros(n_bytes // 4, dtype=torch.int32, device='cuda')
    j = L.PackJob()
    j.W, j.out, j.kind, j.nrows, j.ld, j.col0, j.ncols = W.data_ptr(), out.data_ptr(), 3, nrows, ld, col0, ncols
    j.transpose, j.kpad, j.nt_count, j.scale = transpose, kpad, nt, scale
    L.check(L.lib.nero_pack_batch((L.PackJob * 1)(j), 1, L.stream_ptr()))
    torch.cuda.synchronize()
    raw = out.cpu().numpy()
    sc = raw[:1].view(np.float32)[0]
    Wn = (W.cpu().numpy()[:, col0:col0 + ncols] * np.float32(scale)).astype(np.float32)
    A = Wn if not transpose else Wn.T
    amax = np.abs(A).max()
    assert 2.0 ** 14 <= amax / sc < 2.0 ** 15, (amax, sc)
    planes = raw[64:].view(np.float16).reshape(nt, kpad // 16, 2, 64, 8).astype(np.float64)
    rec = (planes[:, :, 0] + planes[:, :, 1]) * float(sc)                     # [nt, steps, 64, 8]
    want = np.zeros((nt * 32, kpad), np.float64)
    want[:M, :K] = A
    lane = np.arange(64)
    err_abs, err_rel = 0.0, 0.0
    for t in range(nt):
        for c in range(kpad // 16):
            rows = 32 * t + (lane & 31)
            cols = 16 * c + 8 * (lane >> 5)
            blk = np.stack([want[rows, cols + jj] for jj in range(8)], -1)
            d = np.abs(rec[t, c] - blk)
            err_abs = max(err_abs, d.max())
            big = np.abs(blk) >= amax * 2.0 ** -13
            if big.any():
                err_rel = max(err_rel, (d[big] / np.abs(blk[big])).max())
    assert err_abs <= 2.0 ** -24 * amax, (err_abs, amax)
    assert err_rel <= 2.0 ** -22, err_rel
