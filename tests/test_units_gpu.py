"""GPU tier: the HIP encoders / table fetch / transfer function called directly through the C ABI against the unit vectors that
oracle/gen_golden.py dumped from the reference's own helpers (tests/golden/units.npz): Embedder (network/field.py:14-58),
generate_ide_fn(5) (utils/ref_utils.py:53-117), dr.texture on the FG table (network/field.py:610-613; nvdiffrast restated by
oracle/ref_shim.py -- third party, convention unpinned), linear_to_srgb (utils/raw_utils.py:4-10)."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN, ref_fg_lut

pytestmark = pytest.mark.gpu
P = C.c_void_p


def _units():
    import os
    return np.load(os.path.join(GOLDEN, 'units.npz'))


def _cu(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def _p(t):
    return P(None if t is None else t.data_ptr())


def test_hip_positional_encoding_vs_reference_embedder():
    from nero_amd import _lib as L
    z = _units()
    for xk, pk, dim, nf in (('x', 'pe6', 3, 6), ('x4', 'pe10', 4, 10)):
        x = _cu(z[xk])
        n, w = x.shape[0], z[pk].shape[1]
        ld = (w + 3) // 4 * 4
        out = torch.full((64, ld), 7.0, device='cuda')
        L.check(L.lib.nero_encode_pe(_p(x), x.stride(0), dim, nf, n, _p(out), ld, L.stream_ptr()))
        got = out[:n, :w].cpu().numpy()
        # fp32 sin/cos of arguments up to 2^9 |x|: the reference evaluates the same fp32 products; device sincos differs by ulps
        assert np.abs(got - z[pk]).max() < 2e-5, (pk, np.abs(got - z[pk]).max())
        assert float(out[:n, w:].abs().max()) == 0.0 if ld > w else True


def _shade_encode(dirs, rough, pts=None):
    """nero_shade_encode with normal = reflection = dirs, sigmoid(r_raw) = rough.  -> Xd (IDE(n,1)), Xs (IDE(refl,rough)), Xi, Xo, mat"""
    from nero_amd import _lib as L
    n = dirs.shape[0]
    rp = (n + 63) // 64 * 64
    geo = torch.zeros(rp, 8, device='cuda')
    geo[:n, 0:3] = dirs
    geo[:n, 3] = 0.5
    geo[:n, 4:7] = dirs
    geo[:n, 7] = 1.0
    x4 = torch.zeros(rp, 4, device='cuda')
    if pts is not None:
        x4[:n, :3] = pts
    r_raw = torch.zeros(rp, 4, device='cuda')
    r_raw[:n, 0] = torch.logit(rough.double().clamp(1e-7, 1 - 1e-7)).float()[:, 0]
    m_raw, a_raw = torch.zeros(rp, 4, device='cuda'), torch.zeros(rp, 4, device='cuda')
    mat = torch.empty(rp, 8, device='cuda')
    Xd, Xs = torch.empty(rp, 72, device='cuda'), torch.empty(rp, 72, device='cuda')
    Xi, Xo = torch.empty(rp, 128, device='cuda'), torch.empty(rp, 96, device='cuda')
    L.check(L.lib.nero_shade_encode(_p(x4), _p(geo), _p(m_raw), _p(r_raw), _p(a_raw), n, _p(mat), _p(Xd), _p(Xs), _p(Xi), _p(Xo), 0,
                                    L.stream_ptr()))
    return Xd[:n], Xs[:n], Xi[:n], Xo[:n], mat[:n]


def test_hip_ide_vs_reference_generate_ide_fn():
    from oracle import nero_oracle as O
    z = _units()
    dirs, rough = _cu(z['dirs']), _cu(z['rough'])
    pts = _cu(z['x']) * 0.3
    Xd, Xs, Xi, Xo, mat = _shade_encode(dirs, rough, pts)
    # diffuse lights use kappa^-1 = 1 (field.py:580-581)
    assert np.abs(Xd.cpu().numpy() - z['ide_one']).max() < 2e-6
    # specular: the kernel's own sigmoid(raw) roughness drives the attenuation exp(-l(l+1)/2 r), l <= 16
    r_k = mat[:, 1:2].cpu()
    assert np.abs(r_k.numpy() - z['rough']).max() < 2e-7
    # vs the reference at ITS roughness: the logit -> sigmoid round trip of this test moves r by a few ulp, amplified by l(l+1)/2 = 136
    assert np.abs(Xs.cpu().numpy() - z['ide_rough']).max() < 2e-4
    assert np.abs(Xs.cpu().numpy() - O.ide(torch.from_numpy(z['dirs']), r_k).numpy()).max() < 2e-6     # vs the oracle at the kernel's roughness
    # light-MLP inputs: Xi = [PE-8(p) | IDE(refl, rough)], Xo = [PE-8(p) | PE-6(refl)]   (field.py:566-571)
    pe8 = O.pos_enc(pts.cpu(), 8).numpy()
    assert np.abs(Xi[:, :51].cpu().numpy() - pe8).max() < 2e-5 and np.abs(Xo[:, :51].cpu().numpy() - pe8).max() < 2e-5
    assert np.abs(Xi[:, 51:123].cpu().numpy() - Xs.cpu().numpy()).max() == 0.0
    assert np.abs(Xo[:, 51:90].cpu().numpy() - O.pos_enc(torch.from_numpy(z['dirs']), 6).numpy()).max() < 2e-5


def _inter(nov, rough, metallic, albedo, Ld=None, lut=None, exp_max=10.0):
    from nero_amd import _lib as L
    n = nov.shape[0]
    rp = (n + 63) // 64 * 64
    geo, mat = torch.zeros(rp, 8, device='cuda'), torch.zeros(rp, 8, device='cuda')
    geo[:n, 3] = nov
    mat[:n, 0], mat[:n, 1] = metallic, rough
    mat[:n, 2:5] = albedo
    z4 = lambda: torch.zeros(rp, 4, device='cuda')
    Ldt = z4()
    if Ld is not None:
        Ldt[:n, :3] = Ld
    rec = torch.empty(n, 32, device='cuda')
    lut = (ref_fg_lut() if lut is None else lut).cuda().contiguous()
    L.check(L.lib.nero_shade_inter_results(_p(geo), _p(mat), _p(Ldt), _p(z4()), _p(z4()), _p(z4()), _p(lut), C.c_float(exp_max), n, _p(None),
                                           _p(None), _p(rec), L.stream_ptr()))
    return rec


def test_hip_fg_lut_fetch_vs_reference_texture_call():
    """specular_ref = F0*FG.x + FG.y (field.py:614): two evaluations with F0 = 0.04 and F0 = 0.5 recover the fetched (FG.x, FG.y)"""
    z = _units()
    uv = _cu(z['uv'])
    n = uv.shape[0]
    one = torch.ones(n, device='cuda')
    s0 = _inter(uv[:, 0], uv[:, 1], 0.0 * one, torch.ones(n, 3, device='cuda') * 0.3)[:, 3]           # F0 = 0.04
    s1 = _inter(uv[:, 0], uv[:, 1], one, torch.ones(n, 3, device='cuda') * 0.5)[:, 3]                 # F0 = 0.5
    f0 = (s1 - s0) / 0.46
    f1 = s0 - 0.04 * f0
    got = torch.stack([f0, f1], -1).cpu().numpy()
    assert np.abs(got - z['fg']).max() < 3e-6, np.abs(got - z['fg']).max()


def test_hip_linear_to_srgb_vs_reference():
    z = _units()
    lin = _cu(z['lin'])
    n = lin.shape[0]
    one = torch.ones(n, device='cuda')
    rec = _inter(0.5 * one, 0.5 * one, 0 * one, torch.ones(n, 3, device='cuda'), Ld=torch.log(lin))
    got = rec[:, 15:18].cpu().numpy()                                  # clamp(sRGB(diffuse_light), 0, 1)
    want = np.clip(z['srgb'], 0.0, 1.0)
    assert np.abs(got - want).max() < 2e-6, np.abs(got - want).max()
