"""Pre-integrated split-sum BRDF table FG(NoV, roughness) used by the Stage-I shader.

The reference ships this table as a binary asset (assets/bsdf_256_256.bin, loaded at network/field.py:510, indexed
[roughness (v), NoV (u), 2]); reference checkpoints also carry it as the buffer `color_network.FG_LUT`, which
load_state_dict restores verbatim.  For from-scratch construction this module *computes* the table instead of copying
the asset:   A = int (1-Fc) G2 VoH/(NoH NoV),  B = int Fc G2 VoH/(NoH NoV)   over GGX-importance-sampled half vectors,
Fc = (1-VoH)^5, G2 = height-correlated Smith, alpha = roughness^2, texel centres at ((i+.5)/256).  Midpoint
quadrature, float64, deterministic.  tests/test_brdf_lut.py bounds |ours - reference| (<= 2e-3 abs).
"""
import os

import numpy as np
import torch

_CACHE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'assets', 'fg_lut_256.npy')


def compute_fg_lut(res=256, n_phi=48, n_theta=256):
    dt = torch.float64
    u = (torch.arange(res, dtype=dt) + 0.5) / res                 # NoV
    xi_p = (torch.arange(n_phi, dtype=dt) + 0.5) / n_phi          # phi in (0, pi): integrand is even in phi
    xi_t = (torch.arange(n_theta, dtype=dt) + 0.5) / n_theta
    phi = (np.pi * xi_p)[None, :, None]
    out = torch.zeros(res, res, 2, dtype=dt)
    nov = u[:, None, None]
    vx = torch.sqrt(1 - nov * nov)
    for vi in range(res):
        a = float(((vi + 0.5) / res) ** 2)
        cos_t = torch.sqrt((1 - xi_t) / (1 + (a * a - 1) * xi_t))[None, None, :]
        sin_t = torch.sqrt(1 - cos_t * cos_t)
        hx, hz = sin_t * torch.cos(phi), cos_t
        voh = vx * hx + nov * hz
        lz = 2 * voh * hz - nov
        nol = lz.clamp(min=1e-12)

        def lam(c):
            return (-1 + torch.sqrt(1 + a * a * (1 - c * c) / (c * c))) / 2
        g2 = 1.0 / (1.0 + lam(nov) + lam(nol))
        vohc = voh.clamp(0, 1)
        gv = g2 * vohc / (hz.clamp(min=1e-12) * nov)
        fc = (1 - vohc) ** 5
        m = (lz > 0).to(dt)
        out[vi, :, 0] = ((1 - fc) * gv * m).mean(dim=(1, 2))
        out[vi, :, 1] = (fc * gv * m).mean(dim=(1, 2))
    return out.float().numpy()


def fg_lut():
    """[256(roughness), 256(NoV), 2] float32; computed once and cached in-tree."""
    if os.path.exists(_CACHE):
        return np.load(_CACHE)
    lut = compute_fg_lut()
    os.makedirs(os.path.dirname(_CACHE), exist_ok=True)
    np.save(_CACHE, lut)
    return lut
