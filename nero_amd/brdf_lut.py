"""Split-sum BRDF table FG(NoV, roughness) used by the Stage-I shader.

The table is INPUT DATA of the reference: it ships as the binary asset assets/bsdf_256_256.bin, read relative to the working
directory at network/field.py:510 ([1, 256 (roughness, v), 256 (NoV, u), 2] float32), and reference checkpoints carry it as the
buffer `color_network.FG_LUT` (restored verbatim by load_state_dict).  `fg_lut()` therefore loads that asset, resolved in this
order: explicit `path` (shader_config key `fg_lut_path`) -> $NERO_FG_LUT -> `assets/bsdf_256_256.bin` relative to the working
directory (exactly what the reference does; nero_amd dropped into the reference tree reproduces it bit for bit).  Only when none
of these exists does it fall back -- with a loud warning -- to a table this module *computes* (GGX importance map,
height-correlated Smith G2, alpha = roughness^2, piecewise Gauss-Legendre quadrature in float64, `_fg_row`).  The computed table is
not the reference's file, but since round 4 it is the same FUNCTION: measured |computed - asset| is 6.0e-5 mean, 3.9e-4 max (the
asset's own deviation; rounds 1-3: 5.0e-4 / 2.25e-2 from an unconverged midpoint rule), tests/test_fg_lut.py pins both facts.
Round 6 tried to reproduce the asset's generator: the common recipes (Hammersley-sampled GGX split sum after Karis / Filament's DFV with
height-correlated or Schlick-Smith visibility, alpha = roughness or roughness^2, texel centres or corners, 256 ... 4096 samples; stratified grids
up to 1024 x 64; 2^18 random samples) all sit 4e-3 ... 1e-2 from the asset at grazing NoV (the 1 / NoV tail of the estimator), while the asset's
residual against the converged integral is SMOOTH (lag-1 correlation 0.97 along NoV, 0.82 along roughness, mean +1.0e-4 in the scale term): a
systematic offset of whatever produced it, not the noise of a small sample set.  Without the generator the asset cannot be regenerated to 1e-5;
what the product does instead is find the reference's own file (above) -- bench.py and the at-size tests point NERO_FG_LUT at the committed
fixture that holds it bit for bit.
"""
import os

import numpy as np

_CACHE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'assets', 'fg_lut_256.npy')


def _smith_lambda(c, a):
    return (-1.0 + np.sqrt(1.0 + a * a * (1.0 - c * c) / (c * c))) / 2.0


def _fg_row(a, nov, n_outer=96, n_inner=64):
    """A(NoV), B(NoV) of  integral f_r cos = F0 A + B  for one alpha = roughness^2 (float64).

    Half-vector space with the GGX importance map xi -> theta_h (the D peak is in the measure, so low roughness is exact: row 0
    agrees with the reference's table to 6e-5).  The integrand's support NoL > 0 is  phi < phi_max(theta_h) = arccos(-cot(theta_v')
    cot(2 theta_h))  in closed form (theta_v' = elevation of v): theta_h < beta/2 sees every phi, theta_h > pi/2 - beta/2 none.  Both
    integrals therefore run over pieces on which the integrand is smooth -- Gauss-Legendre in phi over (0, phi_max), Gauss-Legendre in
    xi over the two pieces, the second through a cosine substitution that absorbs the square-root behaviour of phi_max at its ends.
    (Round 1-3 used a midpoint rule over the whole square: the jump at NoL = 0 made it converge like n^-0.6, and the cached table was
    2.25e-2 off the reference's at grazing NoV / high roughness; this one is converged to 1e-5 with 96 x 64 nodes.)"""
    from numpy.polynomial.legendre import leggauss
    nov = nov[:, None, None]
    vx = np.sqrt(1.0 - nov * nov)
    beta = np.arctan2(nov, vx)

    def xi_of(theta):
        c2 = np.cos(theta) ** 2
        return (1.0 - c2) / (1.0 + c2 * (a * a - 1.0))

    x, w = leggauss(n_outer)
    t, wt = 0.5 * (x + 1.0), 0.5 * w
    xi_a, xi_b = xi_of(beta / 2.0), xi_of(np.pi / 2.0 - beta / 2.0)
    xi_i, wi = leggauss(n_inner)
    u, wu = 0.5 * (xi_i + 1.0), 0.5 * wi
    A = np.zeros(nov.shape[0])
    B = np.zeros(nov.shape[0])
    tiny = 1e-300
    for piece in (0, 1):
        if piece == 0:
            xi, wxi = xi_a * t[None, :, None], xi_a * wt[None, :, None]
        else:
            sub = 0.5 * (1.0 - np.cos(np.pi * t))[None, :, None]
            dsub = (0.5 * np.pi * np.sin(np.pi * t) * wt)[None, :, None]
            xi, wxi = xi_a + (xi_b - xi_a) * sub, (xi_b - xi_a) * dsub
        cos_t = np.sqrt((1.0 - xi) / (1.0 + (a * a - 1.0) * xi))
        sin_t = np.sqrt(np.maximum(1.0 - cos_t * cos_t, 0.0))
        if piece == 0:
            pmax = np.full_like(xi, np.pi)
        else:
            pmax = np.arccos(np.clip(-(nov / vx) * (2.0 * cos_t * cos_t - 1.0) / np.maximum(2.0 * sin_t * cos_t, tiny), -1.0, 1.0))
        phi, wphi = pmax * u[None, None, :], pmax * wu[None, None, :] / np.pi
        hx, hz = sin_t * np.cos(phi), cos_t
        voh = vx * hx + nov * hz
        nol = np.maximum(2.0 * voh * hz - nov, tiny)
        with np.errstate(over='ignore', divide='ignore'):
            g2 = 1.0 / (1.0 + _smith_lambda(nov, a) + _smith_lambda(nol, a))        # height-correlated Smith
        vohc = np.clip(voh, 0.0, 1.0)
        gv = g2 * vohc / (np.maximum(hz, tiny) * nov)
        fc = (1.0 - vohc) ** 5
        wgt = wxi * wphi
        A += ((1.0 - fc) * gv * wgt).sum(axis=(1, 2))
        B += (fc * gv * wgt).sum(axis=(1, 2))
    return A, B


def compute_fg_lut(res=256, n_outer=96, n_inner=64):
    """[res (roughness, v), res (NoV, u), 2] float32, texel centres (i + 0.5) / res like the reference's asset; about a minute of numpy"""
    nov = (np.arange(res, dtype=np.float64) + 0.5) / res
    out = np.zeros((res, res, 2), dtype=np.float64)
    for vi in range(res):
        out[vi, :, 0], out[vi, :, 1] = _fg_row(float(((vi + 0.5) / res) ** 2), nov, n_outer, n_inner)
    return out.astype(np.float32)


REFERENCE_ASSET = os.path.join('assets', 'bsdf_256_256.bin')      # network/field.py:510, relative to the working directory


def load_asset(path):
    """-> [256,256,2] float32 from a raw little-endian float32 file (the reference's format) or an .npy / .npz('lut')"""
    if path.endswith('.npz'):
        return np.ascontiguousarray(np.load(path)['lut'], dtype=np.float32).reshape(256, 256, 2)
    if path.endswith('.npy'):
        return np.ascontiguousarray(np.load(path), dtype=np.float32).reshape(256, 256, 2)
    return np.fromfile(path, dtype=np.float32).reshape(256, 256, 2)


def resolve_asset(path=None):
    """-> path of the reference FG table or None (see module docstring for the order)"""
    if path:
        if not os.path.exists(path):
            raise FileNotFoundError(f'shader_config.fg_lut_path = {path!r} does not exist')
        return path
    for cand in (os.environ.get('NERO_FG_LUT'), REFERENCE_ASSET):
        if cand and os.path.exists(cand):
            return cand
    return None


def computed_fg_lut():
    """[256,256,2] float32: the table this module integrates itself (cached in-tree)"""
    if os.path.exists(_CACHE):
        return np.load(_CACHE)
    lut = compute_fg_lut()
    os.makedirs(os.path.dirname(_CACHE), exist_ok=True)
    np.save(_CACHE, lut)
    return lut


def fg_lut(path=None):
    """[256(roughness), 256(NoV), 2] float32: the reference asset when it can be found, else the computed table (warns)."""
    found = resolve_asset(path)
    if found is not None:
        return load_asset(found)
    import warnings
    warnings.warn('nero_amd: the reference FG table assets/bsdf_256_256.bin was not found (working directory, $NERO_FG_LUT, '
                  'shader_config.fg_lut_path); falling back to a COMPUTED split-sum table (the same integral, within 4e-4 of the '
                  'asset everywhere).  Run from the reference tree, set NERO_FG_LUT, or load a reference checkpoint (its FG_LUT buffer '
                  'replaces this table).', RuntimeWarning, stacklevel=2)
    return computed_fg_lut()
