"""CPU oracle: a functional restatement of the NeRO Stage-I render step (liuyuan-pal/NeRO).

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module, and only as the checker -- never as the thing that is measured or shipped.  The product path
(nero_amd/) never imports it and fails loudly when the HIP library is missing.

Parity pin: the reference holds no tests or golden vectors for this path, so this oracle is
pinned against outputs of the *unmodified reference itself*, executed in the build container through
oracle/ref_shim.py by oracle/gen_golden.py; the dumped vectors live in tests/golden/*.npz and
tests/test_oracle_golden.py checks this file against them (fp32, rel <= 2e-5 on outputs, 2e-4 on parameter
grads; sample indices identical on the stage-wise teacher-forced vectors).

Differences from the reference that are deliberate (all documented in DESIGN.md):
  * every random draw (coarse jitter, background jitter, occ-loss subset keys) is an INPUT, never drawn here;
  * scans are stated with an explicit order: sequential cumsum / cumprod with a float64 running value rounded to
    float32 per element (this is what torch-CPU does for float32 tensors), and the pdf normaliser is the last
    element of that cumsum rather than an order-unspecified torch.sum;
  * the concat-sort in cat_z_vals is stable (ties keep the original position);
  * IDE uses repeated complex multiplication with z**0 == 1, so it has no NaN at the poles (the reference's
    complex pow gives NaN at x=y=0, utils/ref_utils.py:104);
  * the occ-loss subset is "the occ_loss_max_pn smallest keys" of a per-candidate uniform key tensor (the reference
    uses torch.randperm, network/renderer.py:535-541 -- same distribution, different bits).

All functions are pure: parameters come in a flat dict P of EFFECTIVE weights (see effective_params()).
All file:line citations are relative to /root/reference.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------------------------------------
# parameters
# ----------------------------------------------------------------------------------------------------------------------

def weight_norm_effective(v, g):
    """nn.utils.weight_norm default dim=0: W = g * v / ||v||_row  (network/field.py:118-119, 323-331)."""
    return v * (g / v.norm(dim=1, keepdim=True))


def effective_params(state_dict):
    """state_dict with *.weight_g/*.weight_v/*.bias  ->  {'name.weight': W, 'name.bias': b, ...} (autograd-friendly)."""
    P = {}
    for k, t in state_dict.items():
        if k.endswith('.weight_g'):
            base = k[:-len('.weight_g')]
            P[base + '.weight'] = weight_norm_effective(state_dict[base + '.weight_v'], t)
        elif k.endswith('.weight_v'):
            continue
        else:
            P[k] = t
    return P


# ----------------------------------------------------------------------------------------------------------------------
# encodings
# ----------------------------------------------------------------------------------------------------------------------

def pos_enc(x, n_freq):
    """[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)]   (network/field.py:14-58)."""
    out = [x]
    for k in range(n_freq):
        f = float(2 ** k)
        out.append(torch.sin(x * f))
        out.append(torch.cos(x * f))
    return torch.cat(out, -1)


def _gen_binom(a, k):
    return float(np.prod(a - np.arange(k))) / math.factorial(k)


def _assoc_legendre_coeff(l, m, k):
    return ((-1) ** m * 2 ** l * math.factorial(l) / math.factorial(k) / math.factorial(l - k - m)
            * _gen_binom(0.5 * (l + k + m - 1.0), l))


def _sph_harm_coeff(l, m, k):
    return math.sqrt((2.0 * l + 1.0) * math.factorial(l - m) / (4.0 * math.pi * math.factorial(l + m))) \
        * _assoc_legendre_coeff(l, m, k)


def ide_tables(deg_view=5):
    """(m_list[36], l_list[36], mat[17,36] float32)   (utils/ref_utils.py:41-82)."""
    ml = []
    for i in range(deg_view):
        l = 2 ** i
        for m in range(l + 1):
            ml.append((m, l))
    l_max = 2 ** (deg_view - 1)
    mat = np.zeros((l_max + 1, len(ml)))
    for i, (m, l) in enumerate(ml):
        for k in range(l - m + 1):
            mat[k, i] = _sph_harm_coeff(l, m, k)
    return [m for m, _ in ml], [l for _, l in ml], mat.astype(np.float32)


_IDE_M, _IDE_L, _IDE_MAT = ide_tables(5)


def ide(dirs, kappa_inv):
    """Integrated directional encoding, deg 5 -> 72 channels [Re(36), Im(36)]   (utils/ref_utils.py:85-117).

    kappa_inv: [...,1] tensor or python float.  (x+iy)^m by repeated multiplication, (x+iy)^0 == 1."""
    x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
    mat = torch.as_tensor(_IDE_MAT, dtype=dirs.dtype, device=dirs.device)
    zp = [torch.ones_like(z)]
    for _ in range(mat.shape[0] - 1):
        zp.append(zp[-1] * z)
    vmz = torch.cat(zp, -1)                                    # [...,17]
    poly = vmz @ mat                                           # [...,36]
    re, im = [torch.ones_like(x)], [torch.zeros_like(x)]
    for _ in range(max(_IDE_M)):
        r, i = re[-1], im[-1]
        re.append(r * x - i * y)
        im.append(r * y + i * x)
    pre = torch.cat([re[m] for m in _IDE_M], -1)
    pim = torch.cat([im[m] for m in _IDE_M], -1)
    sigma = torch.as_tensor([0.5 * l * (l + 1) for l in _IDE_L], dtype=dirs.dtype, device=dirs.device)
    att = torch.exp(-sigma * kappa_inv) if torch.is_tensor(kappa_inv) else \
        torch.exp(-sigma * float(kappa_inv)).expand_as(poly)
    return torch.cat([pre * poly * att, pim * poly * att], -1)


def linear_to_srgb(x):
    """utils/raw_utils.py:4-10."""
    eps = torch.finfo(torch.float32).eps
    lo = 323 / 25 * x
    hi = (211 * torch.clamp(x, min=eps) ** (5 / 12) - 11) / 200
    return torch.where(x <= 0.0031308, lo, hi)


def fg_lut_fetch(lut, u, v):
    """bilinear clamp fetch of FG_LUT[1,256(v),256(u),2] at (u,v) in [0,1]; texel centres at (i+.5)/256
    (network/field.py:610-613; nvdiffrast convention, restated -- third-party, parity unpinned)."""
    H, W = lut.shape[1], lut.shape[2]
    uu = torch.clamp(u * W - 0.5, 0.0, W - 1.0)
    vv = torch.clamp(v * H - 0.5, 0.0, H - 1.0)
    u0 = torch.clamp(torch.floor(uu), max=W - 2.0)
    v0 = torch.clamp(torch.floor(vv), max=H - 2.0)
    fu, fv = (uu - u0).unsqueeze(-1), (vv - v0).unsqueeze(-1)
    u0, v0 = u0.long(), v0.long()
    t = lut[0]
    return (t[v0, u0] * (1 - fu) + t[v0, u0 + 1] * fu) * (1 - fv) + (t[v0 + 1, u0] * (1 - fu) + t[v0 + 1, u0 + 1] * fu) * fv


# ----------------------------------------------------------------------------------------------------------------------
# networks
# ----------------------------------------------------------------------------------------------------------------------

def deviation_inv_s(P, act='exp'):
    """SingleVarianceNetwork.forward (network/field.py:190-198): exp(10 v) | 10 v | (10 v)^2"""
    v10 = P['deviation_network.variance'] * 10.0
    if act == 'exp':
        return torch.exp(v10)
    if act == 'linear':
        return v10
    if act == 'square':
        return v10 ** 2
    raise NotImplementedError(act)


def softplus100(x):
    return F.softplus(x, beta=100)          # threshold 20 (network/field.py:124)


def sdf_network(P, x, prefix='sdf_network'):
    """softplus(beta=100) MLP of sdf_n_layers + 1 linear layers on a PE-f input, the input re-injected in front of layer
    sdf_n_layers // 2   (network/field.py:75-101, 130-147; network/renderer.py:118-124).  Every shipped YAML: 9 layers, PE-6, skip into
    layer 4; the shape is read off the weights.  -> [N,257]"""
    n_lin = sum(1 for k in P if k.startswith(f'{prefix}.lin') and k.endswith('.weight'))
    n_freq = (P[f'{prefix}.lin0.weight'].shape[1] - 3) // 6
    skip = (n_lin - 1) // 2
    e = pos_enc(x, n_freq)
    h = e
    for l in range(n_lin):
        if l == skip:
            h = torch.cat([h, e], -1) / math.sqrt(2)
        h = F.linear(h, P[f'{prefix}.lin{l}.weight'], P[f'{prefix}.lin{l}.bias'])
        if l < n_lin - 1:
            h = softplus100(h)
    return h


def sdf_value_and_normal(P, x):
    """forward + d sdf / d x with a differentiable graph   (network/field.py:155-167; renderer.py:486-490).
    One forward serves both (the reference evaluates the network twice; SURVEY.md §0 finding 5)."""
    x = x.detach().requires_grad_(True)
    with torch.enable_grad():
        y = sdf_network(P, x)
        (g,) = torch.autograd.grad(y[..., 0].sum(), x, create_graph=True)
    return y, g


# ---- forced ReLU gates (a TEST hook: tests/test_parity_at_size.py, "gate-teacher-forced" gradient parity) --------------------------
# A ReLU unit whose pre-activation sits within rounding of zero falls on either side depending on the last bit of a 256-term dot
# product; which side decides whether that row contributes to a whole column of a weight gradient.  Two correct fp32 evaluations --
# and the fp64 one -- therefore disagree by 1e-3 on individual gradient tensors at 300 k rows without either being wrong.  To compare
# ARITHMETIC rather than tie-breaking, a test can hand this oracle the gate decisions another evaluation took (the sign masks the HIP
# forward kernels save, nero_fwd_layer.relu_mask): inside `with forced_relu_gates({...})` every ReLU named in the dict computes
# h * gate instead of relu(h).  Keys: '<predictor prefix>@<call index>/<layer>' (call index: outer_light is evaluated twice, diffuse
# then specular), 'outer_nerf/pts<i>', 'outer_nerf/views'.  Outside the context manager nothing changes.
_FORCED = None
_CAPTURE = None
_CALLS = {}


class forced_relu_gates:
    def __init__(self, gates):
        self.gates = gates

    def __enter__(self):
        global _FORCED
        _FORCED = self.gates
        _CALLS.clear()
        self.used = set()
        _CALLS['__used__'] = self.used
        return self

    def __exit__(self, *a):
        global _FORCED
        _FORCED = None
        _CALLS.clear()


class capture_relu_gates(forced_relu_gates):
    """records the decisions instead of forcing them: after the `with` block `.gates` maps every key to the 0/1 gate (ReLU) or the sign
    (L1 terms) THIS evaluation took -- e.g. a float64 run, whose decisions a float32 run is then handed through forced_relu_gates
    (tests/test_oracle_golden.py: float32 ARITHMETIC against float64 without the tie-breaking noise)"""

    def __init__(self):
        super().__init__({})

    def __enter__(self):
        global _CAPTURE
        r = super().__enter__()
        _CAPTURE = self.gates
        return r

    def __exit__(self, *a):
        global _CAPTURE
        _CAPTURE = None
        super().__exit__(*a)


def _relu(h, key):
    if _CAPTURE is not None:
        _CAPTURE[key] = (h.detach() > 0).to(torch.float32)
    if _FORCED is None or key not in _FORCED:
        return F.relu(h)
    g = _FORCED[key]
    assert g.shape == h.shape, (key, tuple(g.shape), tuple(h.shape))
    _CALLS['__used__'].add(key)
    return h * g.to(device=h.device, dtype=h.dtype)


def _abs(x, key):
    """|x| -- or, inside forced_relu_gates with `key` present, x * sign taken from another evaluation (an L1 term whose argument sits
    within rounding of zero is the same kind of tie as a ReLU at zero: one flipped sign of the diffuse-light neutrality term of ONE
    surface point of 4096 moves every outer-light gradient by 1e-4, scripts/r04/diag_outer_light.py)"""
    if _CAPTURE is not None:
        _CAPTURE[key] = torch.where(x.detach() >= 0, 1.0, -1.0).to(torch.float32)
    if _FORCED is None or key not in _FORCED:
        return torch.abs(x)
    g = _FORCED[key]
    assert g.shape == x.shape, (key, tuple(g.shape), tuple(x.shape))
    _CALLS['__used__'].add(key)
    return x * g.to(device=x.device, dtype=x.dtype)


def next_call(prefix):
    """index of this evaluation of the MLP `prefix` inside the current forced_relu_gates context (0 outside one)"""
    if _FORCED is None:
        return 0
    call = _CALLS.get(prefix, 0)
    _CALLS[prefix] = call + 1
    return call


def predictor(P, prefix, x, out_act):
    """make_predictor: 3x(Linear 256 + ReLU) + Linear + activation   (network/field.py:310-346)."""
    call = next_call(prefix)
    h = x
    for i, l in enumerate((0, 2, 4, 6)):
        h = F.linear(h, P[f'{prefix}.{l}.weight'], P[f'{prefix}.{l}.bias'])
        if i < 3:
            h = _relu(h, f'{prefix}@{call}/{i}')
    return out_act(h)


def nerfpp(P, pts4, views, prefix='outer_nerf'):
    """NeRF++ background MLP (network/field.py:258-283): PE-10 of [p/|p|, 1/|p|], 8x256 ReLU, skip cat [pe, h] after
    layer 4; sigma head; feature ++ PE-4(view) -> 128 -> rgb."""
    e = pos_enc(pts4, 10)
    ev = pos_enc(views, 4)
    h = e
    for i in range(8):
        h = _relu(F.linear(h, P[f'{prefix}.pts_linears.{i}.weight'], P[f'{prefix}.pts_linears.{i}.bias']), f'{prefix}/pts{i}')
        if i == 4:
            h = torch.cat([e, h], -1)
    sigma = F.linear(h, P[f'{prefix}.alpha_linear.weight'], P[f'{prefix}.alpha_linear.bias'])
    feat = F.linear(h, P[f'{prefix}.feature_linear.weight'], P[f'{prefix}.feature_linear.bias'])
    h = _relu(F.linear(torch.cat([feat, ev], -1), P[f'{prefix}.views_linears.0.weight'], P[f'{prefix}.views_linears.0.bias']), f'{prefix}/views')
    rgb = F.linear(h, P[f'{prefix}.rgb_linear.weight'], P[f'{prefix}.rgb_linear.bias'])
    return sigma, rgb


# ----------------------------------------------------------------------------------------------------------------------
# sampling (explicit scan order)
# ----------------------------------------------------------------------------------------------------------------------

def seq_cumsum(x):
    """float64 running sum, rounded to x.dtype per element (what torch-CPU cumsum does for float32)."""
    return torch.cumsum(x.double(), -1).to(x.dtype)


def seq_cumprod(x):
    return torch.cumprod(x.double(), -1).to(x.dtype)


def transmittance_weights(alpha):
    """w_i = alpha_i * prod_{j<i} (1 - alpha_j + 1e-7)   (network/renderer.py:381-382, 578; field.py:450)."""
    t = seq_cumprod(1.0 - alpha + 1e-7)
    t = torch.cat([torch.ones_like(t[..., :1]), t[..., :-1]], -1)
    return alpha * t


def sample_pdf_det(bins, weights, n_samples):
    """Deterministic inverse-CDF sampling (network/field.py:399-429, det=True).  -> samples [R,n], inds int64 [R,n]"""
    w = weights + 1e-5
    c = seq_cumsum(w)
    pdf_norm = c[..., -1:]
    cdf = torch.cat([torch.zeros_like(c[..., :1]), seq_cumsum(w / pdf_norm)], -1)           # [R, nb]
    u = torch.linspace(0.5 / n_samples, 1.0 - 0.5 / n_samples, n_samples, dtype=bins.dtype, device=bins.device)
    u = u.expand(list(cdf.shape[:-1]) + [n_samples]).contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    c0, c1 = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    b0, b1 = torch.gather(bins, -1, below), torch.gather(bins, -1, above)
    denom = c1 - c0
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - c0) / denom
    return b0 + t * (b1 - b0), inds


def upsample_weights(o, d, z, sdf, inv_s):
    """NeuS section weights along the ray for importance sampling (network/renderer.py:355-382). inv_s: float."""
    pts = o[:, None, :] + d[:, None, :] * z[..., None]
    radius = torch.linalg.norm(pts, dim=-1)
    inside = (radius[:, :-1] < 1.0) | (radius[:, 1:] < 1.0)
    ps, ns = sdf[:, :-1], sdf[:, 1:]
    dz = z[:, 1:] - z[:, :-1]
    mid = (ps + ns) * 0.5
    cos = (ns - ps) / (dz + 1e-5)
    prev = torch.cat([torch.zeros_like(cos[:, :1]), cos[:, :-1]], -1)
    cos = torch.minimum(prev, cos).clamp(-1e3, 0.0) * inside
    pe_, ne_ = mid - cos * dz * 0.5, mid + cos * dz * 0.5
    pc, nc = torch.sigmoid(pe_ * inv_s), torch.sigmoid(ne_ * inv_s)
    alpha = (pc - nc + 1e-5) / (pc + 1e-5)
    return transmittance_weights(alpha)


def merge_sorted(z, sdf, z_new, sdf_new):
    """concat + stable sort + permute sdf (network/renderer.py:387-401).  -> z, sdf (or None), index int64"""
    zc = torch.cat([z, z_new], -1)
    zs, index = torch.sort(zc, dim=-1, stable=True)
    s = None
    if sdf_new is not None:
        s = torch.gather(torch.cat([sdf, sdf_new], -1), -1, index)
    return zs, s, index


def coarse_z(cfg, near, far, rand1):
    n = cfg['n_samples']
    t = torch.linspace(0.0, 1.0, n, dtype=near.dtype, device=near.device)
    z = near + (far - near) * t[None, :]
    if rand1 is not None:
        z = z + (rand1 - 0.5) * 2.0 / n
    return z


def background_z(cfg, far, rand_bg):
    nb = cfg['n_bg_samples']
    zo = torch.linspace(1e-3, 1.0 - 1.0 / (nb + 1.0), nb, dtype=far.dtype, device=far.device)
    if rand_bg is not None:
        mids = 0.5 * (zo[1:] + zo[:-1])
        upper = torch.cat([mids, zo[-1:]], -1)
        lower = torch.cat([zo[:1], mids], -1)
        zo = lower[None, :] + (upper - lower)[None, :] * rand_bg
    else:
        zo = zo[None, :]
    return far / torch.flip(zo, dims=[-1]) + 1.0 / nb


def sample_ray(P, cfg, o, d, near, far, rand1=None, rand_bg=None, trace=None):
    """network/renderer.py:403-443.  rand1 [R,1], rand_bg [R,n_bg] in [0,1) or None (perturb == 0).
    trace: optional list receiving per-round dicts (z, sdf, weights, z_new, inds, index) for teacher-forced tests."""
    up = cfg['up_sample_steps']
    n_imp = cfg['n_importance'] // up
    z = coarse_z(cfg, near, far, rand1)
    zbg = background_z(cfg, far, rand_bg)
    with torch.no_grad():
        R = z.shape[0]
        pts = o[:, None, :] + d[:, None, :] * z[..., None]
        sdf = sdf_network(P, pts.reshape(-1, 3))[:, 0].reshape(R, -1)
        inv_s_net = float(deviation_inv_s(P, cfg.get('std_act', 'exp')))
        for i in range(up):
            inv_s = min(inv_s_net, 64.0 * 2 ** i) if cfg['clip_sample_variance'] else 64.0 * 2 ** i
            w = upsample_weights(o, d, z, sdf, inv_s)
            z_new, inds = sample_pdf_det(z, w, n_imp)
            last = (i + 1 == up)
            sdf_new = None
            if not last:
                pn = o[:, None, :] + d[:, None, :] * z_new[..., None]
                sdf_new = sdf_network(P, pn.reshape(-1, 3))[:, 0].reshape(R, -1)
            z2, sdf2, index = merge_sorted(z, sdf, z_new, sdf_new)
            if trace is not None:
                trace.append(dict(z=z, sdf=sdf, inv_s=inv_s, weights=w, z_new=z_new, inds=inds, sdf_new=sdf_new,
                                  index=index, z_out=z2))
            z, sdf = z2, sdf2
    return torch.cat([z, zbg], -1)


# ----------------------------------------------------------------------------------------------------------------------
# shading
# ----------------------------------------------------------------------------------------------------------------------

def _exp_act(mx):
    return lambda x: torch.exp(torch.clamp(x, max=mx))


def camera_plane_intersection(pts, dirs, poses):
    """network/field.py:348-367 (the in-place write through a view is reproduced: dirs_ z is patched before use)."""
    R_, t = poses[:, :, :3], poses[:, :, 3:]
    p = (R_ @ pts[:, :, None] + t)[..., 0]
    dd = (R_ @ dirs[:, :, None])[..., 0]
    hits = torch.abs(dd[..., 2]) > 1e-4
    dz = torch.where(hits, dd[..., 2], torch.full_like(dd[..., 2], 1e-4))
    dd = torch.cat([dd[..., :2], dz[..., None]], -1)
    dist = -p[:, 2] / dz
    inter = p + dist.unsqueeze(-1) * dd
    return inter, dist, hits


def ipe(mean, var, min_deg, max_deg):
    """network/field.py:369-378."""
    scales = 2.0 ** torch.arange(min_deg, max_deg, dtype=mean.dtype, device=mean.device)
    shape = mean.shape[:-1] + (-1,)
    sm = torch.reshape(mean[..., None, :] * scales[:, None], shape)
    sv = torch.reshape(var[..., None, :] * scales[:, None] ** 2, shape)
    return torch.exp(-0.5 * torch.cat([sv, sv], -1)) * torch.sin(torch.cat([sm, sm + 0.5 * np.pi], -1))


def human_light(P, pts, refl, poses, rough):
    """network/field.py:536-552."""
    inter, dists, hits = camera_plane_intersection(pts, refl, poses)
    mean = inter[..., :2] * 0.3
    var = rough * (dists[:, None] * 0.3) ** 2
    hits = (hits & (torch.norm(mean, dim=-1) < 1.5) & (dists > 0)).float().unsqueeze(-1)
    mean, var = mean * hits, (var * hits).expand(mean.shape[0], 2)
    hl = predictor(P, 'color_network.human_light_predictor', ipe(mean, var, 0, 6), _exp_act(0.0)) * hits
    return hl[..., :3], torch.clamp(hl[..., 3:], 0.0, 1.0)


def offset_points_to_sphere(points):
    """network/field.py:380-388: points outside radius 0.999 are pulled back onto that sphere."""
    norm = torch.norm(points, dim=-1, keepdim=True)
    return torch.where(norm > 0.999, points / norm * 0.999, points)


def sphere_exit_dist(pts, dirs):
    """network/field.py:390-396."""
    dtx = torch.sum(pts * dirs, -1, keepdim=True)
    xtx = torch.sum(pts ** 2, -1, keepdim=True)
    return -dtx + torch.sqrt(dtx ** 2 - xtx + 1 + 1e-6)


def app_shading(P, scfg, pts, grads, view, feat, poses, want_inter=False):
    """AppShadingNetwork.forward, split-sum shading (network/field.py:591-651), incl. shader_config.sphere_direction
    (predict_specular_lights :558-562, predict_diffuse_lights :582-586)."""
    sphere = scfg.get('sphere_direction', False)

    def outer_in(direction, roughness):
        enc = ide(direction, roughness)
        if not sphere:
            return enc
        q = offset_points_to_sphere(pts)
        sph = F.normalize(q + direction * sphere_exit_dist(q, direction), dim=-1)
        return torch.cat([enc, ide(sph, roughness)], -1)
    exp_max = scfg.get('light_exp_max', 0.0)
    n = F.normalize(grads, dim=-1)
    v = F.normalize(view, dim=-1)
    nov = torch.sum(n * v, -1, keepdim=True)
    refl = nov * n * 2 - v
    fx = torch.cat([feat, pts], -1)
    metallic = predictor(P, 'color_network.metallic_predictor', fx, torch.sigmoid)
    rough = predictor(P, 'color_network.roughness_predictor', fx, torch.sigmoid)
    albedo = predictor(P, 'color_network.albedo_predictor', fx, torch.sigmoid)

    diff_albedo = (1 - metallic) * albedo
    diff_light = predictor(P, 'color_network.outer_light', outer_in(n, 1.0), _exp_act(exp_max))
    diff_color = diff_albedo * diff_light

    spec_albedo = 0.04 * (1 - metallic) + metallic * albedo
    enc_r = ide(refl, rough)
    enc_p = pos_enc(pts, scfg.get('light_pos_freq', 8))
    direct = predictor(P, 'color_network.outer_light', outer_in(refl, rough), _exp_act(exp_max))
    hl, hw = 0, 0
    if scfg.get('human_light', False):
        hl, hw = human_light(P, pts, refl, poses, rough)
    indirect = predictor(P, 'color_network.inner_light', torch.cat([enc_p, enc_r], -1), _exp_act(exp_max))
    occ = predictor(P, 'color_network.inner_weight', torch.cat([enc_p.detach(), pos_enc(refl, 6).detach()], -1), lambda t: t)
    occ = occ * 0.5 + 0.5
    occ_c = torch.clamp(occ, 0.0, 1.0)
    spec_light = indirect * occ_c + (hl * hw + direct * (1 - hw)) * (1 - occ_c)

    fg = fg_lut_fetch(P['color_network.FG_LUT'], torch.clamp(nov[:, 0], 0.0, 1.0), torch.clamp(rough[:, 0], 0.0, 1.0))
    spec_ref = spec_albedo * fg[:, 0:1] + fg[:, 1:2]
    spec_color = spec_ref * spec_light
    color = torch.clamp(linear_to_srgb(diff_color + spec_color), 0.0, 1.0)
    occ_info = {'reflective': refl, 'occ_prob': occ}
    if not want_inter:
        return color, occ_info
    inter = {
        'specular_albedo': spec_albedo, 'specular_ref': torch.clamp(spec_ref, 0, 1),
        'specular_light': torch.clamp(linear_to_srgb(spec_light), 0, 1),
        'specular_color': torch.clamp(linear_to_srgb(spec_color), 0, 1),
        'diffuse_albedo': diff_albedo, 'diffuse_light': torch.clamp(linear_to_srgb(diff_light), 0, 1),
        'diffuse_color': torch.clamp(linear_to_srgb(diff_color), 0, 1),
        'metallic': metallic, 'roughness': rough, 'occ_prob': occ_c, 'indirect_light': indirect * occ_c,
    }
    if scfg.get('human_light', False):
        inter['human_light'] = linear_to_srgb(hl * hw)
    return color, occ_info, inter


# ----------------------------------------------------------------------------------------------------------------------
# secondary-ray occlusion search (occ loss / validation)
# ----------------------------------------------------------------------------------------------------------------------

def section_weights(P, z, origins, dirs, inv_s):
    """get_weights (network/field.py:432-452)."""
    pts = z.unsqueeze(-1) * dirs.unsqueeze(-2) + origins.unsqueeze(-2)
    sdf = sdf_network(P, pts.reshape(-1, 3))[:, 0].reshape(z.shape)
    ps, ns = sdf[:, :-1], sdf[:, 1:]
    dz = z[:, 1:] - z[:, :-1]
    mid = (ps + ns) * 0.5
    cos = (ns - ps) / (dz + 1e-5)
    surf = cos < 0
    cos = torch.clamp(cos, max=0)
    pc = torch.sigmoid((mid - cos * dz * 0.5) * inv_s)
    nc = torch.sigmoid((mid + cos * dz * 0.5) * inv_s)
    alpha = (pc - nc + 1e-5) / (pc + 1e-5) * surf.float()
    return transmittance_weights(alpha), torch.where(surf, mid, -torch.ones_like(mid))


def secondary_ray_occlusion(P, pts, dirs, sn0, sn1, std_act='exp'):
    """get_intersection (network/field.py:454-484), returns sum of the sn1-1 section weights = occ probability.
    Points with |p| >= 0.999 get 0."""
    inside = torch.norm(pts, dim=-1) < 0.999
    out = torch.zeros(pts.shape[0], 1, dtype=pts.dtype, device=pts.device)
    if int(inside.sum()) == 0:
        return out
    p, dd = pts[inside], dirs[inside]
    with torch.no_grad():
        inv_s = float(deviation_inv_s(P, std_act))
        maxd = sphere_exit_dist(p, dd)
        z = maxd * torch.linspace(0, 1, sn0, dtype=p.dtype, device=p.device).unsqueeze(0)
        w, _ = section_weights(P, z, p, dd, inv_s)
        z_new, _ = sample_pdf_det(z, w, sn1)
        w, _ = section_weights(P, z_new, p, dd, inv_s)
    out[inside] = torch.sum(w, -1, keepdim=True)
    return out


# ----------------------------------------------------------------------------------------------------------------------
# render
# ----------------------------------------------------------------------------------------------------------------------

DEFAULT_CFG = {
    'n_samples': 64, 'n_bg_samples': 32, 'n_importance': 64, 'up_sample_steps': 4, 'perturb': 1.0,
    'anneal_end': 50000, 'clip_sample_variance': True, 'freeze_inv_s_step': None, 'apply_occ_loss': True,
    'occ_loss_step': 20000, 'occ_loss_max_pn': 2048, 'occ_sdf_thresh': 0.01, 'rgb_loss': 'charbonier', 'std_act': 'exp',
    'shader_config': {},
}


def render_core(P, cfg, o, d, z_vals, poses, cos_anneal, step, occ_keys=None):
    """network/renderer.py:550-606 (training outputs).  occ_keys: uniform keys, one per occ-loss CANDIDATE in flat
    (ray, sample) order of the candidates; used only when more than occ_loss_max_pn surface samples qualify: the
    occ_loss_max_pn candidates with the smallest keys (ties: lower rank first) are kept."""
    cfg = {**DEFAULT_CFG, **cfg}
    R, T = z_vals.shape
    dists = z_vals[:, 1:] - z_vals[:, :-1]
    dists = torch.cat([dists, dists[:, -1:]], -1)
    mid = z_vals + dists * 0.5
    pts = (o[:, None, :] + d[:, None, :] * mid[..., None]).reshape(-1, 3)
    dirs = F.normalize(d, dim=-1)[:, None, :].expand(R, T, 3).reshape(-1, 3)
    pose_pt = poses[:, None].expand(R, T, 3, 4).reshape(-1, 3, 4)
    inner = (torch.norm(pts, dim=-1) <= 1.0)
    ii = torch.nonzero(inner)[:, 0]
    oi = torch.nonzero(~inner)[:, 0]
    alpha = torch.zeros(R * T, dtype=z_vals.dtype, device=z_vals.device)
    color = torch.zeros(R * T, 3, dtype=z_vals.dtype, device=z_vals.device)
    dflat = dists.reshape(-1)
    out = {}

    if oi.numel() > 0:                                                      # renderer.py:514-520
        po = pts[oi]
        nrm = torch.norm(po, dim=-1, keepdim=True)
        sigma, rgb = nerfpp(P, torch.cat([po / nrm, 1.0 / nrm], -1), -dirs[oi])
        a_o = 1.0 - torch.exp(-F.softplus(sigma[:, 0]) * dflat[oi])
        c_o = linear_to_srgb(torch.exp(torch.clamp(rgb, max=5.0)))
        alpha = alpha.index_put((oi,), a_o)
        color = color.index_put((oi,), c_o)

    if ii.numel() > 0:                                                      # renderer.py:484-512
        pi = pts[ii]
        y, grad = sdf_value_and_normal(P, pi)
        sdf, feat = y[:, 0], y[:, 1:]
        inv_s = deviation_inv_s(P, cfg.get('std_act', 'exp')).clamp(1e-6, 1e6)
        if cfg['freeze_inv_s_step'] is not None and step < cfg['freeze_inv_s_step']:
            inv_s = inv_s.detach()
        di = dirs[ii]
        true_cos = (di * grad).sum(-1)
        iter_cos = -(F.relu(-true_cos * 0.5 + 0.5) * (1.0 - cos_anneal) + F.relu(-true_cos) * cos_anneal)
        e_next = sdf + iter_cos * dflat[ii] * 0.5
        e_prev = sdf - iter_cos * dflat[ii] * 0.5
        pc, nc = torch.sigmoid(e_prev * inv_s), torch.sigmoid(e_next * inv_s)
        a_i = ((pc - nc + 1e-5) / (pc + 1e-5)).clamp(0.0, 1.0)
        c_i, occ_info = app_shading(P, cfg['shader_config'], pi, grad, -di, feat, pose_pt[ii])
        alpha = alpha.index_put((ii,), a_i)
        color = color.index_put((ii,), c_i)
        out['gradient_error'] = (torch.linalg.norm(grad, dim=-1) - 1.0) ** 2
        out['std'] = torch.mean(1.0 / inv_s.expand(ii.numel()))
    else:
        out['gradient_error'] = torch.zeros(1, dtype=z_vals.dtype, device=z_vals.device)
        out['std'] = torch.zeros(1, dtype=z_vals.dtype, device=z_vals.device)

    alpha = alpha.reshape(R, T)
    # compositing uses torch.cumprod over float32 here too (renderer.py:578); differentiable, so use torch.cumprod
    trans = torch.cumprod(torch.cat([torch.ones(R, 1, dtype=alpha.dtype, device=alpha.device), 1.0 - alpha + 1e-7], -1), -1)[:, :-1]
    weights = alpha * trans
    out['ray_rgb'] = (color.reshape(R, T, 3) * weights[..., None]).sum(1)
    out['weights'] = weights

    if step < 1000:                                                         # renderer.py:591-594
        m = torch.norm(pts, dim=-1) < 1.2
        out['sdf_pts'] = pts[m]
        out['sdf_vals'] = sdf_network(P, pts[m])[:, 0]

    if cfg['apply_occ_loss']:                                               # renderer.py:522-548, 596-601
        out['loss_occ'] = torch.zeros(1, dtype=z_vals.dtype, device=z_vals.device)
        if ii.numel() > 0 and step >= cfg['occ_loss_step']:
            m = (torch.norm(pi, dim=-1) < 0.999) & (torch.abs(sdf) < cfg['occ_sdf_thresh']) & ((grad * di).sum(-1) < 0)
            cand = torch.nonzero(m)[:, 0]
            if cand.numel() > cfg['occ_loss_max_pn']:
                keys = occ_keys[:cand.numel()]
                keep = torch.sort(torch.argsort(keys, stable=True)[:cfg['occ_loss_max_pn']])[0]
                cand = cand[keep]
            if cand.numel() > 0:
                gt = secondary_ray_occlusion(P, pi[cand].detach(), occ_info['reflective'][cand].detach(), 64, 16, cfg.get('std_act', 'exp'))
                out['loss_occ'] = F.l1_loss(occ_info['occ_prob'][cand], gt)
            out['occ_count'] = cand.numel()
    out['n_inner'] = ii.numel()
    out['n_outer'] = oi.numel()
    return out


def validation_info(P, cfg, o, d, z_vals, weights, poses):
    """compute_validation_info (network/renderer.py:465-482), inference only"""
    cfg = {**DEFAULT_CFG, **cfg}
    depth = torch.sum(weights * z_vals, -1, keepdim=True)
    pts = depth * d + o
    y, grad = sdf_value_and_normal(P, pts)
    inner = (torch.norm(pts, dim=-1, keepdim=True) <= 1.0).float()
    out = {'depth': depth, 'normal': ((F.normalize(grad, dim=-1) + 1.0) * 0.5) * inner}
    _, occ_info, inter = app_shading(P, cfg['shader_config'], pts, grad, -F.normalize(d, dim=-1), y[:, 1:], poses, want_inter=True)
    out['occ_prob_gt'] = secondary_ray_occlusion(P, pts.detach(), occ_info['reflective'].detach(), 128, 9, cfg.get('std_act', 'exp'))
    for k, v in inter.items():
        out[k] = v * inner
    return {k: v.detach() for k, v in out.items()}


def rgb_loss(cfg, pr, gt):
    """network/renderer.py:332-344."""
    kind = {**DEFAULT_CFG, **cfg}['rgb_loss']
    if kind == 'charbonier':
        return torch.sqrt(torch.sum((gt - pr) ** 2, -1) + 1e-3)
    if kind == 'l2':
        return torch.sum((pr - gt) ** 2, -1)
    if kind == 'l1':
        return torch.sum(torch.abs(pr - gt), -1)
    if kind == 'smooth_l1':
        return torch.sum(F.smooth_l1_loss(pr, gt, reduction='none', beta=0.25), -1)
    raise NotImplementedError


def near_far_from_sphere(o, d):
    """network/renderer.py:230-238."""
    a = torch.sum(d ** 2, -1, keepdim=True)
    b = 2.0 * torch.sum(o * d, -1, keepdim=True)
    mid = 0.5 * (-b) / a
    return torch.clamp(mid - 1.0, min=1e-3), mid + 1.0


def anneal(cfg, step):
    e = {**DEFAULT_CFG, **cfg}['anneal_end']
    return 1.0 if e < 0 else min(1.0, step / e)


def render(P, cfg, o, d, near, far, poses, step, cos_anneal, rand1=None, rand_bg=None, occ_keys=None):
    """NeROShapeRenderer.render (network/renderer.py:445-463), training outputs."""
    cfg = {**DEFAULT_CFG, **cfg}
    z = sample_ray(P, cfg, o, d, near, far, rand1, rand_bg)
    out = render_core(P, cfg, o, d, z, poses, cos_anneal, step, occ_keys)
    out['z_vals'] = z
    return out


def training_loss(cfg, out, gt_rgb, step, eikonal_weight=0.1):
    """Trainer loss assembly (train/trainer.py:127-137 + network/loss.py:8-122): sum of means of every 'loss*' entry."""
    loss = torch.mean(rgb_loss(cfg, out['ray_rgb'], gt_rgb)) + torch.mean(out['gradient_error'] * eikonal_weight)
    if 'loss_occ' in out:
        loss = loss + torch.mean(out['loss_occ'])
    if step < 1000 and 'sdf_vals' in out:
        norm = torch.norm(out['sdf_pts'], dim=-1)
        sdf = out['sdf_vals']
        w = (math.cos(step / 1000 * math.pi) + 1) / 2
        sm = norm < 0.1
        if int(sm.sum()) > 0:
            sl = torch.mean(torch.clamp(sdf[sm] - (norm[sm] - 0.1), min=0.0))
            loss = loss + sl / ((sl > 1e-5).float() + 1e-3) * w
        lm = norm > 1.05
        if int(lm.sum()) > 0:
            ll = torch.clamp((norm[lm] - 1.05) - sdf[lm], min=0.0)
            loss = loss + torch.sum(ll) / (torch.sum(ll > 1e-5) + 1e-3) * w
    return loss
