"""CPU oracle for Stage II (material estimation): functional restatement of MCShadingNetwork (network/field.py:694-1087)
and the loss terms NeROMaterialRenderer.train_step assembles (network/renderer.py:815-848).

TEST INFRASTRUCTURE ONLY (same rules as oracle/nero_oracle.py).  Pinned against vectors dumped from the unmodified reference
(oracle/gen_golden.py::run_material_case -> tests/golden/mat_*.npz; tests/test_oracle_golden.py).  The mesh tracer is an
argument: the reference's tracer is an un-vendored third-party CUDA extension, so both the golden run and the tests use the
brute-force oracle in oracle/tracer_oracle.py behind the contract of NeROMaterialRenderer.trace (renderer.py:719-729).

Every random draw is an input: rand_d / rand_s [P,1,1] (azimuth offsets, field.py:781,804), reg_ang [P,1] and reg_eps [P,1]
(material_regularization, field.py:1069,1073).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import nero_oracle as _O
from .nero_oracle import (camera_plane_intersection, ide, ipe, linear_to_srgb, pos_enc, predictor, sphere_exit_dist,
                          _exp_act)

DEFAULT_SHADER_CFG = {
    'diffuse_sample_num': 512, 'specular_sample_num': 256, 'human_lights': True, 'light_exp_max': 5.0,
    'inner_light_exp_max': 5.0, 'outer_light_version': 'direction', 'geometry_type': 'schlick', 'reg_change': True,
    'change_eps': 0.05, 'change_type': 'gaussian', 'reg_lambda1': 0.005, 'reg_min_max': True, 'random_azimuth': True,
    'is_real': False,
}


def fibonacci_az_el(n):
    """sample_sphere(n, 0) scaled to [0,1]  (utils/base_utils.py:800-813; network/field.py:741-749)  -> az[n], el[n]"""
    num = int(n // 0.5)
    phi = (np.sqrt(5) - 1.0) / 2.0
    idx = np.arange(num - n, num)
    z = 2.0 * idx / num - 1.0
    az = (2 * np.pi * idx * phi) % (2 * np.pi)
    el = np.arcsin(z)
    return (az * 0.5 / np.pi).astype(np.float32), (1 - 2 * el / np.pi).astype(np.float32)


def orthogonal_direction(d):
    """network/field.py:756-766"""
    x, y, z = d[..., 0:1], d[..., 1:2], d[..., 2:3]
    zero = torch.zeros_like(x)
    o0 = torch.cat([y, -x, zero], -1)
    o1 = torch.cat([-z, zero, x], -1)
    m = (torch.norm(o0, dim=-1) > torch.norm(o1, dim=-1)).unsqueeze(-1)
    return F.normalize(torch.where(m, o0, o1), dim=-1)


def sample_diffuse_directions(normals, n, rand):
    """network/field.py:768-787.  rand: [P,1,1] in [0,1) or None (inference)."""
    z = normals
    x = orthogonal_direction(normals)
    y = torch.cross(z, x, dim=-1)
    az_t, el_t = fibonacci_az_el(n)
    az = torch.from_numpy(az_t).to(normals)[None, :, None] * np.pi * 2
    el = torch.from_numpy(el_t).to(normals)[None, :, None]
    el_sqrt = torch.sqrt(el + 1e-7)
    if rand is not None:
        az = (az + rand * np.pi * 2) % (2 * np.pi)
    cz = torch.sqrt(1 - el + 1e-7)
    cx, cy = el_sqrt * torch.cos(az), el_sqrt * torch.sin(az)
    return cx * x.unsqueeze(1) + cy * y.unsqueeze(1) + cz * z.unsqueeze(1)


def sample_specular_directions(refl, rough, n, rand):
    """network/field.py:789-810 (GGX importance sampling; the predicted roughness is used as alpha directly)."""
    z = refl
    x = orthogonal_direction(refl)
    y = torch.cross(z, x, dim=-1)
    az_t, el_t = fibonacci_az_el(n)
    phi = torch.from_numpy(az_t).to(refl)[None, :, None] * np.pi * 2
    el = torch.from_numpy(el_t).to(refl)[None, :, None]
    a = rough.unsqueeze(1)
    cos_t = torch.sqrt((1.0 - el + 1e-6) / (1.0 + (a ** 2 - 1.0) * el + 1e-6) + 1e-6)
    sin_t = torch.sqrt(1 - cos_t ** 2 + 1e-6)
    if rand is not None:
        phi = (phi + rand * np.pi * 2) % (2 * np.pi)
    return torch.cos(phi) * sin_t * x.unsqueeze(1) + torch.sin(phi) * sin_t * y.unsqueeze(1) + cos_t * z.unsqueeze(1)


def sat_dot(a, b):
    return torch.clamp(torch.sum(a * b, -1, keepdim=True), 0.0, 1.0)


def distribution_ggx(noh, a):
    a2 = a ** 2
    denom = noh ** 2 * (a2 - 1.0) + 1.0
    return a2 / (np.pi * denom ** 2 + 1e-4)


def geometry_schlick(nov, nol, a):
    k = a / 2
    return (nov / (nov * (1 - k) + k + 1e-5)) * (nol / (nol * (1 - k) + k + 1e-5))


def geometry_ggx_smith_correlated(nov, nol, a):
    def lam(a2, c):
        c2 = c ** 2
        return 0.5 * torch.sqrt(1 + a2 * (1 - c2) / (c2 + 1e-7)) - 0.5
    return 1.0 / (1.0 + lam(a ** 2, nov) + lam(a ** 2, nol))


def feats_network(P, x, prefix='shader_network.feats_network'):
    """MaterialFeatsNetwork (network/field.py:660-689)"""
    call = _O.next_call(prefix)                      # (forced ReLU gates, a test hook: nero_oracle.forced_relu_gates)
    e = pos_enc(x, 8)
    h = e
    for i, l in enumerate((0, 2, 4, 6)):
        h = _O._relu(F.linear(h, P[f'{prefix}.module0.{l}.weight'], P[f'{prefix}.module0.{l}.bias']), f'{prefix}@{call}/m0_{i}')
    h = torch.cat([h, e], -1)
    for i, l in enumerate((0, 2, 4, 6)):
        h = F.linear(h, P[f'{prefix}.module1.{l}.weight'], P[f'{prefix}.module1.{l}.bias'])
        if i < 3:
            h = _O._relu(h, f'{prefix}@{call}/m1_{i}')
    return h


def predict_materials(P, pts, prefix='shader_network'):
    """network/field.py:915-922"""
    fx = torch.cat([feats_network(P, pts, prefix + '.feats_network'), pts], -1)
    metallic = predictor(P, prefix + '.metallic_predictor', fx, torch.sigmoid)
    rough = predictor(P, prefix + '.roughness_predictor', fx, torch.sigmoid) * (1.0 - 0.04 ** 2) + 0.04 ** 2
    albedo = predictor(P, prefix + '.albedo_predictor', fx, torch.sigmoid)
    return metallic, rough, albedo


def outer_lights(P, cfg, pts, dirs, prefix='shader_network'):
    """predict_outer_lights (network/field.py:836-854)"""
    enc = ide(dirs, 0.0)
    if cfg['outer_light_version'] == 'direction':
        return predictor(P, prefix + '.outer_light', enc, _exp_act(cfg['light_exp_max']))
    if cfg['outer_light_version'] == 'sphere_direction':
        far = (torch.norm(pts, dim=-1, keepdim=True) > 0.999)
        p = torch.where(far, pts * 0.999, pts)
        sph = p + dirs * sphere_exit_dist(p, dirs)
        return predictor(P, prefix + '.outer_light', torch.cat([enc, ide(sph, 0.0)], -1), _exp_act(cfg['light_exp_max']))
    raise NotImplementedError


def human_light_mc(P, pts, dirs, poses, prefix='shader_network'):
    """get_human_light (network/field.py:820-834): zero-variance IPE of the plane hit"""
    inter, dists, hits = camera_plane_intersection(pts, dirs, poses)
    mean = inter[..., :2] * 0.3
    hits = (hits & (torch.norm(mean, dim=-1) < 1.5) & (dists > 0)).float().unsqueeze(-1)
    mean = mean * hits
    hl = predictor(P, prefix + '.human_light', ipe(mean, torch.zeros_like(mean), 0, 6), _exp_act(0.0)) * hits
    return hl[..., :3], torch.clamp(hl[..., 3:], 0.0, 1.0)


def inner_lights(P, cfg, pts, view, normals, prefix='shader_network'):
    """get_inner_lights (network/field.py:812-818)"""
    n = F.normalize(normals, dim=-1)
    v = F.normalize(view, dim=-1)
    refl = torch.sum(v * n, -1, keepdim=True) * n * 2 - v
    return predictor(P, prefix + '.inner_light', torch.cat([pos_enc(pts, 8), ide(refl, 0.0)], -1), _exp_act(cfg['inner_light_exp_max']))


def get_lights(P, cfg, trace_fn, pts, dirs, poses):
    """network/field.py:856-880.  pts/dirs [P,D,3], poses [P,D,3,4] or None.  trace_fn(o[N,3], d[N,3]) ->
    (inters [N,3], normals [N,3] (already flipped + normalised), depth [N,1], hit [N] bool)"""
    shape = pts.shape[:-1]
    pf, df = pts.reshape(-1, 3), dirs.reshape(-1, 3)
    inters, normals, depth, hit = trace_fn((pf + df * 1e-5).detach(), df.detach())
    miss = ~hit
    lights = torch.zeros(pf.shape[0], 3, dtype=pts.dtype, device=pts.device)
    hl_out = torch.zeros(1, 3, dtype=pts.dtype, device=pts.device)
    if int(miss.sum()) > 0:
        mi = torch.nonzero(miss)[:, 0]
        outer = outer_lights(P, cfg, pf[mi], df[mi])
        if cfg['human_lights']:
            hl, hw = human_light_mc(P, pf[mi], df[mi], poses.reshape(-1, 3, 4)[mi])
        else:
            hl, hw = torch.zeros_like(outer), torch.zeros(outer.shape[0], 1, dtype=outer.dtype, device=outer.device)
        lights = lights.index_put((mi,), outer * (1 - hw) + hl * hw)
        hl_out = hl * hw
    if int(hit.sum()) > 0:
        hi = torch.nonzero(hit)[:, 0]
        lights = lights.index_put((hi,), inner_lights(P, cfg, inters[hi], -df[hi], normals[hi]))
    lights = lights * (depth > 1e-5).float()
    return lights.reshape(*shape, 3), hl_out, hit.reshape(*shape)


def mc_shade(P, cfg, trace_fn, pts, view_dirs, normals, poses, rand_d=None, rand_s=None):
    """MCShadingNetwork.forward + shade_mixed (network/field.py:950-1018).  -> rgb [P,3], outputs dict"""
    cfg = {**DEFAULT_SHADER_CFG, **cfg}
    v, n = F.normalize(view_dirs, dim=-1), F.normalize(normals, dim=-1)
    refl = torch.sum(v * n, -1, keepdim=True) * n * 2 - v
    metallic, rough, albedo = predict_materials(P, pts)
    F0 = 0.04 * (1 - metallic) + metallic * albedo
    dn, sn = cfg['diffuse_sample_num'], cfg['specular_sample_num']
    d_dirs = sample_diffuse_directions(n, dn, rand_d)
    s_dirs = sample_specular_directions(refl, rough, sn, rand_s)
    nol_d = sat_dot(d_dirs, n.unsqueeze(1))
    p_d = nol_d / np.pi * (dn / (sn + dn))
    H_s = F.normalize(v.unsqueeze(1) + s_dirs, dim=-1)
    noh_s, voh_s = sat_dot(n.unsqueeze(1), H_s), sat_dot(v.unsqueeze(1), H_s)
    p_s = distribution_ggx(noh_s, rough.unsqueeze(1)) * noh_s / (4 * voh_s + 1e-5) * (sn / (sn + dn))
    dirs = torch.cat([d_dirs, s_dirs], 1)
    prob = torch.cat([p_d, p_s], 1)
    D = dn + sn
    H = F.normalize(v.unsqueeze(1) + dirs, dim=-1)
    hov = torch.clamp(torch.sum(H * v.unsqueeze(1), -1, keepdim=True), 0.0, 1.0)
    fresnel = F0.unsqueeze(1) + (1.0 - F0.unsqueeze(1)) * torch.clamp(1.0 - hov, 0.0, 1.0) ** 5.0
    nov = sat_dot(n, v).unsqueeze(1)
    nol = sat_dot(n.unsqueeze(1), dirs)
    if cfg['geometry_type'] == 'schlick':
        geom = geometry_schlick(nov, nol, rough.unsqueeze(1))
    elif cfg['geometry_type'] == 'ggx_smith':
        geom = geometry_ggx_smith_correlated(nov, nol, rough.unsqueeze(1))
    else:
        raise NotImplementedError
    dist = distribution_ggx(sat_dot(n.unsqueeze(1), H), rough.unsqueeze(1))
    poses_d = poses.unsqueeze(1).repeat(1, D, 1, 1) if poses is not None else None
    lights, hl, hit = get_lights(P, cfg, trace_fn, pts.unsqueeze(1).repeat(1, D, 1), dirs, poses_d)
    spec_w = dist * geom / (4 * nov * prob + 1e-5)
    spec_l = lights * spec_w
    spec_c = torch.mean(fresnel * spec_l, 1)
    kd = 1 - metallic.unsqueeze(1)
    diff_l = lights[:, :dn]
    diff_c = torch.mean(albedo.unsqueeze(1) * kd * diff_l, 1)
    colors = linear_to_srgb(diff_c + spec_c)
    out = {
        'albedo': albedo, 'roughness': rough, 'metallic': metallic, 'human_lights': hl.reshape(-1, 3),
        'diffuse_light': torch.clamp(linear_to_srgb(torch.mean(diff_l, dim=1)), 0, 1),
        'specular_light': torch.clamp(linear_to_srgb(torch.mean(spec_l, dim=1)), 0, 1),
        'diffuse_color': torch.clamp(linear_to_srgb(diff_c), 0, 1), 'specular_color': torch.clamp(linear_to_srgb(spec_c), 0, 1),
        # reference quirk (field.py:1007-1011): `specular_colors` has already been re-assigned to its clamped sRGB value here
        'approximate_light': torch.clamp(linear_to_srgb(torch.mean(kd * diff_l, dim=1) + torch.clamp(linear_to_srgb(spec_c), 0, 1)), 0, 1),
        'hit_fraction': hit.float().mean(),
    }
    return colors, out


def material_regularization(P, cfg, pts, normals, metallic, rough, albedo, step, reg_ang, reg_eps):
    """network/field.py:1061-1087"""
    cfg = {**DEFAULT_SHADER_CFG, **cfg}
    reg = 0
    if cfg['reg_change']:
        n = F.normalize(normals, dim=-1)
        x = orthogonal_direction(n)
        y = torch.cross(n, x, dim=-1)
        ang = reg_ang * np.pi * 2
        if cfg['change_type'] == 'constant':
            change = (torch.cos(ang) * x + torch.sin(ang) * y) * cfg['change_eps']
        elif cfg['change_type'] == 'gaussian':
            change = (torch.cos(ang) * x + torch.sin(ang) * y) * reg_eps
        else:
            raise NotImplementedError
        m0, r0, a0 = predict_materials(P, pts + change)
        reg = reg + torch.mean((_O._abs(m0 - metallic, 'abs/reg_metallic') + _O._abs(r0 - rough, 'abs/reg_roughness')
                                + _O._abs(a0 - albedo, 'abs/reg_albedo')) * cfg['reg_lambda1'], dim=1)
    if cfg['reg_min_max'] and step is not None and step < 2000:
        reg = reg + torch.sum(torch.clamp(rough - 0.98 ** 2, min=0))
        reg = reg + torch.sum(torch.clamp(0.02 ** 2 - rough, min=0))
        reg = reg + torch.sum(torch.clamp(metallic - 0.98, min=0))
        reg = reg + torch.sum(torch.clamp(0.02 - metallic, min=0))
    return reg


def material_train_outputs(P, rcfg, trace_fn, pts, view_dirs, normals, poses, rgb_gt, step, rand_d, rand_s, reg_ang, reg_eps):
    """NeROMaterialRenderer.train_step (network/renderer.py:829-848) -> outputs incl. loss_rgb / loss_mat_reg / loss_diffuse_light"""
    scfg = {**DEFAULT_SHADER_CFG, **rcfg.get('shader_cfg', {})}
    rgb, out = mc_shade(P, scfg, trace_fn, pts, view_dirs, normals, poses, rand_d, rand_s)
    out['rgb_pr'] = rgb
    out['loss_rgb'] = torch.sqrt(torch.sum((rgb_gt - rgb) ** 2, dim=-1) + 1e-3)
    if rcfg.get('reg_mat', True):
        out['loss_mat_reg'] = material_regularization(P, scfg, pts, normals, out['metallic'], out['roughness'], out['albedo'], step,
                                                      reg_ang, reg_eps)
    if rcfg.get('reg_diffuse_light', True):
        dl = out['diffuse_light']
        out['loss_diffuse_light'] = torch.sum(_O._abs(dl - torch.mean(dl, dim=-1, keepdim=True), 'abs/diffuse_light'), dim=-1) * rcfg.get('reg_diffuse_light_lambda', 0.1)
    return out


def material_training_loss(out):
    """train/trainer.py:134-137 over network/loss.py NeRFRenderLoss + MaterialRegLoss"""
    loss = torch.mean(out['loss_rgb'])
    for k in ('loss_mat_reg', 'loss_diffuse_light'):
        if k in out:
            loss = loss + torch.mean(out[k])
    return loss
