"""Parameter containers for the Stage-I fields.  They hold the trainable tensors under exactly the names / shapes the
reference checkpoints use (weight-norm triples `*.weight_g / *.weight_v / *.bias`, `deviation_network.variance`,
buffer `color_network.FG_LUT`; SURVEY.md §5 "Checkpoint / resume"), so `load_state_dict` of a reference `model.pth` works
unchanged, and they are constructed with the same torch initialisers in the same order as the reference constructors
(network/field.py:60-128 SDFNetwork, :184-201 SingleVarianceNetwork, :205-256 NeRFNetwork, :310-346 make_predictor,
:486-533 AppShadingNetwork; network/renderer.py:113-131), so a given `torch.manual_seed` yields the same initial weights.

These modules never run a PyTorch forward: the arithmetic lives in the HIP library (nero_amd/csrc).  What they provide is
`effective()` -- the per-step reparametrised weights W = g * v / ||v||_row as autograd tensors -- which `nero_amd.ops`
packs into the kernels' MFMA operand layout.  The weight-norm forward / backward of ALL Linears is one batched autograd node on CUDA
(nero_amd/wn_fused.py, round 6); on CPU modules it stays torch._weight_norm per Linear.
"""
import math
import threading

import numpy as np
import torch
import torch.nn as nn

from . import brdf_lut


def _wn_linear(d_in, d_out):
    return nn.utils.weight_norm(nn.Linear(d_in, d_out))


_WN = threading.local()


def _eff(lin):
    """effective (weight, bias) of a Linear, weight-normed or plain.  Inside batched_weight_norm() every weight-normed Linear of the model
    goes through ONE autograd node (nero_amd/wn_fused.py) instead of one torch._weight_norm each."""
    if hasattr(lin, 'weight_g'):
        st = getattr(_WN, 'st', None)
        if st is not None:
            if st['w'] is None:                        # first pass: collect the modules in call order
                st['lins'].append(lin)
                return lin.weight_v, lin.bias          # (placeholder of the right shape; this pass's results are discarded)
            return st['w'][id(lin)], lin.bias
        return torch._weight_norm(lin.weight_v, lin.weight_g, 0), lin.bias
    return lin.weight, lin.bias


def batched_weight_norm(collect_fn, owner=None):
    """run `collect_fn()` (something that calls .effective() on the model's networks) with every weight-normed Linear's effective weight
    coming from ONE batched autograd node -> collect_fn's result.  Falls back to the per-Linear torch path when the batch node is not
    usable (CPU modules) or NERO_WN_BATCH=0.  owner: the module the list of weight-normed Linears is cached on (the module tree does not
    change between steps; without the cache every step walks the networks twice)."""
    import os
    from . import wn_fused
    if os.environ.get('NERO_WN_BATCH', '1') == '0' or getattr(_WN, 'st', None) is not None:
        return collect_fn()
    uniq = getattr(owner, '_wn_lins', None) if owner is not None else None
    _WN.st = {'lins': [], 'w': None}
    try:
        if uniq is None:
            collect_fn()
            uniq = list({id(l): l for l in _WN.st['lins']}.values())
            if owner is not None:
                object.__setattr__(owner, '_wn_lins', uniq)            # (not a submodule / parameter: keep nn.Module's registries out of it)
        if not wn_fused.usable(uniq):
            _WN.st = None
            return collect_fn()
        _WN.st['w'] = dict(zip((id(l) for l in uniq), wn_fused.weight_norm_batch(uniq)))
        return collect_fn()
    finally:
        _WN.st = None


class SDFNetwork(nn.Module):
    """PE-f -> n_layers + 1 weight-normed layers (the input re-injected in front of layer n_layers // 2), softplus(beta=100);
    geometric (sphere) initialisation.  Every shipped YAML: f = 6 (39 input columns), n_layers = 8."""

    def __init__(self, d_out=257, d_hidden=256, n_layers=8, bias=0.5, geometric_init=True, n_freq=6):
        super().__init__()
        self.N_FREQ = n_freq
        self.D_PE = 3 + 3 * 2 * n_freq          # 39 at the YAMLs' sdf_freq = 6
        dims = [self.D_PE] + [d_hidden] * n_layers + [d_out]
        self.n_lin = len(dims) - 1
        self.skip = n_layers // 2
        for l in range(self.n_lin):
            out_dim = dims[l + 1] - dims[0] if l + 1 == self.skip else dims[l + 1]
            lin = nn.Linear(dims[l], out_dim)
            if geometric_init:
                if l == self.n_lin - 1:
                    nn.init.normal_(lin.weight, mean=np.sqrt(np.pi) / np.sqrt(dims[l]), std=0.0001)
                    nn.init.constant_(lin.bias, -bias)
                elif l == 0:
                    nn.init.constant_(lin.bias, 0.0)
                    nn.init.constant_(lin.weight[:, 3:], 0.0)
                    nn.init.normal_(lin.weight[:, :3], 0.0, np.sqrt(2) / np.sqrt(out_dim))
                elif l == self.skip:
                    nn.init.constant_(lin.bias, 0.0)
                    nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
                    nn.init.constant_(lin.weight[:, -(dims[0] - 3):], 0.0)
                else:
                    nn.init.constant_(lin.bias, 0.0)
                    nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
            setattr(self, f'lin{l}', nn.utils.weight_norm(lin))

    def effective(self):
        return [_eff(getattr(self, f'lin{l}')) for l in range(self.n_lin)]


class SingleVarianceNetwork(nn.Module):
    """network/field.py:184-198.  activation: 'exp' (every shipped YAML) | 'linear' | 'square'."""

    def __init__(self, init_val, activation='exp'):
        super().__init__()
        if activation not in ('exp', 'linear', 'square'):
            raise NotImplementedError(activation)                # (the reference raises NotImplementedError in forward, field.py:197)
        self.act = activation
        self.register_parameter('variance', nn.Parameter(torch.tensor(init_val)))

    def inv_s(self):
        v10 = self.variance * 10.0
        return torch.exp(v10) if self.act == 'exp' else (v10 if self.act == 'linear' else v10 ** 2)

    def kernel_variance(self):
        """the scalar the HIP kernels are handed as `variance`: they all form inv_s = exp(10 * variance) (sampler.hip, shade.hip), so
        for 'linear' / 'square' they get v' = log(inv_s) / 10 -- a differentiable torch expression of the parameter, through which
        autograd chains d inv_s / d variance (10, resp. 200 v) onto the kernels' d inv_s.  inv_s <= 1e-6 (a non-positive 'linear'
        variance; the reference then divides by a non-positive inv_s) is clamped to 1e-6, the lower clip of render_core
        (network/renderer.py:493)."""
        if self.act == 'exp':
            return self.variance
        return torch.log(self.inv_s().clamp(min=1e-6)) / 10.0


class NeRFNetwork(nn.Module):
    """NeRF++ background field: plain (not weight-normed) Linear layers."""
    D_PE, D_PE_VIEW, W = 4 + 4 * 2 * 10, 3 + 3 * 2 * 4, 256     # 84, 27

    def __init__(self):
        super().__init__()
        W, e = self.W, self.D_PE
        self.pts_linears = nn.ModuleList([nn.Linear(e, W)] +
                                         [nn.Linear(W, W) if i != 4 else nn.Linear(W + e, W) for i in range(7)])
        self.views_linears = nn.ModuleList([nn.Linear(self.D_PE_VIEW + W, W // 2)])
        self.feature_linear = nn.Linear(W, W)
        self.alpha_linear = nn.Linear(W, 1)
        self.rgb_linear = nn.Linear(W // 2, 3)

    def effective(self):
        return {'pts': [_eff(l) for l in self.pts_linears], 'views': _eff(self.views_linears[0]),
                'feature': _eff(self.feature_linear), 'alpha': _eff(self.alpha_linear), 'rgb': _eff(self.rgb_linear)}


class Predictor(nn.Sequential):
    """4 weight-normed Linear layers at Sequential indices 0,2,4,6 (ReLU placeholders at 1,3,5; index 7 is the output
    activation in the reference, parameter-free)."""

    def __init__(self, d_in, d_out):
        super().__init__(_wn_linear(d_in, 256), nn.ReLU(), _wn_linear(256, 256), nn.ReLU(),
                         _wn_linear(256, 256), nn.ReLU(), _wn_linear(256, d_out), nn.Identity())

    def effective(self):
        return [_eff(self[i]) for i in (0, 2, 4, 6)]


class AppShadingNetwork(nn.Module):
    default_cfg = {
        'human_light': False, 'sphere_direction': False, 'light_pos_freq': 8, 'inner_init': -0.95,
        'roughness_init': 0.0, 'metallic_init': 0.0, 'light_exp_max': 0.0,
        'fg_lut_path': None,        # (ours) explicit location of the reference's assets/bsdf_256_256.bin; default: cwd-relative like field.py:510
    }

    def __init__(self, cfg):
        super().__init__()
        self.cfg = {**self.default_cfg, **cfg}
        self.metallic_predictor = Predictor(256 + 3, 1)
        if self.cfg['metallic_init'] != 0:
            nn.init.constant_(self.metallic_predictor[-2].bias, self.cfg['metallic_init'])
        self.roughness_predictor = Predictor(256 + 3, 1)
        if self.cfg['roughness_init'] != 0:
            nn.init.constant_(self.roughness_predictor[-2].bias, self.cfg['roughness_init'])
        self.albedo_predictor = Predictor(256 + 3, 3)
        self.register_buffer('FG_LUT', torch.from_numpy(brdf_lut.fg_lut(self.cfg['fg_lut_path'])).reshape(1, 256, 256, 2))
        pos_dim = 3 + 3 * 2 * self.cfg['light_pos_freq']
        self.outer_light = Predictor(72 * 2 if self.cfg['sphere_direction'] else 72, 3)
        nn.init.constant_(self.outer_light[-2].bias, np.log(0.5))
        self.inner_light = Predictor(pos_dim + 72, 3)
        nn.init.constant_(self.inner_light[-2].bias, np.log(0.5))
        self.inner_weight = Predictor(pos_dim + 39, 1)
        nn.init.constant_(self.inner_weight[-2].bias, self.cfg['inner_init'])
        if self.cfg['human_light']:
            self.human_light_predictor = Predictor(2 * 2 * 6, 4)
            nn.init.constant_(self.human_light_predictor[-2].bias, np.log(0.01))


def build_shape_fields(cfg):
    """-> (sdf_network, deviation_network, outer_nerf, color_network) in the reference's construction order
    (network/renderer.py:117-130)."""
    sdf = SDFNetwork(d_out=cfg['sdf_d_out'], n_layers=cfg['sdf_n_layers'], bias=cfg['sdf_bias'],
                     geometric_init=cfg['geometry_init'], n_freq=cfg['sdf_freq'])
    dev = SingleVarianceNetwork(cfg['inv_s_init'], cfg['std_act'])
    nerf = NeRFNetwork()
    nn.init.constant_(nerf.rgb_linear.bias, math.log(0.5))
    color = AppShadingNetwork(cfg['shader_config'])
    return sdf, dev, nerf, color


# ----------------------------------------------------------------------------------------------------------------------
# Stage II (material) parameter containers: MaterialFeatsNetwork / MCShadingNetwork (network/field.py:660-689, 694-754)
# ----------------------------------------------------------------------------------------------------------------------
def _relu_stack(dims, last_relu):
    mods = []
    for i in range(len(dims) - 1):
        mods.append(_wn_linear(dims[i], dims[i + 1]))
        if i < len(dims) - 2 or last_relu:
            mods.append(nn.ReLU())
    return nn.Sequential(*mods)


class MaterialFeatsNetwork(nn.Module):
    """PE-8(p) -> 4x256 ReLU -> cat[h, pe] -> 3x256 ReLU + Linear 256 (weight-normed; Sequential indices 0,2,4,6)"""

    def __init__(self):
        super().__init__()
        d = 3 + 3 * 2 * 8
        self.module0 = _relu_stack([d, 256, 256, 256, 256], True)
        self.module1 = _relu_stack([d + 256, 256, 256, 256, 256], False)

    def effective(self):
        return [_eff(self.module0[i]) for i in (0, 2, 4, 6)] + [_eff(self.module1[i]) for i in (0, 2, 4, 6)]


def fibonacci_az_el(n):
    """sample_sphere(n, 0) scaled to [0,1] (utils/base_utils.py:800-813; network/field.py:741-749)"""
    num = int(n // 0.5)
    phi = (np.sqrt(5) - 1.0) / 2.0
    idx = np.arange(num - n, num)
    z = 2.0 * idx / num - 1.0
    az = (2 * np.pi * idx * phi) % (2 * np.pi)
    el = np.arcsin(z)
    return az, el


class MCShadingNetwork(nn.Module):
    default_cfg = {
        'diffuse_sample_num': 512, 'specular_sample_num': 256, 'human_lights': True, 'light_exp_max': 5.0,
        'inner_light_exp_max': 5.0, 'outer_light_version': 'direction', 'geometry_type': 'schlick', 'reg_change': True,
        'change_eps': 0.05, 'change_type': 'gaussian', 'reg_lambda1': 0.005, 'reg_min_max': True, 'random_azimuth': True,
        'is_real': False,
    }

    def __init__(self, cfg, ray_trace_fun=None):
        self.cfg = {**self.default_cfg, **cfg}
        super().__init__()
        self.feats_network = MaterialFeatsNetwork()
        self.metallic_predictor = Predictor(256 + 3, 1)
        self.roughness_predictor = Predictor(256 + 3, 1)
        self.albedo_predictor = Predictor(256 + 3, 3)
        if self.cfg['outer_light_version'] == 'direction':
            self.outer_light = Predictor(72, 3)
        elif self.cfg['outer_light_version'] == 'sphere_direction':
            self.outer_light = Predictor(72 * 2, 3)
        else:
            raise NotImplementedError
        nn.init.constant_(self.outer_light[-2].bias, np.log(0.5))
        if self.cfg['human_lights']:
            self.human_light = Predictor(2 * 2 * 6, 4)
            nn.init.constant_(self.human_light[-2].bias, np.log(0.02))
        self.inner_light = Predictor(51 + 72, 3)
        nn.init.constant_(self.inner_light[-2].bias, np.log(0.5))
        az, el = fibonacci_az_el(8192)
        pts = np.stack([np.cos(az) * np.cos(el), np.sin(az) * np.cos(el), np.sin(el)], -1)
        self.register_buffer('light_pts', torch.from_numpy(pts.astype(np.float32)))
        self.ray_trace_fun = ray_trace_fun
