"""Parameter containers for the Stage-I fields.  They hold the trainable tensors under exactly the names / shapes the
reference checkpoints use (weight-norm triples `*.weight_g / *.weight_v / *.bias`, `deviation_network.variance`,
buffer `color_network.FG_LUT`; SURVEY.md §5 "Checkpoint / resume"), so `load_state_dict` of a reference `model.pth` works
unchanged, and they are constructed with the same torch initialisers in the same order as the reference constructors
(network/field.py:60-128 SDFNetwork, :184-201 SingleVarianceNetwork, :205-256 NeRFNetwork, :310-346 make_predictor,
:486-533 AppShadingNetwork; network/renderer.py:113-131), so a given `torch.manual_seed` yields the same initial weights.

These modules never run a PyTorch forward: the arithmetic lives in the HIP library (nero_amd/csrc).  What they provide is
`effective()` -- the per-step reparametrised weights W = g * v / ||v||_row as autograd tensors -- which `nero_amd.ops`
packs into the kernels' MFMA operand layout; the weight-norm backward stays in PyTorch.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import brdf_lut


def _wn_linear(d_in, d_out):
    return nn.utils.weight_norm(nn.Linear(d_in, d_out))


def _eff(lin):
    """effective (weight, bias) of a Linear, weight-normed or plain."""
    if hasattr(lin, 'weight_g'):
        return torch._weight_norm(lin.weight_v, lin.weight_g, 0), lin.bias
    return lin.weight, lin.bias


class SDFNetwork(nn.Module):
    """PE-6 -> 9 weight-normed layers (skip into layer 4), softplus(beta=100); geometric (sphere) initialisation."""
    N_FREQ = 6
    D_PE = 3 + 3 * 2 * 6        # 39

    def __init__(self, d_out=257, d_hidden=256, n_layers=8, bias=0.5, geometric_init=True):
        super().__init__()
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
    def __init__(self, init_val):
        super().__init__()
        self.register_parameter('variance', nn.Parameter(torch.tensor(init_val)))

    def inv_s(self):
        return torch.exp(self.variance * 10.0)


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
    }

    def __init__(self, cfg):
        super().__init__()
        self.cfg = {**self.default_cfg, **cfg}
        if self.cfg['sphere_direction']:
            raise NotImplementedError('shader_config.sphere_direction is not supported by the HIP path yet')
        self.metallic_predictor = Predictor(256 + 3, 1)
        if self.cfg['metallic_init'] != 0:
            nn.init.constant_(self.metallic_predictor[-2].bias, self.cfg['metallic_init'])
        self.roughness_predictor = Predictor(256 + 3, 1)
        if self.cfg['roughness_init'] != 0:
            nn.init.constant_(self.roughness_predictor[-2].bias, self.cfg['roughness_init'])
        self.albedo_predictor = Predictor(256 + 3, 3)
        self.register_buffer('FG_LUT', torch.from_numpy(brdf_lut.fg_lut()).reshape(1, 256, 256, 2))
        pos_dim = 3 + 3 * 2 * self.cfg['light_pos_freq']
        self.outer_light = Predictor(72, 3)
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
                     geometric_init=cfg['geometry_init'])
    dev = SingleVarianceNetwork(cfg['inv_s_init'])
    nerf = NeRFNetwork()
    nn.init.constant_(nerf.rgb_linear.bias, math.log(0.5))
    color = AppShadingNetwork(cfg['shader_config'])
    return sdf, dev, nerf, color
