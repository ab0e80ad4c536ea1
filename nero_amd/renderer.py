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

    def get_human_coordinate_poses(self, poses):
        """per-image "human" frame [R|t]: z = horizontal viewing direction, y = -world z, origin at the camera centre projected
        to z=0 unless fixed_camera (network/renderer.py:240-256)."""
        pn = poses.shape[0]
        cam_cen = (-poses[:, :, :3].permute(0, 2, 1) @ poses[:, :, 3:])[..., 0].clone()
        if not self.cfg['fixed_camera']:
            cam_cen[..., 2] = 0
        Y = torch.zeros([pn, 3], device=poses.device, dtype=poses.dtype)
        Y[:, 2] = -1.0
        Z = torch.clone(poses[:, 2, :3])
        Z[:, 2] = 0
        Z = torch.nn.functional.normalize(Z, dim=-1)
        X = torch.cross(Y, Z, dim=-1)
        Rm = torch.stack([X, Y, Z], 1)
        t = -Rm @ cam_cen[:, :, None]
        return torch.cat([Rm, t], -1)

    def get_anneal_val(self, step):
        e = self.cfg['anneal_end']
        return 1.0 if e < 0 else float(np.min([1.0, step / e]))

    @staticmethod
    def near_far_from_sphere(rays_o, rays_d):
        a = torch.sum(rays_d ** 2, dim=-1, keepdim=True)
        b = 2.0 * torch.sum(rays_o * rays_d, dim=-1, keepdim=True)
        mid = 0.5 * (-b) / a
        return torch.clamp(mid - 1.0, min=1e-3), mid + 1.0

    # ------------------------------------------------------------------------------------------------------------
    def _kernels(self):
        """effective weights (autograd tensors) + packed HIP chains for the current parameter values"""
        from .shape_step import ShapeKernels, flatten_effective, unflatten_effective
        names, eff = flatten_effective(self)
        K = ShapeKernels(unflatten_effective(names, [t.detach() for t in eff]), self.color_network.cfg, eff[0].device).pack()
        return names, eff, K

    def sample_ray(self, rays_o, rays_d, near, far, perturb, rand1=None, rand_bg=None, K=None, trace=None):
        """network/renderer.py:403-443 on the HIP sampler.  rand1 [R,1] / rand_bg [R,n_bg]: optional explicit uniform draws
        (default: torch.rand on the device when perturb > 0, like the reference)."""
        from .shape_step import sample_ray
        if K is None:
            _, _, K = self._kernels()
        R = rays_o.shape[0]
        if perturb > 0:
            if rand1 is None:
                rand1 = torch.rand([R, 1], device=rays_o.device)
            if rand_bg is None:
                rand_bg = torch.rand([R, self.cfg['n_bg_samples']], device=rays_o.device)
        else:
            rand1 = rand_bg = None
        with torch.no_grad():
            return sample_ray(K, self.cfg, rays_o.contiguous(), rays_d.contiguous(), near.contiguous(), far.contiguous(),
                              self.deviation_network.variance.detach(), rand1, rand_bg, trace)

    def render(self, rays_o, rays_d, near, far, human_poses, perturb_overwrite=-1, cos_anneal_ratio=0.0, is_train=True,
               step=None, rand1=None, rand_bg=None, z_vals=None, occ_keys=None):
        """same contract as the reference (network/renderer.py:445-463); extra keyword-only style arguments rand1 / rand_bg /
        z_vals allow tests to inject the random draws or teacher-force the sample positions."""
        perturb = self.cfg['perturb']
        if perturb_overwrite >= 0:
            perturb = perturb_overwrite
        names, eff, K = self._kernels()
        if z_vals is None:
            z_vals = self.sample_ray(rays_o, rays_d, near, far, perturb, rand1, rand_bg, K)
        return self.render_core(rays_o, rays_d, z_vals, human_poses, cos_anneal_ratio=cos_anneal_ratio, step=step,
                                is_train=is_train, _kern=(names, eff, K), occ_keys=occ_keys)

    def render_core(self, rays_o, rays_d, z_vals, human_poses, cos_anneal_ratio=0.0, step=None, is_train=True, _kern=None,
                    occ_keys=None):
        from .shape_step import RenderCore, SDFValue, occ_loss
        if not is_train:
            raise NotImplementedError('validation extras (compute_validation_info) are not on the HIP path yet')
        names, eff, Kpre = _kern if _kern is not None else self._kernels()
        c = self.cfg
        meta = {'names': names, 'shapes': [tuple(t.shape) for t in eff], 'shader_cfg': self.color_network.cfg,
                'anneal': float(cos_anneal_ratio), 'exp_max': float(self.color_network.cfg['light_exp_max']),
                'freeze_inv_s': c['freeze_inv_s_step'] is not None and step < c['freeze_inv_s_step']}
        var = self.deviation_network.variance
        poses = None
        if self.color_network.cfg['human_light']:
            if human_poses is None:
                raise ValueError('shader_config.human_light needs human_poses [R,3,4]')
            poses = human_poses.to(torch.float32).contiguous()
        rgb, gerr, occ_prob = RenderCore.apply(meta, rays_o.contiguous(), rays_d.contiguous(), z_vals.contiguous(), var,
                                               self.color_network.FG_LUT, poses, *eff)
        n_in = gerr.shape[0]
        outputs = {'ray_rgb': rgb, 'gradient_error': gerr if n_in > 0 else torch.zeros(1, device=rgb.device)}
        inv_s = torch.exp(var * 10.0).clip(1e-6, 1e6)
        if meta['freeze_inv_s']:
            inv_s = inv_s.detach()
        outputs['std'] = torch.mean(1.0 / inv_s) if n_in > 0 else torch.zeros(1, device=rgb.device)
        S = meta.get('_state')
        if step is not None and step < 1000:                               # inputs of InitSDFRegLoss (renderer.py:591-594)
            pts = S['pts4'][:, :3]
            m = torch.norm(pts, dim=-1) < 1.2
            outputs['sdf_pts'] = pts[m]
            outputs['sdf_vals'] = SDFValue.apply(Kpre, outputs['sdf_pts'], *eff[:18])
        if c['apply_occ_loss']:
            outputs['loss_occ'] = torch.zeros(1, device=rgb.device)
            if n_in > 0 and step is not None and step >= c['occ_loss_step']:
                outputs['loss_occ'], outputs['_occ_count'] = occ_loss(S, occ_prob, c, var.detach(), occ_keys)
        outputs['_occ_prob'] = occ_prob
        outputs['_state'] = meta.get('_state')
        return outputs

    def compute_rgb_loss(self, rgb_pr, rgb_gt):
        kind = self.cfg['rgb_loss']
        if kind == 'l2':
            return torch.sum((rgb_pr - rgb_gt) ** 2, -1)
        if kind == 'l1':
            return torch.sum(torch.abs(rgb_pr - rgb_gt), -1)
        if kind == 'smooth_l1':
            return torch.sum(torch.nn.functional.smooth_l1_loss(rgb_pr, rgb_gt, reduction='none', beta=0.25), -1)
        if kind == 'charbonier':
            return torch.sqrt(torch.sum((rgb_gt - rgb_pr) ** 2, dim=-1) + 0.001)
        raise NotImplementedError


name2renderer = {'shape': NeROShapeRenderer}
