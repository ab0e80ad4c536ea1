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
        if c['std_act'] not in ('exp', 'linear', 'square'):
            raise NotImplementedError(f"std_act {c['std_act']!r}")      # (as the reference: network/field.py:197)
        # network-shape keys (network/renderer.py:73-76,118-124).  Round 5: sdf_n_layers, sdf_freq and shader_config.light_pos_freq are free
        # (the chain descriptors take any depth / input width; non-YAML values run on the Python-sequenced chains, the C step driver stays
        # laid out for the YAML shapes).  sdf_activation is accepted and ignored, exactly as the reference does: SDFNetwork.__init__ takes
        # the argument and never reads it (network/field.py:72, 130-147).  What raises, and why:
        if c['sdf_d_out'] != 257:
            # AppShadingNetwork hard-codes feats_dim = 256 (network/field.py:499): the reference itself fails in its first forward
            raise NotImplementedError('sdf_d_out must be 257: the shading network consumes exactly 256 SDF features (network/field.py:499)')
        if not 2 <= int(c['sdf_n_layers']) <= 9:
            raise NotImplementedError('sdf_n_layers must be in [2, 9]: a chain descriptor holds NERO_MAX_LAYERS = 10 layers (include/nero_hip.h)')
        if not 1 <= int(c['sdf_freq']) <= 6:
            # sdf_freq 0 builds a different network in the reference (no embedding, another initialisation: network/field.py:80-107)
            raise NotImplementedError('sdf_freq must be in [1, 6]: the PE input (3 + 6 f columns) is re-read by the skip layer from a 40-column '
                                      'aux tile and its Jacobian kernels (nero_pe_vjp / nero_pe_jvp) stage rows of <= 40 floats')
        sc = c['shader_config']
        if not 1 <= int(sc.get('light_pos_freq', 8)) <= 10:
            raise NotImplementedError('shader_config.light_pos_freq must be in [1, 10]')
        if c['n_importance'] % c['up_sample_steps'] != 0:
            raise NotImplementedError('n_importance must be a multiple of up_sample_steps')
        self.sdf_network, self.deviation_network, self.outer_nerf, self.color_network = build_shape_fields(c)
        if training:
            self._init_dataset()

    # ---- dataset-backed ray pool, resident in HBM (network/renderer.py:136-187, 319-326) ---------
    def _init_dataset(self, database=None):
        """`database` is any object with the reference's BaseDatabase interface (get_image / get_K / get_pose / get_img_ids,
        dataset/database.py:20-42).  When omitted it is resolved through the reference's own `dataset.database` module, which
        must then be importable (the renderer is meant to be dropped into the reference tree; dataset IO is out of scope here)."""
        if database is None:
            try:
                from dataset.database import get_database_split, parse_database_name
            except ImportError as e:
                raise ImportError('no database object given and the reference `dataset.database` module is not importable') from e
            database = parse_database_name(self.cfg['database_name'])
            train_ids, test_ids = get_database_split(database)
        else:
            ids = list(database.get_img_ids())
            train_ids, test_ids = ids, ids[:1]
        self.database, self.train_ids, self.test_ids = database, np.asarray(train_ids), list(test_ids)

        def info(ids):                                            # build_imgs_info (network/renderer.py:17-26)
            imgs = np.stack([np.asarray(database.get_image(i)) for i in ids], 0).astype(np.float32)
            if imgs.max() > 1.5:
                imgs = imgs / 255.0                               # color_map_forward (utils/base_utils.py)
            Ks = np.stack([database.get_K(i) for i in ids], 0).astype(np.float32)
            poses = np.stack([database.get_pose(i) for i in ids], 0).astype(np.float32)
            return torch.from_numpy(imgs), torch.from_numpy(Ks), torch.from_numpy(poses)
        self.test_imgs_info = dict(zip(('imgs', 'Ks', 'poses'), info(self.test_ids)))     # imgs [n,h,w,3], unshuffled, host
        self.train_num, self.test_num = len(self.train_ids), len(self.test_ids)
        self.set_ray_pool(*info(train_ids))

    def set_ray_pool(self, imgs, Ks, poses, device=None):
        """imgs [imn,h,w,3] in [0,1], Ks [imn,3,3], poses [imn,3,4] (world->camera).  Builds the pool of every training pixel
        (dirs = K^-1 [u+.5, v+.5, 1], network/renderer.py:167-187) ON THE DEVICE and shuffles it there; train_step slices it
        without any host->device traffic."""
        device = device or next(self.parameters()).device
        imn, h, w, _ = imgs.shape
        ys, xs = torch.meshgrid(torch.arange(h, device=device), torch.arange(w, device=device), indexing='ij')
        coords = torch.stack([xs + 0.5, ys + 0.5, torch.ones_like(xs, dtype=torch.float32)], -1).reshape(1, h * w, 3).float()
        dirs = coords @ torch.inverse(Ks.to(device)).permute(0, 2, 1)                      # imn, h*w, 3
        self.train_poses = poses.to(device).float()
        self._train_human_poses = self.get_human_coordinate_poses(self.train_poses)   # per IMAGE [imn,3,4]; never overwritten
        self.train_batch = {'dirs': dirs.reshape(-1, 3).contiguous(), 'rgbs': imgs.to(device).reshape(-1, 3).float().contiguous(),
                            'idxs': torch.arange(imn, device=device).repeat_interleave(h * w)}
        self.tbn = imn * h * w
        if not hasattr(self, 'test_imgs_info'):                  # pool given directly (no database): validate on the first view
            self.test_imgs_info = {'imgs': imgs[:1].cpu().float(), 'Ks': Ks[:1].cpu().float(), 'poses': poses[:1].cpu().float()}
            self.test_ids, self.database = [0], None
        self._shuffle_train_batch()

    def _shuffle_train_batch(self):
        self.train_batch_i = 0
        perm = torch.randperm(self.tbn, device=self.train_batch['dirs'].device)
        self.train_batch = {k: v[perm] for k, v in self.train_batch.items()}

    def _process_ray_batch(self, ray_batch, poses, human_poses_img=None):
        """world-space rays of a pool slice (network/renderer.py:258-272).  `human_poses_img` [imn,3,4]: the per-image human
        frames of `poses` when the caller has them cached; computed here otherwise, like the reference does on every call."""
        idxs = ray_batch['idxs']
        if idxs.dim() == 2:
            idxs = idxs[..., 0]                                   # the reference's pools carry idxs as [rn,1]
        Rm, t = poses[:, :, :3], poses[:, :, 3:]
        # camera centres per IMAGE: the same -R^T t the reference recomputes for all images on every call; cached while `poses` is unchanged
        key = (poses.data_ptr(), poses._version, tuple(poses.shape))
        cached = getattr(self, '_cam_centres', None)
        if cached is None or cached[0] != key:
            cached = self._cam_centres = (key, (Rm.permute(0, 2, 1) @ -t)[:, :, 0].contiguous())
        rays_o = cached[1][idxs]
        rays_d = (Rm[idxs].permute(0, 2, 1) @ ray_batch['dirs'].unsqueeze(-1))[..., 0]
        rays_d = torch.nn.functional.normalize(rays_d, dim=-1)
        near, far = self._near_far(rays_o, rays_d)
        if human_poses_img is None:
            human_poses_img = self.get_human_coordinate_poses(poses)
        return rays_o, rays_d, near, far, human_poses_img[idxs]

    def train_step(self, step):
        rn = self.cfg['train_ray_num']
        s = slice(self.train_batch_i, self.train_batch_i + rn)
        batch = {k: v[s] for k, v in self.train_batch.items()}
        self.train_batch_i += rn
        if self.train_batch_i + rn >= self.tbn:
            self._shuffle_train_batch()
        rays_o, rays_d, near, far, human_poses = self._process_ray_batch(batch, self.train_poses, self._train_human_poses)
        kern, drv = self._train_driver(step)
        outputs = self.render(rays_o, rays_d, near, far, human_poses, -1, self.get_anneal_val(step), is_train=True, step=step, _kern=kern, _driver=drv)
        outputs['loss_rgb'] = self.compute_rgb_loss(outputs['ray_rgb'], batch['rgbs'])
        return outputs

    def _train_driver(self, step):
        """(kern, driver) for the drop-in training path (forward({'step': s}) under the caller's own optimiser, INTEGRATION.md option A): the
        C-level step driver packed with this step's effective weights, so that sampling, render forward and render backward are one C call
        each instead of ~180 ctypes launches sequenced from Python -- at the reference's own batch (512 rays) the Python-sequenced step was
        HOST-bound (8.3 ms of interpreter time around 5.3 ms of GPU work: scripts/r06/dropin_profile.py).  Same kernels, same order, same
        bits as the Python-sequenced step (tests/test_stage1_driver.py).  (None, None) when the driver does not apply: no gradients wanted,
        the first 1000 steps (InitSDFRegLoss extras), non-YAML network shapes, another GEMM engine, a CPU module, NERO_DROPIN_DRIVER=0."""
        import os
        if not torch.is_grad_enabled() or step is None or step < 1000 or os.environ.get('NERO_DROPIN_DRIVER', '1') == '0':
            return None, None
        dev = next(self.parameters()).device
        if dev.type != 'cuda':
            return None, None
        from . import stage1
        from .shape_step import flatten_effective
        if not stage1.supported(self.cfg, self.color_network.cfg):
            return None, None
        drv = getattr(self, '_train_drv', None)
        if drv is None or not drv.matches_current_modes():
            drv = self._train_drv = stage1.Stage1Driver(self.cfg, self.color_network.cfg, dev)
        names, eff = flatten_effective(self)
        drv.pack([t.detach().contiguous() for t in eff])
        return (names, eff, None), drv

    def render_image(self, pose, K, h, w, step=300000, chunk=None, extras=False):
        """nvs / test_step inner loop (network/renderer.py:189-222, 301-307): render one h x w view in chunks of test_ray_num rays,
        perturb 0, no grad.  -> dict of [h*w, C] tensors (ray_rgb; with extras=True also the validation intermediates)"""
        dev = next(self.parameters()).device
        chunk = chunk or self.cfg['test_ray_num']
        K = torch.as_tensor(np.asarray(K, np.float32), device=dev).reshape(1, 3, 3)
        pose = torch.as_tensor(np.asarray(pose, np.float32), device=dev).reshape(1, 3, 4)
        ys, xs = torch.meshgrid(torch.arange(h, device=dev), torch.arange(w, device=dev), indexing='ij')
        coords = torch.stack([xs + 0.5, ys + 0.5, torch.ones_like(xs, dtype=torch.float32)], -1).reshape(h * w, 3).float()
        dirs = coords @ torch.inverse(K)[0].T
        hp_img = self.get_human_coordinate_poses(pose)
        outs = {}
        with torch.no_grad():
            kern = self._kernels()                                # packed ONCE per image (and cached across images), not per chunk
            drv = None
            if not extras:
                # colours only (nvs): the C-level driver issues a chunk's ~200 launches from one call each (a 1024-ray chunk is
                # launch-bound from Python), and the occlusion-loss march of a training render is left out (a schedule step below
                # occ_loss_step: nothing else depends on `step` without gradients)
                drv = self._inference_driver(kern)                    # (any chunk size: its workspace is sized for sampler + forward only)
                # a schedule step in [1000, occ_loss_step): no sdf_pts extras below, no occlusion-loss march above.  A configuration whose
                # occ_loss_step <= 1000 has no such step: it keeps the caller's step and the Python-sequenced path
                hi = self.cfg['occ_loss_step'] - 1 if self.cfg['apply_occ_loss'] else max(step, 1000)
                if hi >= 1000:
                    step = min(max(step, 1000), hi)
                else:
                    drv = None
            for i in range(0, h * w, chunk):
                batch = {'dirs': dirs[i:i + chunk], 'idxs': torch.zeros(min(chunk, h * w - i), dtype=torch.long, device=dev)}
                ro, rd, near, far, hp = self._process_ray_batch(batch, pose, hp_img)
                o = self.render(ro, rd, near, far, hp, 0, 0, is_train=not extras, step=step, _kern=kern, _driver=drv)
                for k, v in o.items():
                    if not k.startswith('_') and torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == ro.shape[0]:
                        outs.setdefault(k, []).append(v)
        return {k: torch.cat(v, 0) for k, v in outs.items()}

    def _inference_driver(self, kern):
        """a nero_amd.stage1.Stage1Driver packed with the weights of `kern` (= self._kernels() under no_grad), re-packed only when the
        cached kernels changed; None when the C driver does not cover the selected GEMM engines or the device"""
        names, eff, _ = kern
        if eff[0].device.type != 'cuda':
            return None
        from . import stage1
        if not stage1.supported(self.cfg, self.color_network.cfg):
            return None
        drv = getattr(self, '_infer_drv', None)
        if drv is None or not drv.matches_current_modes():
            drv = self._infer_drv = stage1.Stage1Driver(self.cfg, self.color_network.cfg, eff[0].device)
            drv.forward_only = True
            self._infer_drv_key = None
        key = getattr(self, '_kern_cache', (None,))[0]
        if key is None or self._infer_drv_key != key:
            drv.pack([t.detach() for t in eff])
            self._infer_drv_key = key
        return drv

    def extract_fields(self, bound_min=(-1., -1., -1.), bound_max=(1., 1., 1.), resolution=512, chunk=2 ** 21, outside_val=1.0):
        """SDF on a resolution^3 grid for marching cubes (extract_fields, network/field.py:1090-1108; used by extract_mesh.py:24-27):
        value-only SDF chain, points outside the unit sphere set to `outside_val`.  -> float32 numpy [res,res,res] (x,y,z order)"""
        dev = next(self.parameters()).device
        with torch.no_grad():
            _, _, K = self._kernels()
        axes = [torch.linspace(bound_min[a], bound_max[a], resolution, device=dev) for a in range(3)]
        u = torch.empty(resolution ** 3, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for i in range(0, resolution ** 3, chunk):
                idx = torch.arange(i, min(i + chunk, resolution ** 3), device=dev)
                ix, iy, iz = idx // (resolution * resolution), (idx // resolution) % resolution, idx % resolution
                pts = torch.stack([axes[0][ix], axes[1][iy], axes[2][iz]], -1).contiguous()
                val = K.sdf.sdf(pts)[:, 0]
                u[i:i + idx.numel()] = torch.where(torch.norm(pts, dim=-1) >= 1.0, torch.full_like(val, outside_val), val)
        return u.reshape(resolution, resolution, resolution).cpu().numpy()

    def nvs(self, pose, K, h, w):
        """network/renderer.py:189-222 -> [h,w,3] numpy image"""
        return self.render_image(pose, K, h, w)['ray_rgb'].reshape(h, w, 3).cpu().numpy()

    @staticmethod
    def _downsample_view(img, K, ratio):
        """imgs_info_downsample (network/renderer.py:46-61) for one [h,w,3] image: Gaussian blur (sigma = 1/(3 ratio), cv2's
        kernel-size rule and BORDER_REFLECT101, utils/base_utils.py:119-125), bilinear resize with half-pixel centres
        (cv2.INTER_LINEAR), K scaled by diag(dw/w, dh/h, 1).  Host-side glue, restated in torch (cv2 is not a dependency)."""
        h, w, _ = img.shape
        dh, dw = int(ratio * h), int(ratio * w)
        sigma = (1 / ratio) / 3
        ks = int(np.ceil(((sigma - 0.8) / 0.3 + 1) * 2 + 1))
        ks = ks + 1 if ks % 2 == 0 else ks
        x = torch.arange(ks, dtype=torch.float32) - (ks - 1) / 2
        g = torch.exp(-x ** 2 / (2 * sigma ** 2))
        g = g / g.sum()
        t = img.permute(2, 0, 1).unsqueeze(0).float()
        if ks > 1 and min(h, w) > ks // 2:
            t = torch.nn.functional.pad(t, (ks // 2,) * 4, mode='reflect')
            t = torch.nn.functional.conv2d(t, g.view(1, 1, 1, ks).repeat(3, 1, 1, 1), groups=3)
            t = torch.nn.functional.conv2d(t, g.view(1, 1, ks, 1).repeat(3, 1, 1, 1), groups=3)
        t = torch.nn.functional.interpolate(t, size=(dh, dw), mode='bilinear', align_corners=False)
        Kd = torch.diag(torch.tensor([dw / w, dh / h, 1.0])) @ K
        return t[0].permute(1, 2, 0).contiguous(), Kd

    def test_step(self, index, step):
        """network/renderer.py:274-317: renders view test_ids[index] of the TEST split (down-sampled by downsample_ratio when
        test_downsample_ratio) with all validation outputs, plus loss_rgb / gt_rgb and -- when the database provides
        get_depth -- gt_depth / gt_mask (nearest-neighbour down-sampled like cv2.INTER_NEAREST)."""
        info = self.test_imgs_info
        img, K, pose = info['imgs'][index].float(), info['Ks'][index].float(), info['poses'][index].float()
        gt_depth = gt_mask = None
        db = getattr(self, 'database', None)
        if db is not None and hasattr(db, 'get_depth'):
            dm = db.get_depth(self.test_ids[index])
            if dm is not None:
                gt_depth, gt_mask = torch.from_numpy(np.asarray(dm[0], np.float32)), torch.from_numpy(np.asarray(dm[1]).astype(np.int32))
        if self.cfg['test_downsample_ratio']:
            ratio = self.cfg['downsample_ratio']
            img, K = self._downsample_view(img, K, ratio)
            if gt_depth is not None:
                dh, dw = int(ratio * gt_depth.shape[0]), int(ratio * gt_depth.shape[1])
                near = lambda a: torch.nn.functional.interpolate(a[None, None].float(), size=(dh, dw), mode='nearest')[0, 0]
                gt_depth, gt_mask = near(gt_depth), near(gt_mask).to(torch.int32)
        h, w, _ = img.shape
        out = self.render_image(pose.numpy(), K.numpy(), h, w, step=step, extras=True)
        gt = img.reshape(h * w, 3).to(out['ray_rgb'].device)
        out['loss_rgb'] = self.compute_rgb_loss(out['ray_rgb'], gt)
        out['gt_rgb'] = gt.reshape(h, w, 3)
        out['ray_rgb'] = out['ray_rgb'].reshape(h, w, 3)
        if gt_depth is not None:
            out['gt_depth'], out['gt_mask'] = gt_depth.unsqueeze(-1), gt_mask.unsqueeze(-1)
        # (the reference calls zero_grad() here, network/renderer.py:316; set_to_none=False keeps the .grad tensors -- under
        # nero_amd.parallel.GradBucket they are views of the persistent flat all-reduce buffer and must not be severed)
        self.zero_grad(set_to_none=False)
        return out

    def forward(self, data):
        """Trainer entry point (network/renderer.py:608-627).  No process-global default-tensor-type switch: every tensor is
        created on the parameters' device explicitly."""
        if 'eval' in data:
            return self.test_step(data['index'], step=data['step'])
        outputs = self.train_step(data['step'])
        return {k: v for k, v in outputs.items() if not k.startswith('_')}

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

    def _near_far(self, rays_o, rays_d):
        """near_far_from_sphere for the training / rendering batches: on the device one launch of nero_near_far_sphere -- the ten tensor
        ops below fused, the three products added in ATen's order, bit-identical for the unit directions every caller passes
        (tests/test_step_glue.py) -- instead of ten launches in a step that is launch-dominated at the reference's 512 rays"""
        if rays_o.is_cuda and rays_o.dtype == torch.float32 and rays_o.dim() == 2 and not (rays_o.requires_grad or rays_d.requires_grad):
            from . import stage1
            R = rays_o.shape[0]
            o, d = rays_o.contiguous(), rays_d.contiguous()
            near, far = torch.empty((R, 1), dtype=torch.float32, device=o.device), torch.empty((R, 1), dtype=torch.float32, device=o.device)
            stage1.L.check(stage1._lib.nero_near_far_sphere(stage1._p(o), stage1._p(d), R, stage1._p(near), stage1._p(far), stage1.L.stream_ptr()))
            return near, far
        return self.near_far_from_sphere(rays_o, rays_d)

    @staticmethod
    def near_far_from_sphere(rays_o, rays_d):
        a = torch.sum(rays_d ** 2, dim=-1, keepdim=True)
        b = 2.0 * torch.sum(rays_o * rays_d, dim=-1, keepdim=True)
        mid = 0.5 * (-b) / a
        return torch.clamp(mid - 1.0, min=1e-3), mid + 1.0

    # ------------------------------------------------------------------------------------------------------------
    def _kernels(self):
        """effective weights (autograd tensors) + packed HIP chains for the current parameter values.  Under no_grad (inference:
        render_image / nvs / test_step / extract_fields) the packed operand images are cached and re-used until a parameter changes
        (storage pointer or torch's in-place version counter), so a multi-chunk render packs the ten networks once."""
        from .shape_step import ShapeKernels, flatten_effective, unflatten_effective
        from .chain import GEMM_MODE
        key = None
        if not torch.is_grad_enabled():
            # (_param_epoch: bumped by optimisers that update the parameters with raw kernels, which torch's version counters
            # cannot see -- nero_amd.train.FusedShapeOptimizer)
            key = (tuple((p.data_ptr(), p._version) for p in self.parameters()), tuple(sorted(GEMM_MODE.items())),
                   getattr(self, '_param_epoch', 0))
            cached = getattr(self, '_kern_cache', None)
            if cached is not None and cached[0] == key:
                return cached[1]
        names, eff = flatten_effective(self)
        K = ShapeKernels(unflatten_effective(names, [t.detach() for t in eff]), self.color_network.cfg, eff[0].device).pack()
        if key is not None:
            self._kern_cache = (key, (names, eff, K))
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
                              self.deviation_network.kernel_variance().detach().contiguous(), rand1, rand_bg, trace)

    def render(self, rays_o, rays_d, near, far, human_poses, perturb_overwrite=-1, cos_anneal_ratio=0.0, is_train=True,
               step=None, rand1=None, rand_bg=None, z_vals=None, occ_keys=None, _kern=None, _grad_views=None, _driver=None):
        """same contract as the reference (network/renderer.py:445-463); extra keyword-only style arguments rand1 / rand_bg /
        z_vals allow tests to inject the random draws or teacher-force the sample positions.  _driver: a packed
        nero_amd.stage1.Stage1Driver -- sampling and render_core then run as one C call each (nero_stage1_*)."""
        perturb = self.cfg['perturb']
        if perturb_overwrite >= 0:
            perturb = perturb_overwrite
        names, eff, K = _kern if _kern is not None else self._kernels()
        if z_vals is None:
            if _driver is not None:
                R = rays_o.shape[0]
                if perturb > 0:
                    rand1 = torch.rand([R, 1], device=rays_o.device) if rand1 is None else rand1
                    rand_bg = torch.rand([R, self.cfg['n_bg_samples']], device=rays_o.device) if rand_bg is None else rand_bg
                else:
                    rand1 = rand_bg = None
                with torch.no_grad():
                    z_vals = _driver.sample(rays_o.contiguous(), rays_d.contiguous(), near.contiguous(), far.contiguous(),
                                            self.deviation_network.kernel_variance().detach().contiguous(),
                                            rand1.contiguous() if rand1 is not None else None, rand_bg.contiguous() if rand_bg is not None else None)
            else:
                z_vals = self.sample_ray(rays_o, rays_d, near, far, perturb, rand1, rand_bg, K)
        return self.render_core(rays_o, rays_d, z_vals, human_poses, cos_anneal_ratio=cos_anneal_ratio, step=step,
                                is_train=is_train, _kern=(names, eff, K), occ_keys=occ_keys, _grad_views=_grad_views, _driver=_driver)

    def render_core(self, rays_o, rays_d, z_vals, human_poses, cos_anneal_ratio=0.0, step=None, is_train=True, _kern=None,
                    occ_keys=None, _grad_views=None, _driver=None):
        from .shape_step import RenderCore, SDFValue, occ_loss, validation_info
        names, eff, Kpre = _kern if _kern is not None else self._kernels()
        c = self.cfg
        # the C-level driver covers the training render; the InitSDFRegLoss inputs of the first 1000 steps and the validation extras
        # go through the Python-sequenced chains
        use_driver = _driver is not None and is_train and not (step is not None and step < 1000)
        if not use_driver and Kpre is None:
            from .shape_step import ShapeKernels, unflatten_effective
            Kpre = ShapeKernels(unflatten_effective(names, [t.detach() for t in eff]), self.color_network.cfg, eff[0].device).pack()
        meta = {'names': names, 'shapes': [tuple(t.shape) for t in eff], 'shader_cfg': self.color_network.cfg, 'K': Kpre,
                'anneal': float(cos_anneal_ratio), 'exp_max': float(self.color_network.cfg['light_exp_max']),
                'freeze_inv_s': c['freeze_inv_s_step'] is not None and step < c['freeze_inv_s_step'], 'grad_views': _grad_views}
        var = self.deviation_network.kernel_variance()          # (std_act 'exp': the parameter itself)
        poses = None
        if self.color_network.cfg['human_light']:
            if human_poses is None:
                raise ValueError('shader_config.human_light needs human_poses [R,3,4]')
            poses = human_poses.to(torch.float32).contiguous()
        if use_driver:
            from .stage1 import RenderCoreC
            meta['driver'] = _driver
            rgb, gerr, occ_prob = RenderCoreC.apply(meta, rays_o.contiguous(), rays_d.contiguous(), z_vals.contiguous(), var,
                                                    self.color_network.FG_LUT, poses, *eff)
        else:
            rgb, gerr, occ_prob = RenderCore.apply(meta, rays_o.contiguous(), rays_d.contiguous(), z_vals.contiguous(), var,
                                                   self.color_network.FG_LUT, poses, *eff)
        n_in = gerr.shape[0]
        outputs = {'ray_rgb': rgb, 'gradient_error': gerr if n_in > 0 else torch.zeros(1, device=rgb.device)}
        inv_s = self.deviation_network.inv_s().clip(1e-6, 1e6)
        if meta['freeze_inv_s']:
            inv_s = inv_s.detach()
        outputs['std'] = torch.mean(1.0 / inv_s) if n_in > 0 else torch.zeros(1, device=rgb.device)
        S = meta.get('_state')
        if step is not None and step < 1000:                               # inputs of InitSDFRegLoss (renderer.py:591-594)
            pts = S['pts4'][:, :3]
            m = torch.norm(pts, dim=-1) < 1.2
            outputs['sdf_pts'] = pts[m]
            # (the SDF's effective (W, b) pairs come first in `eff`: 2 * (sdf_n_layers + 1) tensors -- 18 only for the YAML depth 8)
            outputs['sdf_vals'] = SDFValue.apply(Kpre, outputs['sdf_pts'], *eff[:2 * self.sdf_network.n_lin])
        if c['apply_occ_loss']:
            outputs['loss_occ'] = torch.zeros(1, device=rgb.device)
            if n_in > 0 and step is not None and step >= c['occ_loss_step']:
                outputs['loss_occ'], outputs['_occ_count'] = occ_loss(S, occ_prob, c, var.detach(), occ_keys)
        if not is_train:                                                    # renderer.py:603-604
            with torch.no_grad():
                outputs.update(validation_info(Kpre, c, self.color_network.cfg, self.color_network.FG_LUT, var.detach(),
                                               rays_o.contiguous(), rays_d.contiguous(), z_vals, S['weights'], poses))
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


def linear_to_srgb(x):
    """utils/raw_utils.py:4-10 (on [P,3] outputs of the HIP shader; loss glue)"""
    eps = torch.finfo(torch.float32).eps
    return torch.where(x <= 0.0031308, 323 / 25 * x, (211 * torch.clamp(x, min=eps) ** (5 / 12) - 11) / 200)


def material_hinge(shader_cfg, rough, metallic, step):
    """the saturation hinge of the first 2000 steps, a SUM over the batch's points (network/field.py:1079-1084); None when inactive"""
    if not (shader_cfg['reg_min_max'] and step is not None and step < 2000):
        return None
    return (torch.sum(torch.clamp(rough - 0.98 ** 2, min=0)) + torch.sum(torch.clamp(0.02 ** 2 - rough, min=0))
            + torch.sum(torch.clamp(metallic - 0.98, min=0)) + torch.sum(torch.clamp(0.02 - metallic, min=0)))


class NeROMaterialRenderer(nn.Module):
    """Stage II: fixed mesh, Monte-Carlo microfacet shading of surface points (network/renderer.py:649-915).  `mesh` may be given
    as (vertices [nV,3], triangles [nT,3]); otherwise cfg['mesh'] is read with trimesh (not a dependency of the hot path)."""
    default_cfg = {
        'train_ray_num': 512, 'test_ray_num': 1024, 'database_name': 'real/bear/raw_1024', 'rgb_loss': 'charbonier',
        'mesh': 'data/meshes/bear_shape-300000.ply', 'shader_cfg': {}, 'reg_mat': True, 'reg_diffuse_light': True,
        'reg_diffuse_light_lambda': 0.1, 'fixed_camera': False,
    }

    def __init__(self, cfg, is_train=True, mesh=None):
        self.cfg = {**self.default_cfg, **cfg}
        super().__init__()
        from .fields import MCShadingNetwork
        from .raytracing import RayTracer
        if mesh is None:
            try:
                import trimesh
            except ImportError as e:
                raise ImportError('pass mesh=(vertices, triangles) or install trimesh to read cfg["mesh"]') from e
            tm = trimesh.load(self.cfg['mesh'], force='mesh', skip_material=True, process=False)
            mesh = (np.asarray(tm.vertices), np.asarray(tm.faces))
        self.ray_tracer = RayTracer(mesh[0], mesh[1])
        self.cfg['shader_cfg'] = dict(self.cfg['shader_cfg'])
        self.cfg['shader_cfg']['is_real'] = self.cfg['database_name'].startswith('real')
        self.shader_network = MCShadingNetwork(self.cfg['shader_cfg'], lambda o, d: self.trace(o, d))

    def trace(self, rays_o, rays_d):
        """network/renderer.py:719-729: flipped + normalised face normals, hit <=> depth < 10"""
        inters, normals, depth = self.ray_tracer.trace(rays_o, rays_d)
        depth = depth.reshape(*depth.shape, 1)
        normals = torch.nn.functional.normalize(-normals, dim=-1)
        return inters, normals, depth, ~(depth >= 10)[..., 0]

    def _kernels(self):
        """effective weights + packed HIP chains of the shader network.  Under no_grad (test_step / predict_materials_of_vertices: one
        shade() per 1024-ray chunk) the packed operand images are cached until a parameter changes, exactly like
        NeROShapeRenderer._kernels -- an 800 x 800 view used to re-flatten and re-pack all networks 625 times."""
        from .material_step import MaterialKernels, flatten_material_effective, unflatten_material_effective
        from .chain import GEMM_MODE
        key = None
        if not torch.is_grad_enabled():
            key = (tuple((p.data_ptr(), p._version) for p in self.shader_network.parameters()), tuple(sorted(GEMM_MODE.items())),
                   getattr(self, '_param_epoch', 0))
            cached = getattr(self, '_kern_cache', None)
            if cached is not None and cached[0] == key:
                return cached[1]
        names, eff = flatten_material_effective(self.shader_network)
        K = MaterialKernels(unflatten_material_effective(names, [t.detach() for t in eff]), self.shader_network.cfg, eff[0].device).pack()
        if key is not None:
            self._kern_cache = (key, (names, eff, K))
        return names, eff, K

    def predict_materials(self, pts, _kern=None):
        """-> metallic [n,1], roughness [n,1] (affine to [0.04^2, 1]), albedo [n,3]   (network/field.py:915-922)"""
        from .material_step import PredictMaterials
        names, eff, K = _kern if _kern is not None else self._kernels()
        drv = getattr(self, '_driver', None)
        if drv is not None:                           # C-level driver (nero_stage2_predict_fwd / _bwd): opens the step's workspace
            from .stage2 import PredictMaterialsC
            raw = PredictMaterialsC.apply(drv, names[:40], getattr(self, '_grad_views', None), getattr(self, '_n_shade', pts.shape[0]), pts, *eff[:40])
        else:
            raw = PredictMaterials.apply(K, names[:40], getattr(self, '_grad_views', None), pts, *eff[:40])
        rmin = 0.04 ** 2
        return torch.sigmoid(raw[:, 0:1]), torch.sigmoid(raw[:, 1:2]) * (1.0 - rmin) + rmin, torch.sigmoid(raw[:, 2:5])

    def shade(self, pts, view_dirs, normals, human_poses, is_train, step=None, rand_d=None, rand_s=None, _reg_pts=None):
        from .material_step import MCShade
        kern = self._kern_override if getattr(self, '_kern_override', None) is not None else self._kernels()
        names, eff, K = kern
        scfg = self.shader_network.cfg
        Pn = pts.shape[0]
        x = pts if _reg_pts is None else torch.cat([pts, _reg_pts], 0)
        self._n_shade = Pn
        metallic, rough, albedo = self.predict_materials(x, kern)
        m2 = (metallic[Pn:], rough[Pn:], albedo[Pn:]) if _reg_pts is not None else None
        metallic, rough, albedo = metallic[:Pn], rough[:Pn], albedo[:Pn]
        if is_train and scfg['random_azimuth']:
            rand_d = torch.rand(Pn, 1, 1, device=pts.device) if rand_d is None else rand_d
            rand_s = torch.rand(Pn, 1, 1, device=pts.device) if rand_s is None else rand_s
        else:
            rand_d = rand_s = None
        mat5 = torch.cat([metallic, rough, albedo], -1)
        drv = getattr(self, '_driver', None)
        if drv is not None:
            from .stage2 import MCShadeC
            rgb_lin, dl, sl, sp = MCShadeC.apply(drv, self.ray_tracer, names[40:], getattr(self, '_grad_views', None), pts, view_dirs, normals,
                                                 mat5, rand_d, rand_s, human_poses, *eff[40:])
        else:
            rgb_lin, dl, sl, sp = MCShade.apply(K, self.ray_tracer, names[40:], getattr(self, '_grad_views', None), pts, view_dirs, normals,
                                                mat5, rand_d, rand_s, human_poses, *eff[40:])
        kd = 1 - metallic
        outputs = {
            'rgb_pr': linear_to_srgb(rgb_lin), 'albedo': albedo, 'roughness': rough, 'metallic': metallic,
            'human_lights': torch.zeros(1, 3, device=pts.device),
            'diffuse_light': torch.clamp(linear_to_srgb(dl), 0, 1), 'specular_light': torch.clamp(linear_to_srgb(sl), 0, 1),
            'diffuse_color': torch.clamp(linear_to_srgb(rgb_lin - sp), 0, 1).detach(), 'specular_color': torch.clamp(linear_to_srgb(sp), 0, 1),
            # reference quirk (field.py:1007-1011): uses the already clamped sRGB specular colour
            'approximate_light': torch.clamp(linear_to_srgb(kd * dl + torch.clamp(linear_to_srgb(sp), 0, 1)), 0, 1).detach(),
        }
        if m2 is not None:
            outputs['_reg_materials'] = m2
        return outputs

    def material_regularization(self, pts, normals, metallic, rough, albedo, step, m2):
        """network/field.py:1061-1087 given the materials m2 at the perturbed points"""
        scfg = self.shader_network.cfg
        reg = 0
        if scfg['reg_change']:
            reg = reg + torch.mean((torch.abs(m2[0] - metallic) + torch.abs(m2[1] - rough) + torch.abs(m2[2] - albedo)) * scfg['reg_lambda1'], dim=1)
        hinge = self.material_hinge(rough, metallic, step)
        if hinge is not None:
            reg = reg + hinge
        return reg

    def material_hinge(self, rough, metallic, step):
        return material_hinge(self.shader_network.cfg, rough, metallic, step)

    def regularization_points(self, pts, normals, reg_ang=None, reg_eps=None):
        """tangent-plane perturbation of the surface points (network/field.py:1066-1076)"""
        scfg = self.shader_network.cfg
        n = torch.nn.functional.normalize(normals, dim=-1)
        o0 = torch.stack([n[:, 1], -n[:, 0], torch.zeros_like(n[:, 0])], -1)
        o1 = torch.stack([-n[:, 2], torch.zeros_like(n[:, 0]), n[:, 0]], -1)
        x = torch.nn.functional.normalize(torch.where((o0.norm(dim=-1) > o1.norm(dim=-1)).unsqueeze(-1), o0, o1), dim=-1)
        y = torch.cross(n, x, dim=-1)
        ang = (torch.rand(pts.shape[0], 1, device=pts.device) if reg_ang is None else reg_ang) * np.pi * 2
        if scfg['change_type'] == 'constant':
            eps = scfg['change_eps']
        elif scfg['change_type'] == 'gaussian':
            eps = torch.normal(mean=0.0, std=scfg['change_eps'], size=[pts.shape[0], 1], device=pts.device) if reg_eps is None else reg_eps
        else:
            raise NotImplementedError
        return pts + (torch.cos(ang) * x + torch.sin(ang) * y) * eps

    def compute_rgb_loss(self, rgb_pr, rgb_gt):
        if self.cfg['rgb_loss'] == 'l1':
            return torch.sum(torch.abs(rgb_pr - rgb_gt), -1)
        if self.cfg['rgb_loss'] == 'charbonier':
            return torch.sqrt(torch.sum((rgb_gt - rgb_pr) ** 2, dim=-1) + 0.001)
        raise NotImplementedError

    # ---- dataset pre-trace: every training pixel -> surface sample, resident in HBM (network/renderer.py:756-808)
    def get_human_coordinate_poses(self, poses):
        return NeROShapeRenderer.get_human_coordinate_poses(self, poses)

    def trace_in_batch(self, rays_o, rays_d, batch_size=1024 ** 2):
        outs = [self.trace(rays_o[i:i + batch_size], rays_d[i:i + batch_size]) for i in range(0, rays_o.shape[0], batch_size)]
        return tuple(torch.cat(x, 0) for x in zip(*outs))

    def _trace_views(self, Ks, poses, h, w, device):
        """camera rays of every pixel of every view (pixel centres +0.5, network/renderer.py:756-776), traced through the mesh on the
        device in chunks of 2^20.  -> rays_o, rays_d, inters, normals [imn*h*w,3], depth [imn*h*w,1], hit [imn*h*w]"""
        ys, xs = torch.meshgrid(torch.arange(h, device=device), torch.arange(w, device=device), indexing='ij')
        coords = torch.stack([xs + 0.5, ys + 0.5, torch.ones_like(xs, dtype=torch.float32)], -1).reshape(1, h * w, 3).float()
        Rm, t = poses[:, :, :3], poses[:, :, 3:]
        rays_d = torch.nn.functional.normalize((coords @ torch.inverse(Ks.to(device)).permute(0, 2, 1)) @ Rm, dim=-1)
        rays_o = (-Rm.permute(0, 2, 1) @ t).permute(0, 2, 1).repeat(1, h * w, 1)
        if float(torch.max(torch.norm(rays_o.reshape(-1, 3), dim=-1) + 1.0)) > 10.0:
            print('warning!!! a camera is farther than 10 from the origin: beyond the ray tracer miss distance')
        rays_o, rays_d = rays_o.reshape(-1, 3).contiguous(), rays_d.reshape(-1, 3).contiguous()
        return (rays_o, rays_d) + self.trace_in_batch(rays_o, rays_d)

    def _init_dataset(self, database=None):
        """network/renderer.py:681-698 with a database object of the reference's BaseDatabase interface (see NeROShapeRenderer)."""
        if database is None:
            try:
                from dataset.database import get_database_split, parse_database_name
            except ImportError as e:
                raise ImportError('no database object given and the reference `dataset.database` module is not importable') from e
            database = parse_database_name(self.cfg['database_name'])
            train_ids, test_ids = get_database_split(database, 'validation')
        else:
            ids = list(database.get_img_ids())
            train_ids, test_ids = ids, ids[:1]
        self.database, self.train_ids, self.test_ids = database, np.asarray(train_ids), list(test_ids)

        def info(ids):
            imgs = np.stack([np.asarray(database.get_image(i)) for i in ids], 0).astype(np.float32)
            if imgs.max() > 1.5:
                imgs = imgs / 255.0
            return (torch.from_numpy(imgs), torch.from_numpy(np.stack([database.get_K(i) for i in ids], 0).astype(np.float32)),
                    torch.from_numpy(np.stack([database.get_pose(i) for i in ids], 0).astype(np.float32)))
        self.test_imgs_info = dict(zip(('imgs', 'Ks', 'poses'), info(self.test_ids)))
        self.train_num, self.test_num = len(self.train_ids), len(self.test_ids)
        self.set_ray_pool(*info(train_ids))

    def set_ray_pool(self, imgs, Ks, poses, device=None):
        """imgs [imn,h,w,3] in [0,1], Ks [imn,3,3], poses [imn,3,4].  Traces all imn*h*w camera rays through the mesh on the device
        (chunks of 2^20 like the reference) and keeps the hits as the training pool -- no host round trip, and one human-frame
        pose per IMAGE (indexed per sample) instead of the reference's per-pixel [N,3,4] copy."""
        device = device or next(self.parameters()).device
        imn, h, w, _ = imgs.shape
        poses = poses.to(device).float()
        rays_o, rays_d, inters, normals, depth, hit = self._trace_views(Ks, poses, h, w, device)
        idx = torch.arange(imn, device=device).repeat_interleave(h * w)
        rgb = imgs.to(device).reshape(-1, 3).float()
        self._human_poses_img = self.get_human_coordinate_poses(poses)
        keep = torch.nonzero(hit)[:, 0]
        self.train_batch = {'rays_o': rays_o[keep], 'rays_d': rays_d[keep], 'inters': inters[keep],
                            'normals': normals[keep], 'depth': depth[keep], 'img_idx': idx[keep], 'rgb': rgb[keep]}
        self.tbn = keep.numel()
        if not hasattr(self, 'test_imgs_info'):
            self.test_imgs_info = {'imgs': imgs[:1].cpu().float(), 'Ks': Ks[:1].cpu().float(), 'poses': poses[:1].cpu().float()}
        self._shuffle_train_batch()

    def _shuffle_train_batch(self):
        self.train_batch_i = 0
        perm = torch.randperm(self.tbn, device=self.train_batch['rgb'].device)
        self.train_batch = {k: v[perm] for k, v in self.train_batch.items()}

    def train_step(self, step):
        rn = self.cfg['train_ray_num']
        s = slice(self.train_batch_i, self.train_batch_i + rn)
        b = {k: v[s] for k, v in self.train_batch.items()}
        out = self.shade_train(b['inters'], -b['rays_d'], b['normals'], self._human_poses_img[b['img_idx']], b['rgb'], step)
        self.train_batch_i += rn
        if self.train_batch_i + rn >= self.tbn:
            self._shuffle_train_batch()
        return out

    def test_step(self, index):
        """network/renderer.py:846-887: shade every mesh-hitting pixel of test view `index` (no random azimuth, no grad), zeros
        elsewhere.  -> dict of [h,w,C]: rgb_gt, rgb_pr, specular_light/color, diffuse_light/color, albedo, metallic, roughness
        (square-rooted: the network predicts roughness^2)."""
        info = self.test_imgs_info
        device = next(self.parameters()).device
        img, K, pose = info['imgs'][index:index + 1].float(), info['Ks'][index:index + 1].float(), info['poses'][index:index + 1].float()
        _, h, w, _ = img.shape
        pose = pose.to(device)
        keys = {'rgb_gt': 3, 'rgb_pr': 3, 'specular_light': 3, 'specular_color': 3, 'diffuse_light': 3, 'diffuse_color': 3,
                'albedo': 3, 'metallic': 1, 'roughness': 1}
        out = {k: torch.zeros(h * w, d, device=device) for k, d in keys.items()}
        with torch.no_grad():
            rays_o, rays_d, inters, normals, depth, hit = self._trace_views(K, pose, h, w, device)
            hp = self.get_human_coordinate_poses(pose)
            rgb = img.to(device).reshape(-1, 3)
            trn = self.cfg['test_ray_num']
            for ri in range(0, h * w, trn):
                sel = torch.nonzero(hit[ri:ri + trn])[:, 0] + ri
                if sel.numel() == 0:
                    continue
                so = self.shade(inters[sel].contiguous(), -rays_d[sel], normals[sel].contiguous(), hp[:1].expand(sel.numel(), 3, 4), False)
                out['rgb_gt'][sel] = rgb[sel]
                for k in ('rgb_pr', 'specular_light', 'specular_color', 'diffuse_color', 'diffuse_light', 'albedo', 'metallic'):
                    out[k][sel] = so[k]
                out['roughness'][sel] = torch.sqrt(so['roughness'])
        return {k: v.reshape(h, w, -1) for k, v in out.items()}

    def forward(self, data):
        if 'eval' in data:
            return self.test_step(data['index'])
        return {k: v for k, v in self.train_step(data['step']).items() if not k.startswith('_')}

    def predict_materials_of_vertices(self, vertices, batch_size=8192):
        """extract_materials.py / NeROMaterialRenderer.predict_materials (network/renderer.py:903-915): per-vertex metallic,
        roughness (square-rooted: the network predicts alpha = roughness^2), albedo as numpy arrays"""
        kern = self._kernels()
        out = {'metallic': [], 'roughness': [], 'albedo': []}
        with torch.no_grad():
            for i in range(0, vertices.shape[0], batch_size):
                m, r, a = self.predict_materials(vertices[i:i + batch_size].float().contiguous(), kern)
                out['metallic'].append(m.cpu().numpy())
                out['roughness'].append(torch.sqrt(torch.clamp(r, min=1e-7)).cpu().numpy())
                out['albedo'].append(a.cpu().numpy())
        return {k: np.concatenate(v, 0) for k, v in out.items()}

    def shade_train(self, pts, view_dirs, normals, human_poses, rgb_gt, step, rand_d=None, rand_s=None, reg_ang=None, reg_eps=None):
        """the arithmetic of train_step (network/renderer.py:837-844) for an explicit batch"""
        reg_pts = self.regularization_points(pts, normals, reg_ang, reg_eps) if (self.cfg['reg_mat'] and self.shader_network.cfg['reg_change']) else None
        out = self.shade(pts, view_dirs, normals, human_poses, True, step, rand_d, rand_s, _reg_pts=reg_pts)
        out['rgb_gt'] = rgb_gt
        out['loss_rgb'] = self.compute_rgb_loss(out['rgb_pr'], rgb_gt)
        if self.cfg['reg_mat']:
            out['loss_mat_reg'] = self.material_regularization(pts, normals, out['metallic'], out['roughness'], out['albedo'], step,
                                                               out.pop('_reg_materials', None))
        if self.cfg['reg_diffuse_light']:
            dl = out['diffuse_light']
            out['loss_diffuse_light'] = torch.sum(torch.abs(dl - torch.mean(dl, dim=-1, keepdim=True)), dim=-1) * self.cfg['reg_diffuse_light_lambda']
        return out


name2renderer = {'shape': NeROShapeRenderer, 'material': NeROMaterialRenderer}
