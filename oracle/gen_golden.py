"""Golden-vector generator: runs the UNMODIFIED reference (/root/reference) under oracle/ref_shim.py on CPU and dumps
small fixtures to tests/golden/.  Run in the build container only:   python oracle/gen_golden.py

TEST INFRASTRUCTURE ONLY (see oracle/nero_oracle.py header).  /root/reference does not exist on the GPU box; the
committed .npz files are what travels.

Weights are NOT stored (8.8 MB each): they are reproduced by   torch.manual_seed(seed) -> build the module tree
(same torch init calls in the same order as the reference ctor) -> perturb_state(...)   and verified against the
per-tensor checksums stored here (tests/test_oracle_golden.py::test_weight_reproduction).
"""
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import ref_shim  # noqa: E402
from oracle.golden_util import perturb_state, synthetic_rays, grad_digest, state_checksums  # noqa: E402

OUT = os.path.join(REPO, 'tests', 'golden')


def run_case(name, cfg, R, step, variance, seed=6033, occ_keys_seed=None):
    renderer, field = ref_shim.load_reference()
    torch.manual_seed(seed)
    net = renderer.NeROShapeRenderer(cfg, training=False)
    perturb_state(net, variance)
    net.train()
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}

    o, d, poses_img, gt = synthetic_rays(R, seed=1, window=200)
    near, far = net.near_far_from_sphere(o, d)
    # renderer.py:240-256; called one pose at a time: the reference writes through an expanded tensor (Y[:,2]=-1),
    # which torch-CPU rejects for pn>1 (it is benign on CUDA)
    hp = torch.cat([net.get_human_coordinate_poses(poses_img[i:i + 1].clone()) for i in range(R)], 0)
    anneal = float(net.get_anneal_val(step))

    # the reference draws rand[R,1] then rand[R,n_bg] from the global generator (renderer.py:416,422)
    torch.manual_seed(3)
    rand1 = torch.rand([R, 1])
    rand_bg = torch.rand([R, net.cfg['n_bg_samples']])

    # stage-wise trace of the sampler, driving the reference's own functions round by round
    trace = {}
    searched = []
    _ss = torch.searchsorted

    def ss(*a, **k):
        r = _ss(*a, **k)
        searched.append(r.clone())
        return r
    sorted_idx = []
    _sort = torch.sort

    def srt(*a, **k):
        r = _sort(*a, **k)
        sorted_idx.append(r[1].clone())
        return r
    torch.manual_seed(3)
    torch.searchsorted, torch.sort = ss, srt
    try:
        with torch.no_grad():
            z_ref = net.sample_ray(o, d, near, far, net.cfg['perturb'])
    finally:
        torch.searchsorted, torch.sort = _ss, _sort
    for i, (a, b) in enumerate(zip(searched, sorted_idx)):
        trace[f'inds{i}'] = a.numpy().astype(np.int32)
        trace[f'index{i}'] = b.numpy().astype(np.int32)

    # occ-loss subset selection: make the reference's randperm "argsort of our keys" so it is reproducible
    occ_keys = None
    _rp = torch.randperm
    if occ_keys_seed is not None:
        g = torch.Generator().manual_seed(occ_keys_seed)
        occ_keys = torch.rand(R * 160, generator=g)
        torch.randperm = lambda n, **k: torch.argsort(occ_keys[:n], stable=True)
    try:
        torch.manual_seed(3)
        net.zero_grad()
        out = net.render(o, d, near, far, hp, -1, anneal, is_train=True, step=step)
    finally:
        torch.randperm = _rp
    out['loss_rgb'] = net.compute_rgb_loss(out['ray_rgb'], gt)
    # trainer loss assembly (train/trainer.py:127-137, network/loss.py): eikonal 0.1, occ, (init_sdf_reg for step<1000)
    loss = out['loss_rgb'].mean() + (out['gradient_error'] * 0.1).mean() + out['loss_occ'].mean()
    if step < 1000:                                                       # InitSDFRegLoss, network/loss.py:90-122
        from network.loss import InitSDFRegLoss
        for k, v in InitSDFRegLoss(cfg)(out, None, step).items():
            loss = loss + torch.mean(v)
    loss.backward()
    grads = {k: p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)
             for k, p in net.named_parameters()}

    rec = dict(
        meta=json.dumps(dict(name=name, cfg=cfg, R=R, step=step, variance=variance, seed=seed, anneal=anneal,
                             occ_keys_seed=occ_keys_seed)),
        o=o.numpy(), d=d.numpy(), poses_img=poses_img.numpy(), human_poses=hp.numpy(), gt=gt.numpy(),
        near=near.numpy(), far=far.numpy(), rand1=rand1.numpy(), rand_bg=rand_bg.numpy(),
        z_vals=z_ref.numpy(), ray_rgb=out['ray_rgb'].detach().numpy(),
        gradient_error=out['gradient_error'].detach().numpy(), std=np.float32(out['std'].detach().numpy()),
        loss_occ=np.float32(out['loss_occ'].detach().numpy().reshape(-1)[0]), loss=np.float32(loss.item()),
    )
    if occ_keys is not None:
        rec['occ_keys'] = occ_keys.numpy()
    for k, v in state_checksums(sd).items():
        rec['ck/' + k] = v
    for k, v in grad_digest(grads).items():
        rec['gd/' + k] = v
    rec.update({'tr/' + k: v for k, v in trace.items()})
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **rec)
    print(name, 'loss', loss.item(), 'N_in', out['gradient_error'].shape[0], 'loss_occ', float(out['loss_occ']))


def run_validation_case(name, cfg, R, step, variance, seed=6033):
    """is_train=False render of the unmodified reference (test_step's inner call, renderer.py:304): all validation outputs"""
    renderer, field = ref_shim.load_reference()
    torch.manual_seed(seed)
    net = renderer.NeROShapeRenderer(cfg, training=False)
    perturb_state(net, variance)
    o, d, poses_img, gt = synthetic_rays(R, seed=1, window=200)
    near, far = net.near_far_from_sphere(o, d)
    hp = torch.cat([net.get_human_coordinate_poses(poses_img[i:i + 1].clone()) for i in range(R)], 0)
    with torch.no_grad():
        out = net.render(o, d, near, far, hp, 0, 0, is_train=False, step=step)
    rec = dict(meta=json.dumps(dict(name=name, cfg=cfg, R=R, step=step, variance=variance, seed=seed, anneal=0.0)),
               o=o.numpy(), d=d.numpy(), near=near.numpy(), far=far.numpy(), human_poses=hp.numpy())
    for k, v in out.items():
        if torch.is_tensor(v):
            rec['out/' + k] = v.detach().numpy()
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **rec)
    print(name, sorted(k for k in out.keys()))


def run_material_case(name, shader_cfg, P_, step, seed=6033):
    """MCShadingNetwork.forward + material_regularization + diffuse-light regulariser of the unmodified reference on surface points
    of a bumpy icosphere, traced by the brute-force oracle tracer behind NeROMaterialRenderer.trace's contract."""
    import torch.nn as nn
    from nero_amd.synthetic import icosphere
    from oracle.tracer_oracle import trace_bruteforce
    renderer, field = ref_shim.load_reference()
    verts, tris = icosphere(3, 0.5, 0.15)
    tris = np.ascontiguousarray(tris[:, ::-1])      # inward winding like a NeuS-extracted mesh: NeRO flips the tracer's normals (renderer.py:722)

    def trace(o, d):                                                     # network/renderer.py:719-729
        pos, nrm, depth, _ = trace_bruteforce(verts, tris, o.detach().numpy(), d.detach().numpy())
        nrm = torch.from_numpy(-nrm).float()
        nrm = torch.nn.functional.normalize(nrm, dim=-1)
        depth = torch.from_numpy(depth).float().reshape(-1, 1)
        return torch.from_numpy(pos).float(), nrm, depth, (depth < 10)[:, 0]

    class Holder(nn.Module):
        pass
    torch.manual_seed(seed)
    net = Holder()
    net.shader_network = field.MCShadingNetwork(shader_cfg, trace)
    perturb_state(net, None)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    # surface points: camera rays that hit the mesh
    o, d, poses_img, gt = synthetic_rays(4 * P_, seed=5, window=120)
    pos, nrm, depth, hit = trace(o, d)
    sel = torch.nonzero(hit)[:P_, 0]
    assert sel.numel() == P_
    pts, view, normals, gt = pos[sel], -d[sel], nrm[sel], gt[sel]
    rr = renderer.NeROShapeRenderer.__new__(renderer.NeROShapeRenderer)   # only for get_human_coordinate_poses (same code in both renderers)
    rr.cfg = {'fixed_camera': False}
    hp = torch.cat([renderer.NeROShapeRenderer.get_human_coordinate_poses(rr, poses_img[sel][i:i + 1].clone()) for i in range(P_)], 0)
    torch.manual_seed(3)
    rand_d, rand_s = torch.rand(P_, 1, 1), torch.rand(P_, 1, 1)
    reg_ang = torch.rand(P_, 1)
    reg_eps = torch.normal(mean=0.0, std=0.05, size=[P_, 1])
    torch.manual_seed(3)
    rgb, out = net.shader_network(pts, view, normals, hp, step, True)
    loss_rgb = torch.sqrt(torch.sum((gt - rgb) ** 2, dim=-1) + 1e-3)
    reg = net.shader_network.material_regularization(pts, normals, out['metallic'], out['roughness'], out['albedo'], step)
    dl = out['diffuse_light']
    white = torch.sum(torch.abs(dl - torch.mean(dl, dim=-1, keepdim=True)), dim=-1) * 0.1
    loss = loss_rgb.mean() + reg.mean() + white.mean()
    loss.backward()
    grads = {k: p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p) for k, p in net.named_parameters()}
    rec = dict(meta=json.dumps(dict(name=name, shader_cfg=shader_cfg, P=P_, step=step, seed=seed)),
               pts=pts.numpy(), view=view.numpy(), normals=normals.numpy(), human_poses=hp.numpy(), gt=gt.numpy(),
               rand_d=rand_d.numpy(), rand_s=rand_s.numpy(), reg_ang=reg_ang.numpy(), reg_eps=reg_eps.numpy(),
               rgb=rgb.detach().numpy(), loss=np.float32(loss.item()), loss_mat_reg=reg.detach().numpy(), loss_white=white.detach().numpy())
    for k in ('albedo', 'roughness', 'metallic', 'diffuse_light', 'specular_light', 'diffuse_color', 'specular_color', 'approximate_light'):
        rec['out/' + k] = out[k].detach().numpy()
    for k, v in state_checksums(sd).items():
        rec['ck/' + k] = v
    for k, v in grad_digest(grads).items():
        rec['gd/' + k] = v
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **rec)
    print(name, 'loss', loss.item(), 'reg', float(reg.mean()), 'white', float(white.mean()))


def unit_vectors():
    """Small per-function vectors from the reference's own encoders / helpers."""
    renderer, field = ref_shim.load_reference()
    from utils.ref_utils import generate_ide_fn
    from utils.raw_utils import linear_to_srgb
    g = torch.Generator().manual_seed(11)
    dirs = torch.nn.functional.normalize(torch.randn(64, 3, generator=g), dim=-1)
    rough = torch.rand(64, 1, generator=g)
    ide_fn = generate_ide_fn(5)
    x = torch.randn(64, 3, generator=g)
    pe6, _ = field.get_embedder(6, 3)
    pe10, _ = field.get_embedder(10, 4)
    x4 = torch.randn(64, 4, generator=g)
    lin = torch.cat([torch.rand(32, 3, generator=g) * 0.01, torch.rand(32, 3, generator=g) * 2], 0)
    lut = torch.from_numpy(np.fromfile('assets/bsdf_256_256.bin', dtype=np.float32).reshape(1, 256, 256, 2))
    uv = torch.rand(1, 64, 1, 2, generator=g)
    uv[0, :4, 0, :] = torch.tensor([[0., 0.], [1., 1.], [0., 1.], [0.999, 0.001]])
    import nvdiffrast.torch as dr
    fg = dr.texture(lut, uv, filter_mode='linear', boundary_mode='clamp').reshape(64, 2)
    # sample_pdf on synthetic bins / weights
    bins = torch.sort(torch.rand(16, 65, generator=g), -1)[0]
    w = torch.rand(16, 64, generator=g) ** 4
    _ss = torch.searchsorted
    got = []

    def ss(*a, **k):
        r = _ss(*a, **k)
        got.append(r.clone())
        return r
    torch.searchsorted = ss
    try:
        smp = field.sample_pdf(bins, w, 16, det=True)
    finally:
        torch.searchsorted = _ss
    np.savez_compressed(
        os.path.join(OUT, 'units.npz'),
        dirs=dirs.numpy(), rough=rough.numpy(), ide_rough=ide_fn(dirs, rough).numpy(), ide_one=ide_fn(dirs, 1.0 * torch.ones(64, 1)).numpy(),
        x=x.numpy(), pe6=pe6(x).numpy(), x4=x4.numpy(), pe10=pe10(x4).numpy(),
        lin=lin.numpy(), srgb=linear_to_srgb(lin).numpy(), uv=uv.reshape(64, 2).numpy(), fg=fg.numpy(),
        bins=bins.numpy(), w=w.numpy(), samples=smp.numpy(), inds=got[0].numpy().astype(np.int32))
    # the reference's FG table (also part of every reference checkpoint as buffer color_network.FG_LUT)
    np.savez_compressed(os.path.join(OUT, 'fg_lut_ref.npz'), lut=lut.numpy().reshape(256, 256, 2))
    print('units ok')


if __name__ == '__main__':
    unit_vectors()
    small = dict(n_samples=16, n_importance=16, n_bg_samples=8, up_sample_steps=4)
    run_case('bell_s25000', dict(small), R=48, step=25000, variance=0.3)
    run_case('bell_s5000_sharp', dict(small, freeze_inv_s_step=15000), R=48, step=5000, variance=0.55)
    run_case('bear_s25000', dict(small, shader_config={'human_light': True}), R=48, step=25000, variance=0.4)
    run_case('bell_occcap', dict(small, occ_loss_max_pn=24), R=48, step=25000, variance=0.5, occ_keys_seed=5)
    run_case('bell_c1', dict(n_samples=32, n_importance=32, n_bg_samples=32), R=32, step=25000, variance=0.3)
    run_case('bell_s500', dict(small, freeze_inv_s_step=15000), R=48, step=500, variance=0.3)
    run_case('bell_sphdir', dict(small, shader_config={'sphere_direction': True}), R=48, step=25000, variance=0.3)
    run_validation_case('bell_val', dict(small), R=48, step=25000, variance=0.5)
    run_validation_case('bear_val', dict(small, shader_config={'human_light': True}), R=48, step=25000, variance=0.5)
    msmall = dict(diffuse_sample_num=16, specular_sample_num=8)
    run_material_case('mat_bell', dict(msmall, human_lights=False, outer_light_version='direction'), P_=24, step=5000)
    run_material_case('mat_bell_early', dict(msmall, human_lights=False, outer_light_version='direction'), P_=24, step=500)
    run_material_case('mat_bear', dict(msmall, human_lights=True, outer_light_version='sphere_direction'), P_=24, step=5000)
    run_material_case('mat_bell_smith', dict(msmall, human_lights=False, outer_light_version='direction', geometry_type='ggx_smith'), P_=24, step=5000)
