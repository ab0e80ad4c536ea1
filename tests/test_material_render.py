"""GPU tier: Stage-II MC shading on the HIP path (BVH tracer + light MLPs + microfacet estimator) vs the CPU oracle
(oracle/nero_oracle_mat.py, pinned to the unmodified reference by tests/test_oracle_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import nero_oracle as O
from oracle import nero_oracle_mat as M
from tests.helpers import T, build_material_case, golden_mesh, load_golden, oracle_trace_fn

pytestmark = pytest.mark.gpu


class OracleTracer:
    """RayTracer-shaped wrapper over the brute-force fp64 oracle, so that the shader parity test sees exactly the hit/miss pattern
    the golden run saw (secondary rays start 1e-5 off the surface: razor-edge self-intersections are tracer-precision dependent)."""

    def __init__(self, v, f):
        self.v, self.f = v, f

    def trace(self, o, d):
        from oracle.tracer_oracle import trace_bruteforce
        pos, nrm, depth, _ = trace_bruteforce(self.v, self.f, o.detach().cpu().numpy(), d.detach().cpu().numpy())
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
        return f(pos), f(nrm), f(depth)


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize('name', ['mat_bell', 'mat_bell_early', 'mat_bear', 'mat_bell_smith'])
def test_mc_shading_outputs_loss_and_grads(name):
    from nero_amd.renderer import NeROMaterialRenderer
    z, meta = load_golden(name)
    ref = build_material_case(meta)
    sd = {k: v for k, v in ref.named_parameters()}
    sd.update({k: v for k, v in ref.named_buffers()})
    P = O.effective_params(sd)
    rcfg = {'shader_cfg': meta['shader_cfg']}
    from tests.helpers import CTracer, tracer_contract
    tr32 = CTracer(*golden_mesh())
    oo = M.material_train_outputs(P, rcfg, tracer_contract(tr32), T(z, 'pts'), T(z, 'view'), T(z, 'normals'), T(z, 'human_poses'),
                                  T(z, 'gt'), meta['step'], T(z, 'rand_d'), T(z, 'rand_s'), T(z, 'reg_ang'), T(z, 'reg_eps'))
    loss_o = M.material_training_loss(oo)
    loss_o.backward()
    torch.manual_seed(meta['seed'])
    net = NeROMaterialRenderer({'shader_cfg': meta['shader_cfg'], 'database_name': 'syn/bell'}, mesh=golden_mesh())
    net.load_state_dict(ref.state_dict())
    net = net.cuda()
    hip_tracer = net.ray_tracer
    net.ray_tracer = CTracer(*golden_mesh())
    c = lambda k: T(z, k, 'cuda')
    # (the ReLU sign masks and L1 sign decisions of THIS forward, for the gated fallback below)
    from nero_amd import chain as CH
    CH.MASK_CAPTURE = []
    signs = {}
    inner_reg = net.material_regularization

    def reg_spy(pts_, nrm_, metallic, rough, albedo, step_, m2):
        if m2 is not None:
            for nm, a, b in (('metallic', m2[0], metallic), ('roughness', m2[1], rough), ('albedo', m2[2], albedo)):
                signs[f'abs/reg_{nm}'] = torch.sign((a - b).detach())
        return inner_reg(pts_, nrm_, metallic, rough, albedo, step_, m2)
    net.material_regularization = reg_spy
    try:
        out = net.shade_train(c('pts'), c('view'), c('normals'), c('human_poses'), c('gt'), meta['step'], c('rand_d'), c('rand_s'),
                              c('reg_ang'), c('reg_eps'))
        capture = CH.MASK_CAPTURE
    finally:
        CH.MASK_CAPTURE = None
        net.material_regularization = inner_reg
    if 'loss_diffuse_light' in out:
        dl_ = out['diffuse_light'].detach()
        signs['abs/diffuse_light'] = torch.sign(dl_ - torch.mean(dl_, dim=-1, keepdim=True))
    assert rel(out['rgb_pr'], oo['rgb_pr']) < 1e-4
    assert abs(float(rel(out['rgb_pr'], torch.from_numpy(z['rgb'])))) < 1e-4          # and the reference's own output
    for k in ('albedo', 'roughness', 'metallic', 'diffuse_light', 'specular_light', 'specular_color'):
        assert rel(out[k], oo[k]) < 1e-4, k
    assert rel(out['loss_mat_reg'], oo['loss_mat_reg']) < 1e-3
    loss = out['loss_rgb'].mean() + out['loss_mat_reg'].mean() + out['loss_diffuse_light'].mean()
    assert abs(float(loss) - float(loss_o)) < 2e-5
    loss.backward()
    # gradients: within 1e-4 of an fp64 oracle run unless fp32 torch itself is equally off (tests/helpers.py)
    from tests.helpers import assert_grads_fp32_grade, named_grads
    ref64 = build_material_case(meta).double()
    sd64 = {k: v for k, v in ref64.named_parameters()}
    sd64.update({k: v for k, v in ref64.named_buffers()})
    d = lambda k: T(z, k).double()
    o64 = M.material_train_outputs(O.effective_params(sd64), rcfg, tracer_contract(CTracer(*golden_mesh(), replay=tr32)), d('pts'), d('view'), d('normals'),
                                   d('human_poses'), d('gt'), meta['step'], d('rand_d'), d('rand_s'), d('reg_ang'), d('reg_eps'))
    M.material_training_loss(o64).backward()
    try:
        assert_grads_fp32_grade(named_grads(net), named_grads(ref), named_grads(ref64), where=name)
        return
    except AssertionError as e:
        ungated = str(e)
    # A ReLU unit of a light MLP within rounding of zero fell on the other side in the HIP forward than in both torch evaluations (24 points:
    # one flipped unit moves a bias gradient by 1e-2 of its maximum).  Decided by ARITHMETIC, as at the benchmarked sizes
    # (tests/test_parity_at_size.py): the oracle in fp64 and fp32 takes the HIP forward's own gate / L1-sign decisions and the plain 1e-4
    # must hold for every tensor.
    from tests.helpers import forced_gates_from_capture
    from tests.test_parity_at_size import _forced_gate_errors
    Pn = z['pts'].shape[0]
    gates = {**forced_gates_from_capture(capture, 2, Pn), **signs}

    def gated(dtype):
        m = build_material_case(meta).to(dtype)
        sdm = {k: v for k, v in m.named_parameters()}
        sdm.update({k: v for k, v in m.named_buffers()})
        f = lambda k: T(z, k).to(dtype)
        with O.forced_relu_gates(gates) as fg:
            og = M.material_train_outputs(O.effective_params(sdm), rcfg, tracer_contract(CTracer(*golden_mesh(), replay=tr32)), f('pts'), f('view'),
                                          f('normals'), f('human_poses'), f('gt'), meta['step'], f('rand_d'), f('rand_s'), f('reg_ang'), f('reg_eps'))
            M.material_training_loss(og).backward()
            assert fg.used == set(gates), sorted(set(gates) - fg.used)[:5]
        return {k: v.cpu() for k, v in named_grads(m).items()}
    fe = _forced_gate_errors({k: v.cpu() for k, v in named_grads(net).items()}, gated(torch.float32), gated(torch.float64))
    # (24 points, a handful of hit rays: the inner-light gradients are cancellation-limited -- fp32 torch under the SAME gates is 5e-4 from
    #  fp64 there; such tensors count as `fp32_floor` (HIP within 3 x of fp32 torch's own error), anything else fails)
    assert fe['n_unexplained'] == 0 and fe['n_fp32_floor'] <= 12, (ungated[:300], fe)


def test_mc_shading_with_hip_tracer_close_to_oracle():
    """same shader, secondary rays through the HIP BVH: identical up to the handful of razor-edge rays"""
    from nero_amd.renderer import NeROMaterialRenderer
    z, meta = load_golden('mat_bell')
    ref = build_material_case(meta)
    net = NeROMaterialRenderer({'shader_cfg': meta['shader_cfg'], 'database_name': 'syn/bell'}, mesh=golden_mesh())
    net.load_state_dict(ref.state_dict())
    net = net.cuda()
    c = lambda k: T(z, k, 'cuda')
    with torch.no_grad():
        out = net.shade(c('pts'), c('view'), c('normals'), c('human_poses'), True, meta['step'], c('rand_d'), c('rand_s'))
    err = (out['rgb_pr'].cpu() - torch.from_numpy(z['rgb'])).abs().max(-1)[0]
    # 24 points x 24 directions: the explicit razor-edge accounting is tests/test_parity_at_size.py (P = 512 x 256).  Measured on
    # MI355X (profiles/r03_parity_at_size.json, 'mat_bell_24x24_hip_bvh'): all 24 points within 1e-4, worst 1.2e-6 -- no ray of this
    # fixed case sits on a razor edge.  The bounds leave room for ONE point whose ray is answered differently (1 of 24 directions).
    from tests.helpers import parity_report
    frac, worst = float((err < 1e-4).float().mean()), float(err.max())
    parity_report('mat_bell_24x24_hip_bvh', fraction_points_within_1e4=frac, worst_point=worst)
    assert frac >= 23 / 24 - 1e-6 and worst < 2e-2


def test_inference_packs_the_shader_network_once_until_a_parameter_changes():
    """NeROMaterialRenderer._kernels under no_grad: one packing for all chunks of a test_step (it re-packed per 1024-ray
    chunk); an in-place parameter update invalidates the cache and the outputs follow the new weights"""
    from nero_amd.renderer import NeROMaterialRenderer
    z, meta = load_golden('mat_bell')
    ref = build_material_case(meta)
    net = NeROMaterialRenderer({'shader_cfg': meta['shader_cfg'], 'database_name': 'syn/bell'}, mesh=golden_mesh())
    net.load_state_dict(ref.state_dict())
    net = net.cuda()
    c = lambda k: T(z, k, 'cuda')
    with torch.no_grad():
        k1 = net._kernels()
        out1 = net.shade(c('pts'), c('view'), c('normals'), c('human_poses'), True, meta['step'], c('rand_d'), c('rand_s'))
        k2 = net._kernels()
        assert k2[2] is k1[2]                                          # the same packed chains served both calls
        out2 = net.shade(c('pts'), c('view'), c('normals'), c('human_poses'), True, meta['step'], c('rand_d'), c('rand_s'))
        assert torch.equal(out1['rgb_pr'], out2['rgb_pr'])
        p = next(q for n, q in net.shader_network.named_parameters() if 'albedo' in n and n.endswith('bias'))
        p.add_(0.25)                                                   # in place: torch's version counter moves
        k3 = net._kernels()
        assert k3[2] is not k1[2]
        out3 = net.shade(c('pts'), c('view'), c('normals'), c('human_poses'), True, meta['step'], c('rand_d'), c('rand_s'))
        assert float((out3['albedo'] - out1['albedo']).abs().max()) > 1e-3
    k4 = net._kernels()                                                # with autograd on nothing is cached: fresh effective-weight leaves
    assert k4[2] is not k3[2]


def test_material_trainer_entry_point_and_pretrace():
    """forward({'step':...}) over the device-side pre-traced pixel pool (camera rays through the HIP BVH) + per-vertex materials"""
    from nero_amd.renderer import NeROMaterialRenderer
    from nero_amd.synthetic import icosphere, look_at_pose
    v, f = icosphere(4, 0.5, 0.15)
    f = np.ascontiguousarray(f[:, ::-1])
    torch.manual_seed(0)
    net = NeROMaterialRenderer({'shader_cfg': dict(diffuse_sample_num=32, specular_sample_num=16, human_lights=True,
                                                   outer_light_version='sphere_direction'),
                                'database_name': 'real/x', 'train_ray_num': 128}, mesh=(v, f)).cuda()
    rg = np.random.default_rng(0)
    imgs = torch.from_numpy(rg.uniform(0, 1, (2, 48, 48, 3)).astype(np.float32))
    K = torch.tensor([[60., 0, 24], [0, 60., 24], [0, 0, 1]]).repeat(2, 1, 1)
    poses = torch.from_numpy(np.stack([look_at_pose(np.array(c, dtype=np.float64)) for c in ([2.5, 0, 0.5], [0, 2.5, 1.0])], 0))
    net.set_ray_pool(imgs, K, poses)
    assert 0.15 * 2 * 48 * 48 < net.tbn < 0.9 * 2 * 48 * 48
    out = net({'step': 100})
    for k in ('rgb_pr', 'rgb_gt', 'loss_rgb', 'loss_mat_reg', 'loss_diffuse_light', 'albedo', 'roughness', 'metallic', 'diffuse_light',
              'specular_light', 'diffuse_color', 'specular_color', 'approximate_light', 'human_lights'):
        assert k in out, k
    (out['loss_rgb'].mean() + out['loss_mat_reg'].mean() + out['loss_diffuse_light'].mean()).backward()
    assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
    mats = net.predict_materials_of_vertices(torch.from_numpy(v).cuda())
    assert mats['albedo'].shape == (v.shape[0], 3) and np.isfinite(mats['roughness']).all()


def test_stage2_pretrace_pool_vs_reference_construct_ray_batch():
    """set_ray_pool (device-side pre-trace through the HIP BVH) against NeROMaterialRenderer._construct_ray_batch of the
    unmodified reference run behind the fp64 tracer oracle on the same 3-view toy database (tests/golden/pools.npz, made by
    oracle/gen_golden_r2.py): hit set, ray origins / directions, surface points, normals, depths, colours, human frames"""
    import os
    from nero_amd.renderer import NeROMaterialRenderer
    from tests.helpers import GOLDEN
    z = np.load(os.path.join(GOLDEN, 'pools.npz'))
    torch.manual_seed(0)
    net = NeROMaterialRenderer({'shader_cfg': dict(diffuse_sample_num=16, specular_sample_num=8, human_lights=True),
                                'database_name': 'real/x', 'train_ray_num': 64, 'test_ray_num': 50}, mesh=golden_mesh()).cuda()
    net._shuffle_train_batch = lambda: setattr(net, 'train_batch_i', 0)           # keep construction order
    net.set_ray_pool(torch.from_numpy(z['imgs']), torch.from_numpy(z['Ks']), torch.from_numpy(z['poses']))
    assert net.tbn == z['s2/rays_o'].shape[0]                                     # identical hit set (360 camera rays, 86 hits)
    tb = {k: v.cpu().numpy() for k, v in net.train_batch.items()}
    assert np.abs(tb['rays_o'] - z['s2/rays_o']).max() < 2e-6 and np.abs(tb['rays_d'] - z['s2/rays_d']).max() < 2e-6
    assert np.abs(tb['inters'] - z['s2/inters']).max() < 2e-5
    assert np.abs(tb['normals'] - z['s2/normals']).max() < 2e-5
    assert np.abs(tb['depth'] - z['s2/depth']).max() < 2e-5
    assert np.array_equal(tb['rgb'], z['s2/rgb'])
    hp = net._human_poses_img[net.train_batch['img_idx']].cpu().numpy()            # one frame per image, indexed per sample
    assert np.abs(hp - z['s2/human_poses']).max() < 1e-6
    # test_step (network/renderer.py:846-887) on the pool's first view: shapes, zeros off the mesh, finite on it
    ev = net({'eval': True, 'index': 0, 'step': 0})
    h, w = z['imgs'].shape[1:3]
    for k, c in (('rgb_gt', 3), ('rgb_pr', 3), ('specular_light', 3), ('specular_color', 3), ('diffuse_light', 3), ('diffuse_color', 3),
                 ('albedo', 3), ('metallic', 1), ('roughness', 1)):
        assert ev[k].shape == (h, w, c), k
    hit0 = torch.from_numpy(z['s2t/hit_mask'])                                      # reference hit mask of view 1
    ev1_pose = net.test_imgs_info                                                   # (pool given directly: validates on view 0)
    assert torch.isfinite(ev['rgb_pr']).all() and float(ev['rgb_pr'].abs().sum()) > 0
    off = ev['rgb_gt'].abs().sum(-1) == 0
    assert float(ev['rgb_pr'][off].abs().sum()) == 0.0
