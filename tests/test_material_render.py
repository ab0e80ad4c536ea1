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
    oo = M.material_train_outputs(P, rcfg, oracle_trace_fn(), T(z, 'pts'), T(z, 'view'), T(z, 'normals'), T(z, 'human_poses'),
                                  T(z, 'gt'), meta['step'], T(z, 'rand_d'), T(z, 'rand_s'), T(z, 'reg_ang'), T(z, 'reg_eps'))
    loss_o = M.material_training_loss(oo)
    loss_o.backward()
    torch.manual_seed(meta['seed'])
    net = NeROMaterialRenderer({'shader_cfg': meta['shader_cfg'], 'database_name': 'syn/bell'}, mesh=golden_mesh())
    net.load_state_dict(ref.state_dict())
    net = net.cuda()
    hip_tracer = net.ray_tracer
    net.ray_tracer = OracleTracer(*golden_mesh())
    c = lambda k: T(z, k, 'cuda')
    out = net.shade_train(c('pts'), c('view'), c('normals'), c('human_poses'), c('gt'), meta['step'], c('rand_d'), c('rand_s'),
                          c('reg_ang'), c('reg_eps'))
    assert rel(out['rgb_pr'], oo['rgb_pr']) < 1e-4
    assert abs(float(rel(out['rgb_pr'], torch.from_numpy(z['rgb'])))) < 1e-4          # and the reference's own output
    for k in ('albedo', 'roughness', 'metallic', 'diffuse_light', 'specular_light', 'specular_color'):
        assert rel(out[k], oo[k]) < 1e-4, k
    assert rel(out['loss_mat_reg'], oo['loss_mat_reg']) < 1e-3
    loss = out['loss_rgb'].mean() + out['loss_mat_reg'].mean() + out['loss_diffuse_light'].mean()
    assert abs(float(loss) - float(loss_o)) < 2e-5
    loss.backward()
    worst = {}
    for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        gq = q.grad if q.grad is not None else torch.zeros_like(q)
        gp = p.grad if p.grad is not None else torch.zeros_like(p)
        if gq.abs().max() < 1e-12 and gp.abs().max() < 1e-12:
            continue
        worst[k] = rel(gp, gq)
    vals = np.array(list(worst.values()))
    # only 24 points x 24 directions: single ReLU-unit sign flips between fp32 evaluation orders show up at the 1e-2 level in the
    # smallest tensors (cf. scripts/dbg_grads64.py for Stage I); typical agreement is 1e-5
    bad = {k: v for k, v in worst.items() if v > 5e-2}
    assert not bad, bad
    print(sorted(worst.items(), key=lambda kv: -kv[1])[:6])
    assert np.quantile(vals, 0.9) < 5e-3 and np.median(vals) < 3e-4, (np.quantile(vals, 0.9), np.median(vals))


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
    assert (err < 1e-4).float().mean() > 0.7 and err.max() < 0.1


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
