"""GPU tier: parity AT THE BENCHMARKED SIZES ("Weak" 1).  The 48-ray fixtures cannot reach the size-dependent code
(multi-slice split-K dW with dw_reduce, head_dw row slicing, > 2^20-row tiles, the 1024-thread ray scan with per > 1), so these
tests run the HIP step against the CPU oracle -- teacher-forced on the oracle's own z_vals, the sampler being ill-conditioned
end to end -- on BASELINE.json's configurations:

  C2        bell Stage I, 4096 rays x (64+64+32) samples, schedule step 25000 (occlusion loss on), exactly bench.py's workload
            (same seed-6033 weights, perturbation, variance 0.5, synthetic rays), with the reference FG table and with the
            product's computed fallback table;
  C3/GPU    bear Stage I (human light), 1024 rays x (64+64+32): the per-GPU share of configs[2];
  C4-shaped bell Stage II, P = 512 surface points x (128+128) MC directions, through the HIP BVH tracer with an explicit,
            oracle-derived exclusion of razor-edge rays, and teacher-forced on the oracle tracer for the gradients.

Outputs <= 1e-4 rel (north_star); gradients by tests/helpers.py::assert_grads_fp32_grade (1e-4 against an fp64 oracle run unless
fp32 torch itself is equally off)."""
import numpy as np
import pytest
import torch

from oracle import nero_oracle as O
from oracle import nero_oracle_mat as M
from tests.helpers import CTracer as _CTracer, tracer_contract as _contract
from tests.helpers import assert_grads_fp32_grade, golden_mesh, named_grads, rel_err

pytestmark = pytest.mark.gpu


def _host_gb():
    try:
        import psutil
        return psutil.virtual_memory().available / 2 ** 30
    except Exception:
        return 64.0


def _shape_case(cfg, variance, R, dtype=torch.float32, device='cpu', seed=6033):
    from nero_amd.renderer import NeROShapeRenderer
    from nero_amd.synthetic import perturb_state
    torch.manual_seed(seed)
    net = NeROShapeRenderer(cfg, training=False)
    perturb_state(net, variance)
    return net.to(dtype).to(device)


def _oracle_step(net, cfg, o, d, z_vals, hp, gt, step, keys, dtype):
    f = lambda a: a.to(dtype)
    sd = {k: v for k, v in net.named_parameters()}
    sd.update({k: v for k, v in net.named_buffers()})
    P = O.effective_params(sd)
    c = {**O.DEFAULT_CFG, **cfg}
    oo = O.render_core(P, c, f(o), f(d), f(z_vals), f(hp), O.anneal(c, step), step, keys)
    loss = O.training_loss(c, oo, f(gt), step)
    loss.backward()
    return oo, loss


def _run_shape(cfg, variance, R, step, with_f64, fallback_lut=False, monkeypatch=None, tmp_path=None):
    from nero_amd.synthetic import synthetic_rays
    from nero_amd.train import shape_training_loss
    if fallback_lut:                                             # construct with NO reference asset in reach: computed table
        monkeypatch.delenv('NERO_FG_LUT', raising=False)
        monkeypatch.chdir(tmp_path)
    o, d, poses, gt = synthetic_rays(R, seed=1)                  # bench.py's pool generator
    ref = _shape_case(cfg, variance, R)
    if fallback_lut:
        from tests.helpers import ref_fg_lut
        assert float((ref.color_network.FG_LUT - ref_fg_lut()).abs().max()) > 1e-3          # really the product default
    c = {**O.DEFAULT_CFG, **cfg}
    hp = ref.get_human_coordinate_poses(poses)
    g = torch.Generator().manual_seed(3)
    rand1, rand_bg, keys = torch.rand(R, 1, generator=g), torch.rand(R, c['n_bg_samples'], generator=g), torch.rand(R * 160, generator=g)
    near, far = O.near_far_from_sphere(o, d)
    with torch.no_grad():
        sd = {k: v.detach() for k, v in ref.state_dict().items()}
        z_vals = O.sample_ray(O.effective_params(sd), c, o, d, near, far, rand1, rand_bg)
    oo, loss_o = _oracle_step(ref, cfg, o, d, z_vals, hp, gt, step, keys, torch.float32)

    net = _shape_case(cfg, variance, R, device='cuda')
    cu = lambda a: a.cuda()
    out = net.render(cu(o), cu(d), cu(near), cu(far), cu(hp), -1, O.anneal(c, step), is_train=True, step=step, z_vals=cu(z_vals),
                     occ_keys=keys)
    n_in = oo['gradient_error'].shape[0]
    assert out['gradient_error'].shape[0] == n_in and n_in > 40 * R        # the size-dependent regime: > 2^17 inner rows at C2
    assert rel_err(out['ray_rgb'], oo['ray_rgb']) < 1e-4
    assert rel_err(out['gradient_error'], oo['gradient_error']) < 1e-4
    assert out['_occ_count'] == oo['occ_count'] > 0
    assert abs(float(out['loss_occ']) - float(oo['loss_occ'])) < 1e-5
    loss = shape_training_loss(net, out, cu(gt), step)
    assert abs(float(loss) - float(loss_o)) < 2e-5, (float(loss), float(loss_o))
    if not with_f64:
        return
    loss.backward()
    ref64 = _shape_case(cfg, variance, R, dtype=torch.float64)
    _oracle_step(ref64, cfg, o, d, z_vals, hp, gt, step, keys, torch.float64)
    rep = assert_grads_fp32_grade(named_grads(net), named_grads(ref), named_grads(ref64), where=f'R={R}')
    worst = sorted(rep.items(), key=lambda kv: -kv[1][0])[:4]
    print(f'[parity@size] R={R} n_in={n_in} loss={float(loss):.6f} worst grads (err_hip, fp32-torch floor): {worst}')


BELL = {'freeze_inv_s_step': 15000, 'apply_occ_loss': True, 'occ_loss_step': 20000}          # == bench.py (configs/shape/syn/bell.yaml)


def test_c2_bell_4096_rays_reference_fg_table():
    R = 4096 if _host_gb() > 90 else 2048             # the fp64 oracle run needs ~46 GB of host memory at 4096 rays
    _run_shape(BELL, 0.5, R, 25000, with_f64=True)


def test_c2_bell_4096_rays_product_default_fg_table(monkeypatch, tmp_path):
    """same workload built the way bench.py builds it on a box without the reference tree: the computed fallback table on both
    sides (the oracle reads the model's FG_LUT buffer)"""
    R = 4096 if _host_gb() > 50 else 2048
    _run_shape(BELL, 0.5, R, 25000, with_f64=False, fallback_lut=True, monkeypatch=monkeypatch, tmp_path=tmp_path)


def test_c3_bear_1024_rays_per_gpu():
    cfg = {**BELL, 'shader_config': {'human_light': True}}                                   # configs/shape/real/bear.yaml
    _run_shape(cfg, 0.5, 1024, 25000, with_f64=True)


# ----------------------------------------------------------------------------------------------------------------------
# Stage II at P = 512 x (128 + 128)
# ----------------------------------------------------------------------------------------------------------------------
class _Recording:
    def __init__(self, inner):
        self.inner, self.depth, self.o, self.d = inner, [], [], []

    def trace(self, o, d):
        out = self.inner.trace(o, d)
        self.depth.append(out[2].detach().cpu().numpy().reshape(-1))
        self.o.append(o.detach().cpu().numpy())
        self.d.append(d.detach().cpu().numpy())
        return out


def _material_inputs(Pn, seed=5):
    """surface points = camera-ray hits on the golden mesh (like oracle/gen_golden.py::run_material_case)"""
    from nero_amd.synthetic import synthetic_rays
    from oracle.tracer_oracle import trace_bruteforce_margins
    v, f = golden_mesh()
    o, d, poses_img, gt = synthetic_rays(6 * Pn, seed=seed, window=120)
    pos, nrm, depth, tri, amb = trace_bruteforce_margins(v, f, o.numpy(), d.numpy())
    sel = np.nonzero((tri >= 0) & ~amb)[0][:Pn]
    assert sel.shape[0] == Pn
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    n = torch.nn.functional.normalize(-t(nrm[sel]), dim=-1)
    g = torch.Generator().manual_seed(3)
    return dict(pts=t(pos[sel]), view=-d[sel], normals=n, poses=poses_img[sel], gt=gt[sel], rand_d=torch.rand(Pn, 1, 1, generator=g),
                rand_s=torch.rand(Pn, 1, 1, generator=g), reg_ang=torch.rand(Pn, 1, generator=g),
                reg_eps=torch.normal(mean=0.0, std=0.05, size=[Pn, 1], generator=g))


def _material_pair(shader_cfg, dtype=torch.float32):
    from tests.helpers import MatHolder
    from nero_amd.synthetic import perturb_state
    torch.manual_seed(6033)
    ref = MatHolder(shader_cfg)
    perturb_state(ref, None)
    return ref.to(dtype)


@pytest.mark.parametrize('shader_cfg', [dict(diffuse_sample_num=128, specular_sample_num=128, human_lights=False, outer_light_version='direction'),
                                        dict(diffuse_sample_num=64, specular_sample_num=64, human_lights=True,
                                             outer_light_version='sphere_direction')],
                         ids=['bell_D256', 'bear_D128'])
def test_c4_shaped_stage2_teacher_forced_tracer(shader_cfg):
    """P = 512, both sides fed the oracle tracer's hits: outputs, losses and every gradient"""
    from nero_amd.renderer import NeROMaterialRenderer
    Pn, step = 512, 5000
    I = _material_inputs(Pn)
    hp = None
    rcfg = {'shader_cfg': shader_cfg}

    tracers = {}

    def oracle(dtype):
        ref = _material_pair(shader_cfg, dtype)
        sd = {k: v for k, v in ref.named_parameters()}
        sd.update({k: v for k, v in ref.named_buffers()})
        f = lambda a: a.to(dtype)
        tr = tracers[dtype] = _CTracer(*golden_mesh(), replay=tracers.get(torch.float32))    # the fp64 run replays the fp32 run's hits
        from nero_amd.renderer import NeROShapeRenderer
        hpl = NeROShapeRenderer.get_human_coordinate_poses(type('c', (), {'cfg': {'fixed_camera': False}})(), I['poses'])
        oo = M.material_train_outputs(O.effective_params(sd), rcfg, _contract(tr), f(I['pts']), f(I['view']), f(I['normals']), f(hpl),
                                      f(I['gt']), step, f(I['rand_d']), f(I['rand_s']), f(I['reg_ang']), f(I['reg_eps']))
        loss = M.material_training_loss(oo)
        loss.backward()
        return ref, oo, loss, hpl
    ref, oo, loss_o, hp = oracle(torch.float32)
    net = NeROMaterialRenderer({'shader_cfg': shader_cfg, 'database_name': 'syn/bell'}, mesh=golden_mesh())
    net.load_state_dict(ref.state_dict())
    net = net.cuda()
    net.ray_tracer = _CTracer(*golden_mesh())
    c = lambda k: I[k].cuda()
    out = net.shade_train(c('pts'), c('view'), c('normals'), hp.cuda(), c('gt'), step, c('rand_d'), c('rand_s'), c('reg_ang'), c('reg_eps'))
    assert rel_err(out['rgb_pr'], oo['rgb_pr']) < 1e-4
    for k in ('albedo', 'roughness', 'metallic', 'diffuse_light', 'specular_light', 'specular_color'):
        assert rel_err(out[k], oo[k]) < 1e-4, k
    assert rel_err(out['loss_mat_reg'], oo['loss_mat_reg']) < 1e-3
    loss = out['loss_rgb'].mean() + out['loss_mat_reg'].mean() + out['loss_diffuse_light'].mean()
    assert abs(float(loss) - float(loss_o)) < 2e-5
    loss.backward()
    ref64 = oracle(torch.float64)[0]
    rep = assert_grads_fp32_grade(named_grads(net), named_grads(ref), named_grads(ref64), where='stage2 P=512')
    print('[parity@size] stage II worst grads:', sorted(rep.items(), key=lambda kv: -kv[1][0])[:4], 'hit fraction', float(oo['hit_fraction']))


def test_c4_shaped_stage2_hip_tracer_with_explicit_edge_exclusion():
    """The same P = 512 x 256 step with the secondary rays traced by the HIP BVH.  A float32 tracer may answer differently from
    the fp64 oracle only on rays the oracle itself flags as razor-edge (a triangle edge within 1e-4 barycentric units of deciding
    the closest hit, or a candidate intersection within 5e-6 of the ray origin -- the rays start 1e-5 off the surface, so the
    triangle they left sits at t = -1e-5 exactly and is never such a candidate).  So:
    (1) every ray on which the two tracers disagree is such a ray; (2) the points that own no flagged ray -- counted, reported,
    and required to be the large majority -- match the oracle shading to 1e-4."""
    from nero_amd.renderer import NeROMaterialRenderer
    from nero_amd.renderer import NeROShapeRenderer
    shader_cfg = dict(diffuse_sample_num=128, specular_sample_num=128, human_lights=False, outer_light_version='direction')
    Pn, D, step = 512, 256, 5000
    I = _material_inputs(Pn)
    ref = _material_pair(shader_cfg)
    sd = {k: v.detach() for k, v in ref.state_dict().items()}
    tr = _CTracer(*golden_mesh(), eps_edge=1e-4, eps_t=5e-6)
    hp = NeROShapeRenderer.get_human_coordinate_poses(type('c', (), {'cfg': {'fixed_camera': False}})(), I['poses'])
    with torch.no_grad():
        rgb_o, oo = M.mc_shade(O.effective_params(sd), {**M.DEFAULT_SHADER_CFG, **shader_cfg}, _contract(tr), I['pts'], I['view'], I['normals'],
                               hp, I['rand_d'], I['rand_s'])
    amb, hit_o = np.concatenate(tr.amb), np.concatenate(tr.hit)
    assert amb.shape[0] == Pn * D
    net = NeROMaterialRenderer({'shader_cfg': shader_cfg, 'database_name': 'syn/bell'}, mesh=golden_mesh())
    net.load_state_dict(ref.state_dict())
    net = net.cuda()
    rec = _Recording(net.ray_tracer)
    net.ray_tracer = rec
    c = lambda k: I[k].cuda()
    with torch.no_grad():
        out = net.shade(c('pts'), c('view'), c('normals'), hp.cuda(), True, step, c('rand_d'), c('rand_s'))
    hit_h = np.concatenate(rec.depth) < 10.0
    assert hit_h.shape == hit_o.shape
    differ = hit_h != hit_o
    if (differ & ~amb).any():                                   # say what the unexplained rays look like before failing
        from oracle.tracer_oracle import trace_bruteforce_margins
        idx = np.nonzero(differ & ~amb)[0][:8]
        ro, rd = np.concatenate(rec.o)[idx], np.concatenate(rec.d)[idx]
        loose = trace_bruteforce_margins(*golden_mesh(), ro, rd, eps_edge=1e-3, eps_t=9e-6)
        raise AssertionError(f'{int((differ & ~amb).sum())} rays differ between the HIP BVH and the fp64 oracle without being razor-edge: rows '
                             f'{idx}, oracle depth on the HIP rays {loose[2]}, flagged at (1e-3, 9e-6): {loose[4]}, hip depth '
                             f'{np.concatenate(rec.depth)[idx]}, oracle depth on its own rays {np.concatenate([r[2] for r in tr.raw])[idx]}')
    pt_amb = amb.reshape(Pn, D).any(axis=1)
    ok = torch.from_numpy(~pt_amb)
    print(f'[parity@size] stage II / HIP tracer: {int(amb.sum())} razor-edge rays of {Pn * D} ({int(differ.sum())} answered differently), '
          f'{int(pt_amb.sum())} of {Pn} points excluded; hit fraction {hit_o.mean():.3f}')
    assert ok.float().mean() > 0.6
    scale = float(rgb_o.abs().max())
    perr = (out['rgb_pr'].cpu() - rgb_o).abs().max(-1)[0] / scale
    good = perr[ok]
    print(f'[parity@size] non-excluded points: {int((good < 1e-4).sum())} of {good.numel()} within 1e-4, worst {float(good.max()):.2e}; '
          f'excluded points worst {float(perr[~ok].max()) if pt_amb.any() else 0.0:.2e}')
    # Hit / miss patterns agree on every non-flagged ray (asserted above), so what remains on the non-excluded points is the float32
    # vs float64 hit POSITION (|dx| ~ 1e-7, amplified 2^7-fold by PE-8 into the inner-light MLP) and, on rays that graze a shared
    # edge, the choice between two coplanar-depth triangles with different normals: a handful of points at the 1e-3 level.
    assert float((good < 1e-4).float().mean()) > 0.97, float((good < 1e-4).float().mean())
    assert float(good.max()) < 1e-2, float(good.max())
