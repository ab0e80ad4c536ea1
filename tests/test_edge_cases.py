"""GPU tier: ragged and degenerate batches through the same oracle comparison as tests/test_parity_at_size.py (outputs <= 1e-4,
loss, every gradient by the fp64 criterion).  The kernels tile rays by 4 per block, rows by 64 per workgroup and weight-gradient
rows by 16-row chunks in 256 slices; the hit / miss split and the inner / outer compaction produce EMPTY partitions for some
batches.  None of the fixtures or the benchmark sizes reach those corners:

  Stage I   R = 1, 7, 63, 65, 130 rays (a partial ray block, a partial 64-row tile, one ray);
            a batch in which half of the rays never enter the unit sphere (zero inner samples on those rays), and one in which
            NO ray does (the inner partition -- SDF, shader, eikonal term, occlusion loss -- is empty and only NeRF++ learns);
  Stage II  P = 1, 5, 63, 65 surface points; a mesh no secondary ray hits (hit partition empty) and a closed shell every
            secondary ray hits (miss partition empty)."""
import numpy as np
import pytest
import torch

from oracle import nero_oracle as O
from tests.test_parity_at_size import BEAR2, BELL, BELL2, ODEV, _free, _material_inputs, _run_shape, _shape_case, _stage2_teacher_forced

pytestmark = pytest.mark.gpu


def _rays_missing_the_sphere(R, seed=0):
    """camera at distance 3, directions whose closest approach to the origin is 1.2 .. 2: every sample of such a ray has |p| > 1"""
    g = torch.Generator().manual_seed(seed)
    az = torch.rand(R, generator=g) * 2 * np.pi
    o = torch.stack([3 * torch.cos(az), 3 * torch.sin(az), 0.3 * torch.ones(R)], -1)
    to_c = -o / o.norm(dim=-1, keepdim=True)
    side = torch.nn.functional.normalize(torch.cross(to_c, torch.tensor([[0.0, 0.0, 1.0]]).expand(R, 3), dim=-1), dim=-1)
    miss = 1.2 + 0.8 * torch.rand(R, generator=g)                          # closest-approach distance
    ang = torch.asin(miss / o.norm(dim=-1))
    d = torch.nn.functional.normalize(torch.cos(ang)[:, None] * to_c + torch.sin(ang)[:, None] * side, dim=-1)
    closest = torch.linalg.cross(o, d).norm(dim=-1)
    assert float(closest.min()) > 1.15
    return o.contiguous(), d.contiguous()


def _rays_entering_the_sphere(R):
    """the first R rays of bench.py's pool generator that pass within 0.8 of the origin"""
    from nero_amd.synthetic import synthetic_rays
    o, d, poses, gt = synthetic_rays(4 * R + 64, seed=1)
    sel = torch.nonzero(torch.linalg.cross(o, d).norm(dim=-1) < 0.8)[:R, 0]
    assert sel.numel() == R
    return o[sel].contiguous(), d[sel].contiguous(), poses[sel].contiguous(), gt[sel].contiguous()


@pytest.mark.parametrize('R', [1, 7, 63, 65, 130])
def test_stage1_ragged_ray_counts(R):
    rec = _run_shape(f'edge_s1_R{R}', BELL, 0.5, R, 25000, with_f64=True, size_asserts=False, small_batch=True, rays=_rays_entering_the_sphere(R))
    assert rec['rays'] == R and rec['n_in'] > 20 * R


def test_stage1_bear_ragged():
    _run_shape('edge_s1_bear_R37', {**BELL, 'shader_config': {'human_light': True}}, 0.5, 37, 25000, with_f64=True, size_asserts=False, small_batch=True,
               rays=_rays_entering_the_sphere(37))


def test_stage1_half_of_the_rays_never_enter_the_sphere():
    from nero_amd.synthetic import synthetic_rays
    R = 96
    o, d, poses, gt = synthetic_rays(R, seed=1)
    om, dm = _rays_missing_the_sphere(R // 2)
    o, d = o.clone(), d.clone()
    o[::2], d[::2] = om, dm                                               # interleaved: ray blocks mix both kinds
    rec = _run_shape('edge_s1_half_outside', BELL, 0.5, R, 25000, with_f64=True, rays=(o, d, poses, gt), size_asserts=False, small_batch=True)
    assert 0 < rec['n_in'] < 100 * (R // 2)


def test_stage1_no_ray_enters_the_sphere():
    """inner partition empty (n_in = 0): SDF, shader, eikonal term and occlusion loss see no sample -- gradient_error is zeros(1) on both
    sides, as in the reference (network/renderer.py:570-577: `else: gradient_error = torch.zeros(1)`) -- and only the NeRF++ network learns"""
    from nero_amd.synthetic import synthetic_rays
    R = 40
    _, _, poses, gt = synthetic_rays(R, seed=1)
    o, d = _rays_missing_the_sphere(R, seed=4)
    rec = _run_shape('edge_s1_no_inner', BELL, 0.5, R, 25000, with_f64=True, rays=(o, d, poses, gt), size_asserts=False, small_batch=True)
    assert rec['n_in'] == 0 and rec['occ_count'] == 0
    # and the own sampler + render on the same rays: same partition, finite colours
    net = _shape_case(BELL, 0.5, device='cuda')
    near, far = O.near_far_from_sphere(o, d)
    with torch.no_grad():
        o2 = net.render(o.cuda(), d.cuda(), near.cuda(), far.cuda(), torch.eye(3, 4).repeat(R, 1, 1).cuda(), 0, 1.0, is_train=True, step=25000)
    assert o2['_state']['n_in'] == 0 and o2['_state']['n_out'] == R * 160 and bool(torch.isfinite(o2['ray_rgb']).all())


def _slice_inputs(I, n):
    return {k: v[:n].contiguous() for k, v in I.items()}


@pytest.mark.parametrize('Pn', [1, 5, 63, 65])
def test_stage2_ragged_point_counts(Pn):
    """outputs and loss against the oracle at P = Pn; the gradients through ADDITIVITY of the production step (C driver + HIP glue):
    the loss is a mean over points, so the gradient of the 96-point batch must equal (Pn g[first Pn] + (96 - Pn) g[rest]) / 96 -- the same
    rows through differently sized launches (a lone partial tile, 1-row weight-gradient slices, ...).  (Against the fp64 oracle a batch
    of a few hundred light rows is dominated by the conditioning of the GGX direction map in fp32: two fp32 evaluations differ from fp64
    by unrelated 1e-3-sized amounts, measured, whichever engine runs the MLPs -- that comparison needs the large batches of
    tests/test_parity_at_size.py.)"""
    from tests.helpers import golden_mesh
    from nero_amd.train import MaterialTrainStep
    scfg = dict(diffuse_sample_num=32, specular_sample_num=32, **BEAR2)
    N = 96
    I = _material_inputs(N)
    rec = _stage2_teacher_forced(f'edge_s2_P{Pn}', Pn, scfg, inputs=_slice_inputs(I, Pn), check_grads=False)
    assert rec['points'] == Pn

    def run(lo, hi):
        n = hi - lo
        pool = {'pts': I['pts'][lo:hi], 'view': I['view'][lo:hi], 'normals': I['normals'][lo:hi], 'rgb': I['gt'][lo:hi]}
        pool = {k: v.contiguous().cuda() for k, v in pool.items()}
        pool['img_idx'] = torch.arange(n, device='cuda')
        ts = MaterialTrainStep({'shader_cfg': scfg, 'database_name': 'real/bear'}, golden_mesh(), points_per_rank=n, device='cuda:0',
                               pool=(pool, I['poses'][lo:hi].contiguous().cuda()))
        assert ts.drv is not None and ts.fused_glue
        rands = {k: I[k][lo:hi].cuda() for k in ('rand_d', 'rand_s', 'reg_ang', 'reg_eps')}
        info = ts.forward_backward(5000, rands)
        torch.cuda.synchronize()
        return info['out']['rgb_pr'].clone(), ts.bucket.flat.clone(), float(info['loss']), ts
    rgb, g, loss, ts = run(0, N)
    rgb_a, g_a, loss_a, _ = run(0, Pn)
    rgb_b, g_b, loss_b, _ = run(Pn, N)
    assert float((torch.cat([rgb_a, rgb_b]) - rgb).abs().max()) < 1e-6
    assert abs((Pn * loss_a + (N - Pn) * loss_b) / N - loss) < 1e-6
    comb = (Pn * g_a.double() + (N - Pn) * g_b.double()) / N
    off, worst = 0, 0.0
    for p in ts.bucket.params:
        a, b = comb[off:off + p.numel()], g[off:off + p.numel()].double()
        off += p.numel()
        if float(b.abs().max()) > 1e-12:
            worst = max(worst, float((a - b).abs().max() / b.abs().max()))
    from tests.helpers import parity_report
    parity_report(f'edge_s2_P{Pn}_additivity', points=Pn, of=N, worst_relative_gradient_difference=worst)
    assert worst < 2e-5, worst


def test_stage2_no_secondary_ray_hits_the_mesh():
    """hit partition empty: the mesh is moved out of reach, every light comes from the outer (and human) MLPs"""
    from tests.helpers import golden_mesh
    v, f = golden_mesh()
    I = _slice_inputs(_material_inputs(96), 80)
    rec = _stage2_teacher_forced('edge_s2_all_miss', 80, dict(diffuse_sample_num=32, specular_sample_num=16, **BEAR2), inputs=I,
                                 mesh=(v + np.array([[50.0, 0.0, 0.0]], dtype=v.dtype), f), small_batch=True)
    assert rec['hit_fraction'] == 0.0


def test_stage2_every_secondary_ray_hits_the_mesh():
    """miss partition empty: a closed shell of radius 4 around the points -- every secondary ray ends on it (inner light only)"""
    from nero_amd.synthetic import icosphere
    v, f = icosphere(3, 4.0, 0.0)
    I = _slice_inputs(_material_inputs(96), 72)
    rec = _stage2_teacher_forced('edge_s2_all_hit', 72, dict(diffuse_sample_num=16, specular_sample_num=32, **BELL2), inputs=I,
                                 mesh=(v, np.ascontiguousarray(f)), small_batch=True)
    assert rec['hit_fraction'] == 1.0


@pytest.mark.parametrize('far', [True, False])
def test_stage2_human_light_rows_follow_the_plane_mask(far, monkeypatch):
    """Round 6: the human-light MLP owns a row only for the miss rays that reach the photographer's region of the camera plane.  far=True: the
    cameras are moved 2000 units away, no ray reaches the region -> zero human-light rows, the step runs, the human-light weights get an exactly
    zero gradient; far=False: some rows.  Both: the same loss / outputs / gradients as the step that gives every miss ray a row (NERO_MC_SKIP_DEAD=0)."""
    from nero_amd.train import MaterialTrainStep
    from tests.helpers import golden_mesh
    N = 64
    I = _material_inputs(N)
    poses = I['poses'].clone()
    if far:
        poses[:, :, 3] = poses[:, :, 3] * 2000.0
    scfg = dict(diffuse_sample_num=32, specular_sample_num=32, **BEAR2)
    res = {}
    for skip in ('1', '0'):
        monkeypatch.setenv('NERO_MC_SKIP_DEAD', skip)
        pool = {'pts': I['pts'], 'view': I['view'], 'normals': I['normals'], 'rgb': I['gt']}
        pool = {k: v.contiguous().cuda() for k, v in pool.items()}
        pool['img_idx'] = torch.arange(N, device='cuda')
        ts = MaterialTrainStep({'shader_cfg': scfg, 'database_name': 'real/bear'}, golden_mesh(), points_per_rank=N, device='cuda:0',
                               pool=(pool, poses.contiguous().cuda()))
        rands = {k: I[k].cuda() for k in ('rand_d', 'rand_s', 'reg_ang', 'reg_eps')}
        info = ts.forward_backward(5000, rands)
        torch.cuda.synchronize()
        n_miss, n_hit, n_rays, n_hum = ts.drv.last_counts
        res[skip] = (float(info['loss']), info['out']['rgb_pr'].clone(), ts.bucket.flat.clone(), (n_miss, n_hit, n_hum), [p.numel() for p in ts.bucket.params])
    (l1, r1, g1, c1, sizes), (l0, r0, g0, c0, _) = res['1'], res['0']
    assert c0[2] == c0[0]                                        # every miss ray owns a row when nothing is skipped
    assert (c1[2] == 0) if far else (0 < c1[2] < c1[0]), c1
    assert abs(l1 - l0) <= 1e-6 * max(1.0, abs(l0)) and float((r1 - r0).abs().max()) <= 1e-6
    off = 0
    for n_ in sizes:
        a, b = g0[off:off + n_], g1[off:off + n_]
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()) + 1e-12
        off += n_
