"""GPU tier: parity AT THE BENCHMARKED SIZES (VERDICT r1 "Weak" 1, VERDICT r2 "Weak" 1-4).  The 48-ray fixtures cannot reach the
size-dependent code (multi-slice split-K dW with dw_reduce, head_dw row slicing, > 2^20-row tiles, the 1024-thread ray scan), so
these tests run the HIP step against the oracle -- teacher-forced on the oracle's own z_vals / tracer hits, the sampler being
ill-conditioned end to end -- on BASELINE.json's configurations, at the sizes bench.py measures:

  C2        bell Stage I, 4096 rays x (64+64+32) samples, schedule step 25000 (occlusion loss on): exactly bench.py's workload
            (same seed-6033 weights, perturbation, variance 0.5, synthetic rays), with the reference FG table and with the
            product's computed fallback table.  ALWAYS 4096 rays (round 2 silently halved it on small hosts);
  C3/GPU    bear Stage I (human light), 1024 rays x (64+64+32): the per-GPU share of configs[2];
  C4        bell Stage II, P = 4096 surface points x (128+128) AND x (512+256, the YAML default) MC directions: bench.py's two
            `stage2` legs (1.05 M / 3.1 M light rows per launch), teacher-forced on the C tracer oracle;
  C5/GPU    bear Stage II (sphere_direction + human_lights), P = 2048 x (256+256): the per-GPU share of configs[4];
  C4/BVH    bell Stage II, P = 512 x 256 through the HIP BVH tracer with an explicit, oracle-derived exclusion of razor-edge rays.

WHERE THE ORACLE RUNS.  oracle/nero_oracle*.py is plain torch code, pinned on the CPU against the reference's dumps by
tests/test_oracle_golden.py.  Here the SAME functions are evaluated through ATen's GPU backend (fp32 and fp64 tensors on cuda:0):
an fp64 oracle pass over 3.1 M light rows takes ~1 s there and minutes (plus ~100 GB of RAM) on the box's host cores, and GPU-box
minutes are budgeted.  What decides pass / fail is the fp64 run; the fp32 run only supplies the "fp32 torch is equally far from fp64"
floor of clause (a) below, and how often that clause is used is counted and bounded.  The brute-force tracer oracle stays on the
host (C, OpenMP).

Outputs <= 1e-4 rel (north_star); gradients by tests/helpers.py::assert_grads_fp32_grade (1e-4 against the fp64 oracle run, with
two counted escape clauses).  Every test appends what it really ran (R / P, n_in, razor-edge counts, clause counts, worst errors) to
gpurun_out/parity_at_size.json and asserts bounds on those numbers."""
import gc

import numpy as np
import pytest
import torch

from oracle import nero_oracle as O
from oracle import nero_oracle_mat as M
from tests.helpers import CTracer as _CTracer, tracer_contract as _contract
from tests.helpers import assert_grads_fp32_grade, golden_mesh, named_grads, parity_report, rel_err

pytestmark = pytest.mark.gpu

ODEV = 'cuda'                    # where the oracle's ATen ops execute (see the module docstring)

# bounds on the escape clauses of assert_grads_fp32_grade, from the measured counts (gpurun_out/parity_at_size.json of the round-3
# runs) plus margin: (b) is reserved for first-layer predictor tensors with |g| ~ 1e-6 of their MLP's scale, (a) for ReLU-tie flips
# measured (MI355X, round 3): clause (b) 0 tensors in every case; clause (a) 10-27 of 84-124 tensors (light / NeRF++ MLPs: ReLU ties)
MAX_CLAUSE_B = 2
MAX_CLAUSE_A = 32
# gate-teacher-forced runs (round 4): tensors beyond the plain 1e-4.  Measured on MI355X: Stage II 0 of 84 / 96 in every case (worst 1.5e-5
# ... 5.2e-5); Stage I 2 of 124 / 136 -- outer_nerf.alpha_linear.{weight, bias} at 2.2e-3 / 2.9e-3, with fp32 torch UNDER THE SAME GATES
# 2.2e-3 / 3.0e-3 from fp64: the NeRF++ density gradient is a sum of T_k (c_k - C_behind_k) over background samples of nearly equal
# colour, a cancellation that amplifies the 1e-6 fp32 noise of the colours themselves; no gate and no kernel.  `unexplained` (beyond
# 1e-4 AND beyond 3x the forced fp32 floor) must be empty.
STAGE1_FLOOR_TENSORS = {'outer_nerf.alpha_linear.weight', 'outer_nerf.alpha_linear.bias'}
MAX_FORCED_FLOOR_STAGE2 = 2


def _free():
    gc.collect()
    torch.cuda.empty_cache()


def _shape_case(cfg, variance, dtype=torch.float32, device='cpu', seed=6033):
    from nero_amd.renderer import NeROShapeRenderer
    from nero_amd.synthetic import perturb_state
    torch.manual_seed(seed)
    net = NeROShapeRenderer(cfg, training=False)           # constructed + perturbed on the CPU: identical weights on every side
    perturb_state(net, variance)
    return net.to(dtype).to(device)


def _forced_gate_errors(g_hip, g32f, g64f, tol=1e-4, floor_factor=3.0):
    """Gradients under FORCED ReLU gates: the fp64 oracle (g64f) and the fp32 oracle (g32f) both took the HIP forward's decisions, so what
    separates the three is arithmetic.  Per tensor err = max|a - f64| / max|f64|.  A tensor is `plain` when err(hip) <= tol; the
    remainder is split into `fp32_floor` -- fp32 torch under the same gates is as far from fp64 (err(hip) <= floor_factor x the largest
    err(torch32) of its MLP): an ill-conditioned sum, not a kernel -- and `unexplained`, which must be empty."""
    floors, errs = {}, {}
    for k, g in g64f.items():
        if float(g.abs().max()) < 1e-12 and float(g_hip[k].abs().max()) < 1e-12:
            continue
        errs[k] = rel_err(g_hip[k], g)
        from tests.helpers import _mlp_of
        floors[_mlp_of(k)] = max(floors.get(_mlp_of(k), 0.0), rel_err(g32f[k], g))
    from tests.helpers import _mlp_of
    floor_of = lambda k: floors[_mlp_of(k)]
    plain = [k for k, e in errs.items() if e <= tol]
    at_floor = {k: (e, floor_of(k)) for k, e in errs.items() if not e <= tol and e <= floor_factor * floor_of(k)}
    unexplained = {k: (e, floor_of(k)) for k, e in errs.items() if not e <= tol and not e <= floor_factor * floor_of(k)}
    vals = np.array(list(errs.values()))
    return dict(n_tensors=len(errs), n_plain=len(plain), n_fp32_floor=len(at_floor), n_unexplained=len(unexplained),
                fp32_floor={k: [float(a), float(b)] for k, (a, b) in at_floor.items()},
                unexplained={k: [float(a), float(b)] for k, (a, b) in unexplained.items()},
                median_err=float(np.median(vals)), max_err=float(vals.max()), worst=sorted(errs.items(), key=lambda t: -t[1])[:4])


def _oracle_step(net, cfg, o, d, z_vals, hp, gt, step, keys, dtype, ODEV=ODEV, gates=None):
    """one oracle forward + loss + backward in `dtype` on ODEV.  -> (small outputs on the CPU, loss, named grads)
    gates: ReLU decisions to force (oracle.nero_oracle.forced_relu_gates) -- every key must be consumed"""
    if gates is not None:
        with O.forced_relu_gates(gates) as fg:
            res = _oracle_step(net, cfg, o, d, z_vals, hp, gt, step, keys, dtype, ODEV)
            assert fg.used == set(gates), sorted(set(gates) - fg.used)[:5]
        return res
    f = lambda a: a.to(ODEV).to(dtype)
    sd = {k: v for k, v in net.named_parameters()}
    sd.update({k: v for k, v in net.named_buffers()})
    P = O.effective_params(sd)
    c = {**O.DEFAULT_CFG, **cfg}
    with torch.device(ODEV):
        oo = O.render_core(P, c, f(o), f(d), f(z_vals), f(hp), O.anneal(c, step), step, keys.to(ODEV))
        loss = O.training_loss(c, oo, f(gt), step)
        loss.backward()
    small = {k: (oo[k].detach().cpu() if torch.is_tensor(oo[k]) else oo[k]) for k in ('ray_rgb', 'gradient_error', 'loss_occ', 'occ_count', 'n_inner')
             if k in oo}
    g = {k: v.cpu() for k, v in named_grads(net).items()}
    loss = float(loss)
    del oo, P, sd
    _free()
    return small, loss, g


def _run_shape(test_id, cfg, variance, R, step, with_f64, fallback_lut=False, monkeypatch=None, tmp_path=None, cpu_floor=False, rays=None,
               size_asserts=True, small_batch=False):
    """rays: optional (o, d, poses, gt) instead of bench.py's pool generator; size_asserts=False drops the 'this IS the big regime' checks
    (tests/test_edge_cases.py runs ragged and degenerate batches through the same comparison)"""
    from nero_amd.synthetic import synthetic_rays
    from nero_amd.train import shape_training_loss
    if fallback_lut:                                             # construct with NO reference asset in reach: computed table
        monkeypatch.delenv('NERO_FG_LUT', raising=False)
        monkeypatch.chdir(tmp_path)
    o, d, poses, gt = rays if rays is not None else synthetic_rays(R, seed=1)                  # bench.py's pool generator
    ref = _shape_case(cfg, variance, device=ODEV)
    if fallback_lut:
        from tests.helpers import ref_fg_lut
        assert 1e-5 < float((ref.color_network.FG_LUT.cpu() - ref_fg_lut()).abs().max()) < 5e-4    # really the product default (computed: close to the asset, not the asset)
    c = {**O.DEFAULT_CFG, **cfg}
    hp = ref.get_human_coordinate_poses(poses)
    g = torch.Generator().manual_seed(3)
    rand1, rand_bg, keys = torch.rand(R, 1, generator=g), torch.rand(R, c['n_bg_samples'], generator=g), torch.rand(R * 160, generator=g)
    near, far = O.near_far_from_sphere(o, d)
    with torch.no_grad(), torch.device(ODEV):
        sd = {k: v.detach() for k, v in ref.state_dict().items()}
        cu = lambda a: a.to(ODEV)
        z_vals = O.sample_ray(O.effective_params(sd), c, cu(o), cu(d), cu(near), cu(far), cu(rand1), cu(rand_bg)).cpu()
    oo, loss_o, g32 = _oracle_step(ref, cfg, o, d, z_vals, hp, gt, step, keys, torch.float32)
    del ref
    _free()

    net = _shape_case(cfg, variance, device='cuda')
    cu = lambda a: a.cuda()
    from nero_amd import chain as CH
    CH.MASK_CAPTURE = [] if with_f64 else None                     # the ReLU sign masks of this forward (gate-forced gradient parity below)
    try:
        out = net.render(cu(o), cu(d), cu(near), cu(far), cu(hp), -1, O.anneal(c, step), is_train=True, step=step, z_vals=cu(z_vals),
                         occ_keys=keys)
        capture = CH.MASK_CAPTURE
    finally:
        CH.MASK_CAPTURE = None
    n_in = oo['n_inner']                                          # (an empty inner partition yields gradient_error = zeros(1) on both sides)
    rec = dict(rays=R, n_in=int(n_in), oracle_device=ODEV, occ_count=int(out.get('_occ_count', 0)),
               err_ray_rgb=rel_err(out['ray_rgb'], oo['ray_rgb']), err_gradient_error=rel_err(out['gradient_error'], oo['gradient_error']))
    parity_report(test_id, **rec)
    assert out['gradient_error'].shape[0] == max(n_in, 1) and out['_state']['n_in'] == n_in
    assert n_in > 40 * R or not size_asserts                               # the size-dependent regime: > 2^17 inner rows at C2
    assert rec['err_ray_rgb'] < 1e-4
    assert rec['err_gradient_error'] < 1e-4
    assert out.get('_occ_count', 0) == oo.get('occ_count', 0) and (oo.get('occ_count', 0) > 0 or not size_asserts)
    assert abs(float(out['loss_occ']) - float(oo['loss_occ'])) < 1e-5
    loss = shape_training_loss(net, out, cu(gt), step)
    assert abs(float(loss) - loss_o) < 2e-5, (float(loss), loss_o)
    if not with_f64:
        return rec
    loss.backward()
    g_hip = {k: v.cpu() for k, v in named_grads(net).items()}
    from tests.helpers import forced_gates_from_capture
    gates = forced_gates_from_capture(capture, 1, n_in) if n_in > 0 else None
    del net, out, loss, capture
    _free()
    ref64 = _shape_case(cfg, variance, dtype=torch.float64, device=ODEV)
    _, _, g64 = _oracle_step(ref64, cfg, o, d, z_vals, hp, gt, step, keys, torch.float64)
    if gates is not None and not small_batch:
        # GATE-TEACHER-FORCED: the same fp64 oracle, every ReLU of the light / material / NeRF++ MLPs taking the decision the HIP
        # forward took (its saved sign masks).  What is left between the two gradients is arithmetic, not tie-breaking: the plain
        # 1e-4 of north_star must then hold for EVERY tensor, no escape clause.
        ref64.zero_grad(set_to_none=True)
        _, _, g64f = _oracle_step(ref64, cfg, o, d, z_vals, hp, gt, step, keys, torch.float64, gates=gates)
        ref32 = _shape_case(cfg, variance, device=ODEV)
        _, _, g32f = _oracle_step(ref32, cfg, o, d, z_vals, hp, gt, step, keys, torch.float32, gates=gates)
        rec['forced_gates'] = _forced_gate_errors(g_hip, g32f, g64f)
        del g64f, g32f, ref32
    del ref64, gates
    _free()
    g32s = [g32]
    if cpu_floor:
        # a second fp32 evaluation of the oracle, on ATen's CPU backend: which ReLU ties an fp32 run resolves differently from fp64
        # depends on its GEMM summation order, so "how far is fp32 torch from fp64" is the spread over both backends
        ref_cpu = _shape_case(cfg, variance)
        g32s.append(_oracle_step(ref_cpu, cfg, o, d, z_vals, hp, gt, step, keys, torch.float32, 'cpu')[2])
    info = {}
    if small_batch:
        from tests.helpers import assert_grads_small_batch
        assert_grads_small_batch(g_hip, g32s, g64, where=f'{test_id} R={R}', info=info)
        rec.update(info)
        parity_report(test_id, **rec)
        return rec
    assert_grads_fp32_grade(g_hip, g32s, g64, where=f'{test_id} R={R}', info=info)
    rec.update(info, fp32_floor_backends=['cuda', 'cpu'] if cpu_floor else ['cuda'])
    parity_report(test_id, **rec)
    assert info['n_clause_b'] <= MAX_CLAUSE_B and info['n_clause_a'] <= MAX_CLAUSE_A, info
    if 'forced_gates' in rec:
        assert rec['forced_gates']['n_unexplained'] == 0 and set(rec['forced_gates']['fp32_floor']) <= STAGE1_FLOOR_TENSORS, rec['forced_gates']
    return rec


BELL = {'freeze_inv_s_step': 15000, 'apply_occ_loss': True, 'occ_loss_step': 20000}          # == bench.py (configs/shape/syn/bell.yaml)


def test_c2_bell_4096_rays_reference_fg_table():
    _run_shape('c2_bell_4096_ref_fg', BELL, 0.5, 4096, 25000, with_f64=True)


def test_c2_bell_4096_rays_product_default_fg_table(monkeypatch, tmp_path):
    """same workload built the way bench.py builds it on a box without the reference tree: the computed fallback table on both
    sides (the oracle reads the model's FG_LUT buffer)"""
    _run_shape('c2_bell_4096_fallback_fg', BELL, 0.5, 4096, 25000, with_f64=False, fallback_lut=True, monkeypatch=monkeypatch,
               tmp_path=tmp_path)


def test_c3_bear_1024_rays_per_gpu():
    cfg = {**BELL, 'shader_config': {'human_light': True}}                                   # configs/shape/real/bear.yaml
    # (the human-light MLP is driven by the few rays that hit the camera plane: one flipped ReLU tie there is a 1e-4-sized share of its
    # whole gradient, so this case takes its fp32-torch floor over both ATen backends)
    _run_shape('c3_bear_1024', cfg, 0.5, 1024, 25000, with_f64=True, cpu_floor=True)


def test_c2_end_to_end_without_teacher_forcing_statistics():
    """C2 (4096 rays) with the product's OWN hierarchical sampler -- no oracle z_vals handed over.  The sampler is ill-conditioned end
    to end (one SDF value within rounding of a section boundary moves a sample into the neighbouring section), so index
    equality with another fp32 evaluation cannot be demanded; this test MEASURES how far apart two fp32 evaluations land at the
    benchmarked size, records it, and bounds it: the fraction of z values equal to 1e-5, the fraction of rays with all 128 inner z equal,
    and the resulting ray colours (median / 99th percentile / worst)."""
    from nero_amd.synthetic import synthetic_rays
    R, cfg = 4096, BELL
    c = {**O.DEFAULT_CFG, **cfg}
    o, d, poses, gt = synthetic_rays(R, seed=1)
    g = torch.Generator().manual_seed(3)
    rand1, rand_bg, keys = torch.rand(R, 1, generator=g), torch.rand(R, c['n_bg_samples'], generator=g), torch.rand(R * 160, generator=g)
    near, far = O.near_far_from_sphere(o, d)
    ref = _shape_case(cfg, 0.5, device=ODEV)
    hp = ref.get_human_coordinate_poses(poses)
    cu = lambda a: a.to(ODEV)
    with torch.no_grad(), torch.device(ODEV):
        sd = {k: v.detach() for k, v in ref.state_dict().items()}
        P = O.effective_params(sd)
        zo = O.sample_ray(P, c, cu(o), cu(d), cu(near), cu(far), cu(rand1), cu(rand_bg))
        rgb_o = O.render_core(P, c, cu(o), cu(d), zo, cu(hp), O.anneal(c, 25000), 25000, keys.to(ODEV))['ray_rgb'].cpu()
        zo = zo.cpu()
    del ref, P, sd
    _free()
    net = _shape_case(cfg, 0.5, device='cuda')
    with torch.no_grad():
        zg = net.sample_ray(o.cuda(), d.cuda(), near.cuda(), far.cuda(), 1.0, rand1.cuda(), rand_bg.cuda())
        out = net.render(o.cuda(), d.cuda(), near.cuda(), far.cuda(), hp.cuda(), -1, O.anneal(c, 25000), is_train=True, step=25000, z_vals=zg)
    zg = zg.cpu()
    nb = c['n_bg_samples']
    dz = (zg[:, :-nb] - zo[:, :-nb]).abs()
    e = (out['ray_rgb'].cpu() - rgb_o).abs().max(-1)[0]
    rec = dict(rays=R, frac_z_equal_1e5=float((dz < 1e-5).float().mean()), frac_rays_all_z_equal_1e5=float((dz.max(-1)[0] < 1e-5).float().mean()),
               worst_dz=float(dz.max()), bg_z_rel=float((zg[:, -nb:] / zo[:, -nb:] - 1).abs().max()),
               rgb_abs_err_median=float(e.median()), rgb_abs_err_p99=float(e.kthvalue(int(0.99 * R))[0]), rgb_abs_err_worst=float(e.max()))
    parity_report('c2_bell_4096_end_to_end_no_teacher_forcing', **rec)
    assert rec['bg_z_rel'] < 1e-6                                   # the background z do not depend on the SDF: exact arithmetic
    # measured on MI355X (round 3): 98.86 % of the z equal, 90.0 % of the rays entirely equal (worst dz 3.5e-3: one sample in a neighbouring
    # section); colours: median 1.2e-7, 99th percentile 4.8e-7, worst ray 1.0e-4 -- the rendering integral barely notices a moved sample
    assert rec['frac_z_equal_1e5'] > 0.98 and rec['frac_rays_all_z_equal_1e5'] > 0.85, rec
    assert rec['rgb_abs_err_median'] < 2e-6 and rec['rgb_abs_err_p99'] < 1e-5 and rec['rgb_abs_err_worst'] < 1e-3, rec


# ----------------------------------------------------------------------------------------------------------------------
# Stage II at the benchmarked sizes
# ----------------------------------------------------------------------------------------------------------------------
class _Recording:
    def __init__(self, inner):
        self.inner, self.depth, self.o, self.d, self.pos, self.nrm = inner, [], [], [], [], []

    def trace(self, o, d):
        out = self.inner.trace(o, d)
        self.depth.append(out[2].detach().cpu().numpy().reshape(-1))
        self.pos.append(out[0].detach().cpu().numpy())
        self.nrm.append(out[1].detach().cpu().numpy())
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


def _material_pair(shader_cfg, dtype=torch.float32, device='cpu'):
    from tests.helpers import MatHolder
    from nero_amd.synthetic import perturb_state
    torch.manual_seed(6033)
    ref = MatHolder(shader_cfg)
    perturb_state(ref, None)
    return ref.to(dtype).to(device)


def _stage2_teacher_forced(test_id, Pn, shader_cfg, step=5000, inputs=None, mesh=None, small_batch=False, check_grads=True):
    """both sides fed the oracle tracer's hits: outputs, losses and every gradient, oracle in fp32 + fp64 on ODEV.
    inputs / mesh: optional point set and (vertices, triangles) instead of camera-ray hits on the golden mesh (tests/test_edge_cases.py)"""
    from nero_amd.renderer import NeROMaterialRenderer, NeROShapeRenderer
    I = inputs if inputs is not None else _material_inputs(Pn)
    mesh = mesh if mesh is not None else golden_mesh()
    rcfg = {'shader_cfg': shader_cfg}
    D = shader_cfg['diffuse_sample_num'] + shader_cfg['specular_sample_num']
    hpl = NeROShapeRenderer.get_human_coordinate_poses(type('c', (), {'cfg': {'fixed_camera': False}})(), I['poses'])
    tracers = {}
    keys = ('rgb_pr', 'albedo', 'roughness', 'metallic', 'diffuse_light', 'specular_light', 'specular_color', 'loss_mat_reg')

    def oracle(dtype, gates=None):
        if gates is not None:
            with O.forced_relu_gates(gates) as fg:
                res = oracle(dtype)
                assert fg.used == set(gates), sorted(set(gates) - fg.used)[:5]
            return res
        ref = _material_pair(shader_cfg, dtype, ODEV)
        sd = {k: v for k, v in ref.named_parameters()}
        sd.update({k: v for k, v in ref.named_buffers()})
        f = lambda a: a.to(ODEV).to(dtype)
        tr = _CTracer(*mesh, replay=tracers.get('src'))          # every later run (fp64, forced gates) replays the FIRST fp32 run's hits
        tracers.setdefault('src', tr)
        with torch.device(ODEV):
            oo = M.material_train_outputs(O.effective_params(sd), rcfg, _contract(tr), f(I['pts']), f(I['view']), f(I['normals']), f(hpl),
                                          f(I['gt']), step, f(I['rand_d']), f(I['rand_s']), f(I['reg_ang']), f(I['reg_eps']))
            loss = M.material_training_loss(oo)
            loss.backward()
        small = {k: oo[k].detach().cpu() for k in keys}
        hit_fraction = float(oo['hit_fraction'])
        g = {k: v.cpu() for k, v in named_grads(ref).items()}
        state = {k: v.detach().cpu() for k, v in ref.state_dict().items()}
        loss = float(loss)
        del oo, sd, ref
        _free()
        return small, loss, g, hit_fraction, state
    oo, loss_o, g32, hit_fraction, state = oracle(torch.float32)
    net = NeROMaterialRenderer({'shader_cfg': shader_cfg, 'database_name': 'syn/bell'}, mesh=mesh)
    net.load_state_dict({k: v.float() for k, v in state.items()})
    net = net.cuda()
    # teacher forcing: the HIP step is handed the hits the fp32 oracle run obtained for the same (point, direction) slots -- a ray of
    # the 1-3 M that grazes an edge must not flip between the two sides -- and its own secondary rays (origins p + 1e-5 w, GGX /
    # cosine directions: nero_mc_dirs) are REQUIRED to equal the oracle's to 2e-5
    net.ray_tracer = _CTracer(*mesh, replay=tracers['src'], ray_tol=2e-5)
    c = lambda k: I[k].cuda()
    from nero_amd import chain as CH
    CH.MASK_CAPTURE = [] if check_grads else None
    signs = {}                                                     # the L1 terms' sign decisions of this forward (forced like the ReLU gates)
    inner_reg = net.material_regularization

    def reg_spy(pts_, nrm_, metallic, rough, albedo, step_, m2):
        if m2 is not None:
            for nm, a, b in (('metallic', m2[0], metallic), ('roughness', m2[1], rough), ('albedo', m2[2], albedo)):
                signs[f'abs/reg_{nm}'] = torch.sign((a - b).detach())
        return inner_reg(pts_, nrm_, metallic, rough, albedo, step_, m2)
    net.material_regularization = reg_spy
    try:
        out = net.shade_train(c('pts'), c('view'), c('normals'), hpl.cuda(), c('gt'), step, c('rand_d'), c('rand_s'), c('reg_ang'), c('reg_eps'))
        capture = CH.MASK_CAPTURE
    finally:
        CH.MASK_CAPTURE = None
        net.material_regularization = inner_reg
    if 'loss_diffuse_light' in out:
        dl_ = out['diffuse_light'].detach()
        signs['abs/diffuse_light'] = torch.sign(dl_ - torch.mean(dl_, dim=-1, keepdim=True))
    rec = dict(points=Pn, directions=D, light_rows=Pn * D, hit_fraction=hit_fraction, oracle_device=ODEV,
               max_secondary_ray_deviation=net.ray_tracer.max_ray_dev, errs={k: rel_err(out[k], oo[k]) for k in keys})
    parity_report(test_id, **rec)
    assert rec['errs']['rgb_pr'] < 1e-4, rec
    for k in ('albedo', 'roughness', 'metallic', 'diffuse_light', 'specular_light', 'specular_color'):
        assert rec['errs'][k] < 1e-4, (k, rec)
    # loss_mat_reg is |m(p) - m(p + eps)| of two nearly equal predictions: a cancellation that amplifies the 1e-6 of the predictions
    # themselves.  It is held to the plain 1e-4 against the FLOAT64 oracle, or to 3 x the distance the float32 oracle itself has from
    # float64 (VERDICT r4 'weak' 3: rounds 3-4 compared against the float32 oracle and had to allow 1e-3)
    reg_hip = out['loss_mat_reg'].detach().cpu()
    o64 = oracle(torch.float64)
    reg64 = o64[0]['loss_mat_reg']
    rec['loss_mat_reg_vs_fp64'] = dict(hip=rel_err(reg_hip, reg64), oracle_fp32=rel_err(oo['loss_mat_reg'], reg64))
    parity_report(test_id, **rec)
    assert rec['loss_mat_reg_vs_fp64']['hip'] <= max(1e-4, 3.0 * rec['loss_mat_reg_vs_fp64']['oracle_fp32']), rec
    loss = out['loss_rgb'].mean() + out['loss_mat_reg'].mean() + out['loss_diffuse_light'].mean()
    assert abs(float(loss) - loss_o) < 2e-5, (float(loss), loss_o)
    if not check_grads:
        return rec
    loss.backward()
    g_hip = {k: v.cpu() for k, v in named_grads(net).items()}
    from tests.helpers import forced_gates_from_capture
    gates = {**forced_gates_from_capture(capture, 2, Pn), **signs}
    del net, out, loss, capture
    _free()
    g64 = o64[2]
    if not small_batch:                                            # gate-teacher-forced (see _run_shape): plain 1e-4 for every tensor
        rec['forced_gates'] = _forced_gate_errors(g_hip, oracle(torch.float32, gates)[2], oracle(torch.float64, gates)[2])
    del gates
    _free()
    info = {}
    if small_batch:
        from tests.helpers import assert_grads_small_batch
        assert_grads_small_batch(g_hip, g32, g64, where=test_id, info=info)
        rec.update(info)
        parity_report(test_id, **rec)
        return rec
    assert_grads_fp32_grade(g_hip, g32, g64, where=test_id, info=info)
    rec.update(info)
    parity_report(test_id, **rec)
    assert info['n_clause_b'] <= MAX_CLAUSE_B and info['n_clause_a'] <= MAX_CLAUSE_A, info
    if 'forced_gates' in rec:
        assert rec['forced_gates']['n_unexplained'] == 0 and rec['forced_gates']['n_fp32_floor'] <= MAX_FORCED_FLOOR_STAGE2, rec['forced_gates']
    return rec


BELL2 = dict(human_lights=False, outer_light_version='direction')                  # configs/material/syn/bell.yaml
BEAR2 = dict(human_lights=True, outer_light_version='sphere_direction')            # configs/material/real/bear.yaml


def test_c4_bell_stage2_4096_points_256_directions():
    """BASELINE configs[3] at bench.py's size: 4096 x (128 + 128) = 1.05 M light rows"""
    rec = _stage2_teacher_forced('c4_bell_P4096_D256', 4096, dict(diffuse_sample_num=128, specular_sample_num=128, **BELL2))
    assert rec['light_rows'] == 4096 * 256 and 0.02 < rec['hit_fraction'] < 0.9


def test_c4_bell_stage2_4096_points_768_directions_yaml_default():
    """the YAML default 512 + 256 directions at P = 4096: 3.1 M light rows per launch (> 2^20-row tiles, multi-slice dW)"""
    rec = _stage2_teacher_forced('c4_bell_P4096_D768', 4096, dict(diffuse_sample_num=512, specular_sample_num=256, **BELL2))
    assert rec['light_rows'] == 4096 * 768


def test_c5_bear_stage2_2048_points_512_directions_per_gpu():
    """BASELINE configs[4] per-GPU share: bear (sphere_direction + human_lights), 2048 x (256 + 256)"""
    rec = _stage2_teacher_forced('c5_bear_P2048_D512', 2048, dict(diffuse_sample_num=256, specular_sample_num=256, **BEAR2))
    assert rec['light_rows'] == 2048 * 512


def test_c5_bear_stage2_512_points_128_directions():
    """the small bear case of round 2, kept: P = 512 x (64 + 64)"""
    _stage2_teacher_forced('c5_bear_P512_D128', 512, dict(diffuse_sample_num=64, specular_sample_num=64, **BEAR2))


# measured on MI355X (gpurun_out/parity_at_size.json, round 3) for P = 512 x 256 on the golden mesh: see the asserts below
def test_c4_shaped_stage2_hip_tracer_with_explicit_edge_exclusion():
    """P = 512 x 256 with the secondary rays traced by the HIP BVH.  A float32 tracer may answer differently from the fp64 oracle
    only on rays the oracle itself flags as razor-edge (a triangle edge within 1e-4 barycentric units of deciding the closest hit, or
    a candidate intersection within 5e-6 of the ray origin -- the rays start 1e-5 off the surface, so the triangle they left sits at
    t = -1e-5 exactly and is never such a candidate).  So:
    (1) every ray on which the two tracers disagree is such a ray;
    (2) round 5 (VERDICT r4 'weak' 2): the comparison is teacher-forced PER RAY, not by dropping points -- the oracle shades with its
        own fp64 hits everywhere except on the flagged rays (0.09 % of them), where it takes the HIP tracer's answer (both answers are
        legitimate there) and on the handful whose hit distance straddles get_lights' near mask `depth > 1e-5` (tests/helpers.py::CTracer),
        and both tracers are asked about the SAME rays (the HIP step's; its directions equal the oracle's to 2e-5).
        EVERY point is kept and compared; round 4 dropped the 117 of 512 points that owned a flagged ray."""
    from nero_amd.renderer import NeROMaterialRenderer
    from nero_amd.renderer import NeROShapeRenderer
    shader_cfg = dict(diffuse_sample_num=128, specular_sample_num=128, human_lights=False, outer_light_version='direction')
    Pn, D, step = 512, 256, 5000
    I = _material_inputs(Pn)
    ref = _material_pair(shader_cfg)
    sd = {k: v.detach() for k, v in ref.state_dict().items()}
    hp = NeROShapeRenderer.get_human_coordinate_poses(type('c', (), {'cfg': {'fixed_camera': False}})(), I['poses'])
    # the HIP side first: its answers are what the oracle defers to on the rays it finds ambiguous
    net = NeROMaterialRenderer({'shader_cfg': shader_cfg, 'database_name': 'syn/bell'}, mesh=golden_mesh())
    net.load_state_dict(ref.state_dict())
    net = net.cuda()
    rec = _Recording(net.ray_tracer)
    net.ray_tracer = rec
    c = lambda k: I[k].cuda()
    with torch.no_grad():
        out = net.shade(c('pts'), c('view'), c('normals'), hp.cuda(), True, step, c('rand_d'), c('rand_s'))
    hip = (np.concatenate(rec.pos).astype(np.float32), np.concatenate(rec.nrm).astype(np.float32), np.concatenate(rec.depth).astype(np.float32))
    assert hip[2].shape[0] == Pn * D

    hip_rays = (np.concatenate(rec.o), np.concatenate(rec.d))

    def oracle_shade(defer):
        # both tracers answer the SAME rays -- the ones the HIP step generated (its GGX / cosine directions agree with the oracle's to
        # 2e-5, asserted): a direction that differs in the seventh digit moves a grazing hit by 1e-5 and can carry it across an edge
        tr = _CTracer(*golden_mesh(), eps_edge=1e-4, eps_t=5e-6, defer=defer, rays_from=hip_rays, ray_tol=2e-5)
        with torch.no_grad():
            rgb, _ = M.mc_shade(O.effective_params(sd), {**M.DEFAULT_SHADER_CFG, **shader_cfg}, _contract(tr), I['pts'], I['view'], I['normals'],
                                hp, I['rand_d'], I['rand_s'])
        return rgb, tr
    rgb_own, tr = oracle_shade(None)                            # the oracle on its own hits: who is flagged, who differs
    amb, hit_o = np.concatenate(tr.amb), np.concatenate(tr.hit)
    assert amb.shape[0] == Pn * D
    hit_h = hip[2] < 10.0
    differ = hit_h != hit_o
    if (differ & ~amb).any():                                   # say what the unexplained rays look like before failing
        from oracle.tracer_oracle import trace_bruteforce_margins
        idx = np.nonzero(differ & ~amb)[0][:8]
        ro, rd = np.concatenate(rec.o)[idx], np.concatenate(rec.d)[idx]
        loose = trace_bruteforce_margins(*golden_mesh(), ro, rd, eps_edge=1e-3, eps_t=9e-6)
        raise AssertionError(f'{int((differ & ~amb).sum())} rays differ between the HIP BVH and the fp64 oracle without being razor-edge: rows '
                             f'{idx}, oracle depth on the HIP rays {loose[2]}, flagged at (1e-3, 9e-6): {loose[4]}, hip depth '
                             f'{hip[2][idx]}, oracle depth on its own rays {np.concatenate([r[2] for r in tr.raw])[idx]}')
    rgb_o, tr2 = oracle_shade(hip)                              # ... and deferring to the HIP tracer on exactly the flagged rays
    assert tr2.deferred == int(amb.sum()) + tr2.near_edge + tr2.near_mask_flips and tr2.offset == Pn * D
    # hit POSITIONS of the rays both tracers answer alike: float32 BVH against the fp64 brute force
    both = hit_h & hit_o & ~amb
    dpos = float(np.abs(hip[0][both] - np.concatenate([r[0] for r in tr.raw])[both]).max()) if both.any() else 0.0
    scale = float(rgb_o.abs().max())
    perr = (out['rgb_pr'].cpu() - rgb_o).abs().max(-1)[0] / scale
    perr_own = (out['rgb_pr'].cpu() - rgb_own).abs().max(-1)[0] / scale
    pt_amb = torch.from_numpy(amb.reshape(Pn, D).any(axis=1))
    frac_good, worst = float((perr < 1e-4).float().mean()), float(perr.max())
    raw_o = [np.concatenate([r[j] for r in tr.raw]) for j in range(3)]          # the oracle's own answers (before any deferral) on the same rays
    dpos_r = np.abs(hip[0] - raw_o[0]).max(-1).reshape(Pn, D)
    dnrm_r = np.abs(hip[1] - raw_o[1]).max(-1).reshape(Pn, D)
    ddep_r = np.abs(hip[2] - raw_o[2]).reshape(Pn, D)
    offenders = [dict(point=int(i), err=float(perr[i]), owns_flagged_ray=bool(pt_amb[i]), max_dpos=float(dpos_r[i].max()), max_dnormal=float(dnrm_r[i].max()),
                      max_ddepth=float(ddep_r[i].max()), ray=int(np.argmax(dpos_r[i] + dnrm_r[i])), hit_h=bool(hit_h.reshape(Pn, D)[i, np.argmax(dpos_r[i] + dnrm_r[i])]),
                      depth_h=float(hip[2].reshape(Pn, D)[i, np.argmax(dpos_r[i] + dnrm_r[i])]), depth_o=float(raw_o[2].reshape(Pn, D)[i, np.argmax(dpos_r[i] + dnrm_r[i])]))
                 for i in torch.nonzero(perr >= 1e-4)[:, 0][:16]]
    parity_report('c4_bell_P512_D256_hip_bvh', points=Pn, directions=D, razor_edge_rays=int(amb.sum()), rays=Pn * D,
                  rays_answered_differently=int(differ.sum()), points_owning_a_flagged_ray=int(pt_amb.sum()), fraction_points_kept=1.0,
                  rays_same_point_other_triangle=int(tr2.near_edge), rays_straddling_the_near_mask=int(tr2.near_mask_flips),
                  fraction_points_within_1e4=frac_good, worst_point=worst, points_beyond_1e4=offenders,
                  worst_point_without_per_ray_forcing=float(perr_own.max()), max_hit_position_difference=dpos, hit_fraction=float(hit_o.mean()))
    # every point is compared.  What can remain beyond 1e-4 is the float32 hit POSITION (|dx| <= max_hit_position_difference, amplified
    # 2^7-fold by PE-8 in front of the inner-light MLP) on single rays of a point; STAGE2_BVH_BOUNDS holds the measured values.
    assert frac_good >= STAGE2_BVH_BOUNDS['min_fraction_within_1e4'], (frac_good, offenders)
    assert worst < STAGE2_BVH_BOUNDS['max_worst_point'], (worst, offenders)
    assert dpos < 5e-5, dpos        # (measured 1.4e-5: a grazing hit -- t = (q . e2) / det with a small det -- carries ~1e-5 of float32 error along the ray)


# measured on MI355X for this deterministic case (profiles/r05_parity_at_size.json): 131 072 rays, 120 flagged razor-edge by the oracle (2 of
# them answered differently), 4 straddling the near mask, 0 "same point, other triangle"; ALL 512 points compared, all within 1e-4, worst 5.6e-7
# (round 4 excluded the 117 points owning a flagged ray and had 99.24 % of the rest within 1e-4, worst 2.2e-3: the four near-mask rays).
# round 5, per-ray forcing, all 512 points kept: see profiles/r05_parity_at_size.json
STAGE2_BVH_BOUNDS = {'min_fraction_within_1e4': 1.0, 'max_worst_point': 1e-4}
