"""GPU tier: the Stage-I render step on the HIP library vs the CPU oracle (oracle/nero_oracle.py), which is itself pinned to
the unmodified reference by tests/test_oracle_golden.py.  Tolerance: 1e-4 rel fp32 on outputs (BASELINE.json north_star),
bit-exact on sample indices under teacher forcing."""
import numpy as np
import pytest
import torch

from oracle import nero_oracle as O
from tests.helpers import T, assert_grads_fp32_grade, build_case_model, load_golden, named_grads

pytestmark = pytest.mark.gpu

CASES = ['bell_s25000', 'bell_s5000_sharp', 'bell_c1', 'bear_s25000', 'bell_sphdir', 'bell_noclip_l1', 'bell_l2', 'bell_smoothl1']
# non-YAML network-shape keys (round 5; oracle/gen_golden_r5.py): sdf_n_layers 6 / sdf_freq 4 / light_pos_freq 6 and 9 / 5 / 10 --
# the Python-sequenced chains (the C step driver keeps the YAML shapes, nero_amd.stage1.supported)
SHAPE_KEY_CASES = ['bell_shape_keys', 'bell_deep_sdf']


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _oracle_grads(meta, z, dtype, occ, keys=None, gates=None):
    """parameter gradients of the oracle evaluated in `dtype` on the golden case (teacher-forced on its z_vals); gates: ReLU decisions
    to force (oracle.nero_oracle.forced_relu_gates)"""
    if gates is not None:
        with O.forced_relu_gates(gates) as fg:
            res = _oracle_grads(meta, z, dtype, occ, keys)
            assert fg.used == set(gates), sorted(set(gates) - fg.used)[:5]
        return res
    ref = build_case_model(meta).to(dtype)
    sd = {k: v for k, v in ref.named_parameters()}
    sd.update({k: v for k, v in ref.named_buffers()})
    P = O.effective_params(sd)
    f = lambda k: T(z, k).to(dtype)
    if occ:
        cfg = {**O.DEFAULT_CFG, **meta['cfg']}
        oo = O.render_core(P, cfg, f('o'), f('d'), f('z_vals'), f('human_poses'), meta['anneal'], meta['step'], keys)
        O.training_loss(cfg, oo, f('gt'), meta['step']).backward()
    else:
        cfg = {**O.DEFAULT_CFG, **meta['cfg'], 'apply_occ_loss': False}
        oo = O.render_core(P, cfg, f('o'), f('d'), f('z_vals'), f('human_poses'), meta['anneal'], meta['step'])
        (O.rgb_loss(cfg, oo['ray_rgb'], f('gt')).mean() + (oo['gradient_error'] * 0.1).mean()).backward()
    return named_grads(ref)


def _check_grads(name, net, ref, meta, z, occ, keys, capture, n_in):
    """HIP parameter gradients against the fp64 oracle: the ungated criterion of tests/helpers.py first (1e-4, or fp32 torch equally far,
    or 1e-4 of the MLP's gradient scale).  A ReLU unit within rounding of zero may fall on the other side in the HIP forward than in BOTH
    torch evaluations -- with 48 rays one flipped unit moves a bias gradient by 1e-3 of its maximum, and which evaluation flips depends on
    the last bit of a 256-term sum (the one-accumulator chain kernels of round 5 round differently from rounds 1-4: bell_smoothl1 then
    failed on exactly one outer-light bias).  Such a case is decided by ARITHMETIC: the oracle in fp64 and fp32 is handed the HIP
    forward's own gate decisions (its saved sign masks) and the plain 1e-4 must hold for every tensor -- no escape clause beyond the two
    NeRF++ density-head tensors whose fp32 floor tests/test_parity_at_size.py documents."""
    from tests.helpers import forced_gates_from_capture
    from tests.test_parity_at_size import STAGE1_FLOOR_TENSORS, _forced_gate_errors
    g_hip = {k: v.cpu() for k, v in named_grads(net).items()}
    g64 = _oracle_grads(meta, z, torch.float64, occ, keys)
    try:
        assert_grads_fp32_grade(g_hip, named_grads(ref), g64, where=name)
        return
    except AssertionError as e:
        ungated = str(e)
    sc = {**{'light_pos_freq': 8}, **meta['cfg'].get('shader_config', {})}
    pd = 3 + 6 * int(sc['light_pos_freq'])
    r8 = lambda k: (k + 7) // 8 * 8
    gates = forced_gates_from_capture(capture, 1, n_in, k_inner_light=r8(pd + 72), k_inner_weight=r8(pd + 39))
    fe = _forced_gate_errors(g_hip, _oracle_grads(meta, z, torch.float32, occ, keys, gates), _oracle_grads(meta, z, torch.float64, occ, keys, gates))
    assert fe['n_unexplained'] == 0 and set(fe['fp32_floor']) <= STAGE1_FLOOR_TENSORS, (ungated[:400], fe)


def _oracle_P(net):
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    return O.effective_params(sd)


@pytest.mark.parametrize('name', CASES)
def test_sampler_stagewise_teacher_forced(name):
    """each up-sampling round is fed the ORACLE's (z, sdf) of that round: searchsorted indices and the merge permutation
    must be identical; weights / new z within fp32 tolerance."""
    from nero_amd import _lib as L
    import ctypes as C
    z, meta = load_golden(name)
    net = build_case_model(meta).cuda()
    cfg = {**O.DEFAULT_CFG, **meta['cfg']}
    trace = []
    O.sample_ray(_oracle_P(net), cfg, T(z, 'o'), T(z, 'd'), T(z, 'near'), T(z, 'far'), T(z, 'rand1'), T(z, 'rand_bg'), trace)
    o, d = T(z, 'o', 'cuda'), T(z, 'd', 'cuda')
    R = o.shape[0]
    st = L.stream_ptr()
    P = C.c_void_p
    for t in trace:
        n, m = t['z'].shape[1], t['z_new'].shape[1]
        zc, sc = t['z'].cuda().contiguous(), t['sdf'].cuda().contiguous()
        z_new = torch.empty(R, m, device='cuda'); w = torch.empty(R, n - 1, device='cuda')
        inds = torch.empty(R, m, dtype=torch.int32, device='cuda')
        L.check(L.lib.nero_upsample(P(o.data_ptr()), P(d.data_ptr()), P(zc.data_ptr()), n, P(sc.data_ptr()), n, n, P(None),
                                    C.c_float(t['inv_s']), m, R, P(z_new.data_ptr()), P(w.data_ptr()), P(inds.data_ptr()), st))
        assert rel(w, t['weights']) < 2e-5
        # indices from the kernel's own weights: identical except where a 1-ulp weight difference crosses a cdf edge
        assert (inds.cpu() != t['inds'].int()).float().mean() < 2e-3
        # exact contract: sample_pdf on the oracle's weights
        wt = t['weights'].cuda().contiguous()
        out = torch.empty(R, m, device='cuda'); inds2 = torch.empty(R, m, dtype=torch.int32, device='cuda')
        L.check(L.lib.nero_sample_pdf(P(zc.data_ptr()), n, P(wt.data_ptr()), n - 1, n, m, R, P(out.data_ptr()), P(inds2.data_ptr()), st))
        assert torch.equal(inds2.cpu(), t['inds'].int())
        assert (out.cpu() - t['z_new']).abs().max() < 2e-6
        # exact contract: merge permutation on the oracle's z_new
        zt = torch.empty(R, n + m, device='cuda'); zt[:, :n] = zc
        index = torch.empty(R, n + m, dtype=torch.int32, device='cuda')
        zn = t['z_new'].cuda().contiguous()
        L.check(L.lib.nero_merge_sorted(P(zt.data_ptr()), n + m, n, P(None), 0, P(zn.data_ptr()), m, P(None), 0, R, P(index.data_ptr()), st))
        assert torch.equal(index.cpu(), t['index'].int())
        assert torch.equal(zt.cpu(), t['z_out'])


@pytest.mark.parametrize('name', CASES + SHAPE_KEY_CASES)
def test_sampler_end_to_end(name):
    z, meta = load_golden(name)
    net = build_case_model(meta).cuda()
    cfg = {**O.DEFAULT_CFG, **meta['cfg']}
    zo = O.sample_ray(_oracle_P(net), cfg, T(z, 'o'), T(z, 'd'), T(z, 'near'), T(z, 'far'), T(z, 'rand1'), T(z, 'rand_bg'))
    zg = net.sample_ray(T(z, 'o', 'cuda'), T(z, 'd', 'cuda'), T(z, 'near', 'cuda'), T(z, 'far', 'cuda'), 1.0,
                        T(z, 'rand1', 'cuda'), T(z, 'rand_bg', 'cuda')).cpu()
    nb = cfg['n_bg_samples']
    dz = (zg[:, :-nb] - zo[:, :-nb]).abs()
    assert dz.max() < 2e-3 and (dz < 1e-5).float().mean() > 0.95, (dz.max(), (dz < 1e-5).float().mean())
    assert (zg[:, -nb:] / zo[:, -nb:] - 1).abs().max() < 1e-6


@pytest.mark.parametrize('name', CASES + SHAPE_KEY_CASES)
def test_render_core_outputs_and_grads(name):
    """teacher-forced on the golden z_vals: ray_rgb / gradient_error / occ_prob and every parameter gradient vs the oracle"""
    z, meta = load_golden(name)
    net = build_case_model(meta).cuda()
    ref = build_case_model(meta)                       # CPU copy for the oracle
    sd = {k: v for k, v in ref.named_parameters()}
    sd.update({k: v for k, v in ref.named_buffers()})
    P = O.effective_params(sd)
    cfg = {**O.DEFAULT_CFG, **meta['cfg'], 'apply_occ_loss': False}
    oo = O.render_core(P, cfg, T(z, 'o'), T(z, 'd'), T(z, 'z_vals'), T(z, 'human_poses'), meta['anneal'], meta['step'])
    loss_o = O.rgb_loss(cfg, oo['ray_rgb'], T(z, 'gt')).mean() + (oo['gradient_error'] * 0.1).mean()
    loss_o.backward()
    from nero_amd import chain as CH
    CH.MASK_CAPTURE = []
    try:
        out = net.render(T(z, 'o', 'cuda'), T(z, 'd', 'cuda'), T(z, 'near', 'cuda'), T(z, 'far', 'cuda'), T(z, 'human_poses', 'cuda'),
                         -1, meta['anneal'], is_train=True, step=meta['step'], z_vals=T(z, 'z_vals', 'cuda'))
        capture = CH.MASK_CAPTURE
    finally:
        CH.MASK_CAPTURE = None
    assert out['gradient_error'].shape == oo['gradient_error'].shape
    assert rel(out['ray_rgb'], oo['ray_rgb']) < 1e-4
    assert rel(out['gradient_error'], oo['gradient_error']) < 1e-4
    loss = net.compute_rgb_loss(out['ray_rgb'], T(z, 'gt', 'cuda')).mean() + (out['gradient_error'] * 0.1).mean()
    assert abs(float(loss) - float(loss_o)) < 1e-5
    loss.backward()
    # gradients: within 1e-4 of the fp64 oracle unless torch-fp32 itself is equally off (tests/helpers.py); else under forced gates
    _check_grads(name, net, ref, meta, z, False, None, capture, out['_state']['n_in'])


@pytest.mark.parametrize('act,variance', [('linear', 2.0), ('square', 0.45)])
def test_std_act_linear_and_square_render_and_variance_gradient(act, variance):
    """std_act != 'exp' (network/field.py:193-196; no shipped YAML uses it): sampler, render_core, occlusion-loss march and the gradient
    of the variance parameter against the oracle (the kernels are handed v' = log(inv_s) / 10, nero_amd.fields.SingleVarianceNetwork)"""
    z, meta = load_golden('bell_s25000')
    meta = {**meta, 'cfg': {**meta['cfg'], 'std_act': act}, 'variance': variance}
    net = build_case_model(meta).cuda()
    ref = build_case_model(meta)
    assert abs(float(net.deviation_network.inv_s()) - 20.0) < 0.5
    sd = {k: v for k, v in ref.named_parameters()}
    sd.update({k: v for k, v in ref.named_buffers()})
    P = O.effective_params(sd)
    cfg = {**O.DEFAULT_CFG, **meta['cfg']}
    with torch.no_grad():
        zo = O.sample_ray(P, cfg, T(z, 'o'), T(z, 'd'), T(z, 'near'), T(z, 'far'), T(z, 'rand1'), T(z, 'rand_bg'))
        zg = net.sample_ray(T(z, 'o', 'cuda'), T(z, 'd', 'cuda'), T(z, 'near', 'cuda'), T(z, 'far', 'cuda'), 1.0, T(z, 'rand1', 'cuda'),
                            T(z, 'rand_bg', 'cuda')).cpu()
    assert float(((zg - zo).abs() < 1e-5).float().mean()) > 0.97          # (the sampler is ill-conditioned end to end: a statistic)
    g = torch.Generator().manual_seed(3)
    keys = torch.rand(zo.numel(), generator=g)
    oo = O.render_core(P, cfg, T(z, 'o'), T(z, 'd'), zo, T(z, 'human_poses'), meta['anneal'], meta['step'], keys)
    loss_o = O.training_loss(cfg, oo, T(z, 'gt'), meta['step'])
    loss_o.backward()
    out = net.render(T(z, 'o', 'cuda'), T(z, 'd', 'cuda'), T(z, 'near', 'cuda'), T(z, 'far', 'cuda'), T(z, 'human_poses', 'cuda'),
                     -1, meta['anneal'], is_train=True, step=meta['step'], z_vals=zo.cuda(), occ_keys=keys)
    assert rel(out['ray_rgb'], oo['ray_rgb']) < 1e-4 and rel(out['gradient_error'], oo['gradient_error']) < 1e-4
    assert abs(float(out['std']) - float(oo['std'])) < 1e-6 * max(1.0, float(oo['std']))
    assert abs(float(out['loss_occ']) - float(oo['loss_occ'])) < 1e-5
    from nero_amd.train import shape_training_loss
    loss = shape_training_loss(net, out, T(z, 'gt', 'cuda'), meta['step'])
    assert abs(float(loss) - float(loss_o)) < 2e-5
    loss.backward()
    gv, gvo = float(net.deviation_network.variance.grad), float(ref.deviation_network.variance.grad)
    assert abs(gvo) > 1e-9 and abs(gv - gvo) < 2e-4 * abs(gvo), (gv, gvo)


@pytest.mark.parametrize('name', ['bell_s25000', 'bell_occcap', 'bell_s500', 'bear_s25000', 'bell_sphdir', 'bell_noclip_l1', 'bell_l2', 'bell_smoothl1']
                         + SHAPE_KEY_CASES)
def test_full_training_loss_with_occ_and_init_reg(name):
    """trainer loss incl. the occlusion loss (step >= 20000, with and without the random cap) and the InitSDFRegLoss inputs
    (step < 1000), teacher-forced on the golden z_vals; loss, loss_occ and gradients vs the oracle"""
    from nero_amd.train import shape_training_loss
    z, meta = load_golden(name)
    net = build_case_model(meta).cuda()
    ref = build_case_model(meta)
    sd = {k: v for k, v in ref.named_parameters()}
    sd.update({k: v for k, v in ref.named_buffers()})
    P = O.effective_params(sd)
    cfg = {**O.DEFAULT_CFG, **meta['cfg']}
    keys = T(z, 'occ_keys') if 'occ_keys' in z.files else None
    oo = O.render_core(P, cfg, T(z, 'o'), T(z, 'd'), T(z, 'z_vals'), T(z, 'human_poses'), meta['anneal'], meta['step'], keys)
    loss_o = O.training_loss(cfg, oo, T(z, 'gt'), meta['step'])
    loss_o.backward()
    from nero_amd import chain as CH
    CH.MASK_CAPTURE = []
    try:
        out = net.render(T(z, 'o', 'cuda'), T(z, 'd', 'cuda'), T(z, 'near', 'cuda'), T(z, 'far', 'cuda'), T(z, 'human_poses', 'cuda'),
                         -1, meta['anneal'], is_train=True, step=meta['step'], z_vals=T(z, 'z_vals', 'cuda'), occ_keys=keys)
        capture = CH.MASK_CAPTURE
    finally:
        CH.MASK_CAPTURE = None
    if meta['step'] >= 20000:
        assert out['_occ_count'] == oo['occ_count'] > 0
        assert abs(float(out['loss_occ']) - float(oo['loss_occ'])) < 1e-5
    if meta['step'] < 1000:
        assert out['sdf_vals'].shape == oo['sdf_vals'].shape
        assert rel(out['sdf_vals'], oo['sdf_vals']) < 2e-5
    loss = shape_training_loss(net, out, T(z, 'gt', 'cuda'), meta['step'])
    assert abs(float(loss) - float(loss_o)) < 2e-5
    assert abs(float(loss) - float(z['loss'])) < 5e-5            # and the unmodified reference's own loss value
    loss.backward()
    _check_grads(name, net, ref, meta, z, True, keys, capture, out['_state']['n_in'])


def test_non_yaml_network_shape_keys_on_the_fused_trainer():
    """sdf_n_layers / sdf_freq / light_pos_freq away from the YAML values: the fused training step runs (Python-sequenced chains, fused
    weight-norm + Adam kernels over the 7-layer SDF network), its first loss equals the drop-in renderer's on the same batch, and
    training reduces it; sdf_d_out != 257 raises like the reference's first forward would"""
    from nero_amd.renderer import NeROShapeRenderer
    from nero_amd.train import ShapeTrainStep
    cfg = dict(n_samples=16, n_importance=16, n_bg_samples=8, up_sample_steps=4, sdf_n_layers=6, sdf_freq=4, sdf_activation='sigmoid',
               shader_config={'light_pos_freq': 6}, freeze_inv_s_step=15000, apply_occ_loss=True, occ_loss_step=20000)
    ts = ShapeTrainStep(cfg, rays_per_rank=256, pool_rays=2048, device='cuda:0', variance=0.4, prime_fraction=0.0)
    assert ts.drv is None and ts.net.sdf_network.n_lin == 7 and ts.net.sdf_network.lin0.weight_v.shape[1] == 27
    losses = [float(ts.step(25000 + i)['loss']) for i in range(12)]
    assert all(np.isfinite(losses)) and min(losses[-3:]) < losses[0], losses
    # a from-scratch run starts below step 1000, where render() adds the inputs of InitSDFRegLoss (network/renderer.py:591-594) through
    # SDFValue -- an autograd node over the SDF's OWN 2 * (sdf_n_layers + 1) tensors, not the YAML depth's 18: forward + backward at both
    # a shallower and a deeper network than the YAML's (the slice was hard-wired to 18)
    from nero_amd.train import shape_training_loss
    for n_layers in (6, 9):
        torch.manual_seed(3)
        net = NeROShapeRenderer({**cfg, 'sdf_n_layers': n_layers}, training=False).cuda()
        g = torch.Generator().manual_seed(5)
        d = torch.nn.functional.normalize(torch.randn(128, 3, generator=g), dim=-1)
        o = -3.0 * d + 0.3 * torch.randn(128, 3, generator=g)
        o, d = o.cuda(), d.cuda()
        near, far = net.near_far_from_sphere(o, d)
        out = net.render(o, d, near, far, None, -1, 0.0, is_train=True, step=300)
        assert out['sdf_vals'].shape[0] == out['sdf_pts'].shape[0] > 0
        loss = shape_training_loss(net, out, torch.rand(128, 3, device='cuda'), 300) + out['sdf_vals'].abs().mean()
        loss.backward()
        gs = {k: p.grad for k, p in net.sdf_network.named_parameters()}
        assert len(gs) == 3 * (n_layers + 1) and all(v is not None and bool(torch.isfinite(v).all()) for v in gs.values()), n_layers
        assert float(gs[f'lin{n_layers}.weight_v'].abs().sum()) > 0
    with pytest.raises(NotImplementedError):
        NeROShapeRenderer({'sdf_d_out': 129}, training=False)
    with pytest.raises(NotImplementedError):
        NeROShapeRenderer({'sdf_freq': 7}, training=False)


def test_trainer_entry_point_with_database_object():
    """forward({'step': ...}) over an HBM-resident ray pool built from a database-like object: output keys of the reference's
    train_step and a finite backward."""
    from nero_amd.renderer import NeROShapeRenderer
    from nero_amd.synthetic import look_at_pose

    class FakeDB:
        def __init__(self):
            rg = np.random.default_rng(0)
            self.imgs = rg.uniform(0, 1, (3, 32, 32, 3)).astype(np.float32)
            self.K = np.array([[40., 0, 16], [0, 40., 16], [0, 0, 1]], np.float32)
            self.poses = [look_at_pose(np.array(c, dtype=np.float64)) for c in ([3, 0, 0.5], [0, 3, 1.0], [-2, -2, 1.5])]
        def get_img_ids(self): return [0, 1, 2]
        def get_image(self, i): return self.imgs[i]
        def get_K(self, i): return self.K
        def get_pose(self, i): return self.poses[i]

    torch.manual_seed(1)
    net = NeROShapeRenderer({'train_ray_num': 256, 'n_samples': 16, 'n_importance': 16, 'n_bg_samples': 8,
                             'shader_config': {'human_light': True}}, training=False).cuda()
    net._init_dataset(FakeDB())
    out = net({'step': 25000})
    for k in ('ray_rgb', 'gradient_error', 'std', 'loss_occ', 'loss_rgb'):
        assert k in out, k
    assert out['ray_rgb'].shape == (256, 3) and out['loss_rgb'].shape == (256,)
    loss = out['loss_rgb'].mean() + (out['gradient_error'] * 0.1).mean() + out['loss_occ'].mean()
    loss.backward()
    assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)


@pytest.mark.parametrize('name', ['bell_val', 'bear_val', 'bell_shape_keys', 'bell_deep_sdf'])
def test_validation_path(name):
    """is_train=False: ray_rgb + compute_validation_info outputs (depth, normal, shader intermediates, marched occlusion)
    vs the oracle on the same z_vals, and render_image / nvs plumbing.  bell_shape_keys / bell_deep_sdf: the non-YAML network shapes
    (light_pos_freq != 8 re-encodes the light networks' position columns; ADVICE r5: validation_info skipped that step)"""
    from tests.test_oracle_golden import VAL_KEYS
    z, meta = load_golden(name)
    net = build_case_model(meta).cuda()
    P = _oracle_P(net)
    cfg = {**O.DEFAULT_CFG, **meta['cfg']}
    with torch.no_grad():
        zv = O.sample_ray(P, cfg, T(z, 'o'), T(z, 'd'), T(z, 'near'), T(z, 'far'))
        oo = O.render_core(P, cfg, T(z, 'o'), T(z, 'd'), zv, T(z, 'human_poses'), 0.0, meta['step'])
        val = O.validation_info(P, cfg, T(z, 'o'), T(z, 'd'), zv, oo['weights'], T(z, 'human_poses'))
        out = net.render(T(z, 'o', 'cuda'), T(z, 'd', 'cuda'), T(z, 'near', 'cuda'), T(z, 'far', 'cuda'), T(z, 'human_poses', 'cuda'),
                         0, 0, is_train=False, step=meta['step'], z_vals=zv.cuda())
    assert rel(out['ray_rgb'], oo['ray_rgb']) < 1e-4
    keys = VAL_KEYS + (['human_light'] if name.startswith('bear') else [])
    for k in keys:
        assert out[k].shape == val[k].shape, k
        assert rel(out[k], val[k]) < (2e-3 if k == 'occ_prob_gt' else 2e-4), (k, rel(out[k], val[k]))


def test_nvs_renders_an_image():
    from nero_amd.renderer import NeROShapeRenderer
    from nero_amd.synthetic import look_at_pose
    torch.manual_seed(2)
    net = NeROShapeRenderer({'n_samples': 16, 'n_importance': 16, 'n_bg_samples': 8, 'test_ray_num': 500}, training=False).cuda()
    K = np.array([[30., 0, 12], [0, 30., 12], [0, 0, 1]], np.float32)
    img = net.nvs(look_at_pose(np.array([0., -3., 1.])), K, 24, 24)
    assert img.shape == (24, 24, 3) and np.isfinite(img).all() and img.std() > 1e-3


def test_nvs_through_the_c_driver_equals_the_python_sequenced_render(monkeypatch):
    """render_image(extras=False) issues each chunk through nero_stage1_sample / _render_fwd; same kernels, same order: same bits"""
    from nero_amd.renderer import NeROShapeRenderer
    from nero_amd.synthetic import look_at_pose, perturb_state
    torch.manual_seed(5)
    net = NeROShapeRenderer({'apply_occ_loss': True, 'occ_loss_step': 20000}, training=False)
    perturb_state(net, 0.4)
    net = net.cuda()
    K = np.array([[90., 0, 32], [0, 90., 32], [0, 0, 1]], np.float32)
    pose = look_at_pose(np.array([0.5, -2.8, 0.6], np.float32))
    a = net.render_image(pose, K, 64, 64, chunk=1000)['ray_rgb']
    assert getattr(net, '_infer_drv', None) is not None
    monkeypatch.setattr(NeROShapeRenderer, '_inference_driver', lambda self, kern: None)
    b = net.render_image(pose, K, 64, 64, chunk=1000)['ray_rgb']
    assert a.shape == (4096, 3) and bool(torch.isfinite(a).all()) and float(a.std()) > 1e-3
    assert torch.equal(a, b), float((a - b).abs().max())


def test_extract_fields_grid():
    """SDF grid for mesh extraction (field.py:1090-1108) vs the oracle's SDF on the same grid points"""
    z, meta = load_golden('bell_s25000')
    net = build_case_model(meta).cuda()
    u = net.extract_fields(resolution=24, chunk=5000)
    P = _oracle_P(net)
    ax = torch.linspace(-1, 1, 24)
    g = torch.stack(torch.meshgrid(ax, ax, ax, indexing='ij'), -1).reshape(-1, 3)
    with torch.no_grad():
        ref = O.sdf_network(P, g)[:, 0]
    ref = torch.where(torch.norm(g, dim=-1) >= 1.0, torch.ones_like(ref), ref).reshape(24, 24, 24)
    assert np.abs(u - ref.numpy()).max() < 2e-5
    assert (u < 0).sum() > 10


def test_training_reduces_the_loss():
    """functional check of the whole loop (render -> loss -> backward -> all-reduce(no-op) -> Adam): the rgb loss on a fixed small
    batch drops when the same batch is fitted for a few dozen steps"""
    from nero_amd.train import ShapeTrainStep
    ts = ShapeTrainStep({'n_samples': 16, 'n_importance': 16, 'n_bg_samples': 8, 'freeze_inv_s_step': 15000}, rays_per_rank=256,
                        pool_rays=256, device='cuda:0', variance=0.3)
    ts.pool['gt'] = (0.5 + 0.5 * torch.sin(ts.pool['d'] * 3.0)).contiguous()       # a smooth, learnable target
    first = last = None
    for i in range(40):
        ts.cursor = 0
        info = ts.step(6000 + i)
        v = float(info['loss'])
        first = v if first is None else first
        last = v
        assert np.isfinite(v)
    assert last < 0.8 * first, (first, last)


def test_train_eval_train_sequence_with_several_training_images():
    """the Trainer's normal loop: train -> validate -> train (the eval path used to overwrite the per-image human
    frames with the single test view's, so the next train step indexed out of bounds for imn > 1).  Also the drop-in surface of
    test_step (network/renderer.py:274-317): test split view, down-sampling, loss_rgb / gt_rgb / gt_depth / gt_mask."""
    from nero_amd.renderer import NeROShapeRenderer
    from nero_amd.synthetic import look_at_pose

    class FakeDB:
        def __init__(self):
            rg = np.random.default_rng(0)
            self.imgs = rg.uniform(0, 1, (4, 32, 40, 3)).astype(np.float32)
            self.K = np.array([[40., 0, 20], [0, 40., 16], [0, 0, 1]], np.float32)
            self.poses = [look_at_pose(np.array(c, dtype=np.float64)) for c in ([3, 0, 0.5], [0, 3, 1.0], [-2, -2, 1.5], [2, -2, 0.7])]
        def get_img_ids(self): return [0, 1, 2, 3]
        def get_image(self, i): return self.imgs[i]
        def get_K(self, i): return self.K
        def get_pose(self, i): return self.poses[i]
        def get_depth(self, i): return np.full((32, 40), 2.0 + i, np.float32), np.ones((32, 40), bool)

    torch.manual_seed(1)
    net = NeROShapeRenderer({'train_ray_num': 256, 'test_ray_num': 100, 'n_samples': 16, 'n_importance': 16, 'n_bg_samples': 8,
                             'shader_config': {'human_light': True}, 'downsample_ratio': 0.5}, training=False).cuda()
    net._init_dataset(FakeDB())
    hp_before = net._train_human_poses.clone()
    o1 = net({'step': 25000})
    ev = net({'eval': True, 'index': 0, 'step': 25000})
    assert torch.equal(net._train_human_poses, hp_before) and net._train_human_poses.shape[0] == 4
    for k in ('ray_rgb', 'gt_rgb', 'loss_rgb', 'gt_depth', 'gt_mask', 'depth', 'normal', 'metallic', 'roughness', 'occ_prob_gt', 'human_light'):
        assert k in ev, k
    assert ev['ray_rgb'].shape == (16, 20, 3) and ev['gt_rgb'].shape == (16, 20, 3) and ev['loss_rgb'].shape == (16 * 20,)
    assert ev['gt_depth'].shape == (16, 20, 1) and float(ev['gt_depth'].mean()) == 2.0          # view test_ids[0] = image 0, nearest
    o2 = net({'step': 25001})
    loss = o2['loss_rgb'].mean() + (o2['gradient_error'] * 0.1).mean() + o2['loss_occ'].mean()
    loss.backward()
    assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
    # inference cache: a second validation render re-uses the packed operand images, a parameter update invalidates them
    k1 = net._kern_cache[1][2]
    net({'eval': True, 'index': 0, 'step': 25000})
    assert net._kern_cache[1][2] is k1
    with torch.no_grad():
        net.sdf_network.lin8.bias.add_(0.01)
    ev2 = net({'eval': True, 'index': 0, 'step': 25000})
    assert net._kern_cache[1][2] is not k1
    assert float((ev2['depth'] - ev['depth']).abs().max()) > 1e-4
