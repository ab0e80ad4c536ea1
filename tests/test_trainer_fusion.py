"""GPU tier: the fused trainer loop (nero_wn_forward_batch + batched operand packing + nero_wn_adam_batch)
against the torch path it replaces (torch._weight_norm with autograd + torch.optim.Adam, i.e. what the reference's Trainer does:
nn.utils.weight_norm in network/field.py:118-119, 323-331 and train/trainer.py:105-170)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = {'n_samples': 16, 'n_importance': 16, 'n_bg_samples': 8, 'freeze_inv_s_step': 15000, 'perturb': 0.0, 'apply_occ_loss': True,
       'occ_loss_step': 20000}


def test_weight_norm_kernels_vs_torch():
    """forward: W = g v / ||v||_row; backward + Adam: one step against torch autograd through torch._weight_norm + torch.optim.Adam"""
    from nero_amd import _lib as L
    g0 = torch.Generator().manual_seed(0)
    shapes = [(256, 259), (217, 256), (3, 256), (257, 256)]
    vs = [torch.randn(s, generator=g0).cuda() for s in shapes]
    gs = [(torch.rand(s[0], 1, generator=g0) + 0.5).cuda() for s in shapes]
    dWs = [torch.randn(s, generator=g0).cuda() * 1e-3 for s in shapes]
    bias, dbias = torch.randn(300, generator=g0).cuda(), torch.randn(300, generator=g0).cuda() * 1e-2
    # torch reference
    ref_p = [t.clone().requires_grad_(True) for t in vs + gs] + [bias.clone().requires_grad_(True)]
    opt = torch.optim.Adam(ref_p, lr=3e-4)
    n = len(shapes)
    for it in range(2):
        opt.zero_grad()
        loss = sum((torch._weight_norm(ref_p[i], ref_p[n + i], 0) * dWs[i]).sum() for i in range(n)) + (ref_p[-1] * dbias).sum()
        loss.backward()
        opt.step()
    # fused kernels
    v2, g2, b2 = [t.clone() for t in vs], [t.clone() for t in gs], bias.clone()
    weff = [torch.empty_like(t) for t in vs]
    inv = [torch.empty(s[0], device='cuda') for s in shapes]
    z = lambda t: torch.zeros_like(t)
    mv, vv, mg, vg = [z(t) for t in vs], [z(t) for t in vs], [z(t) for t in gs], [z(t) for t in gs]
    mb, vb = z(bias), z(bias)
    jobs = (L.WnJob * n)()
    for i, j in enumerate(jobs):
        j.v = j.v_rw = v2[i].data_ptr(); j.g = j.g_rw = g2[i].data_ptr(); j.w_eff = weff[i].data_ptr(); j.inv_norm = inv[i].data_ptr()
        j.dW = dWs[i].data_ptr(); j.m_v, j.v_v, j.m_g, j.v_g = mv[i].data_ptr(), vv[i].data_ptr(), mg[i].data_ptr(), vg[i].data_ptr()
        j.rows, j.cols = shapes[i]
    pj = (L.AdamJob * 1)()
    pj[0].p, pj[0].grad, pj[0].m, pj[0].v, pj[0].n = b2.data_ptr(), dbias.data_ptr(), mb.data_ptr(), vb.data_ptr(), 300
    for it in range(2):
        L.check(L.lib.nero_wn_forward_batch(jobs, n, L.stream_ptr()))
        if it == 0:
            for i in range(n):
                want = torch._weight_norm(vs[i], gs[i], 0)
                assert float((weff[i] - want).abs().max() / want.abs().max()) < 1e-6
        L.check(L.lib.nero_wn_adam_batch(jobs, n, pj, 1, C.c_float(3e-4), C.c_float(0.9), C.c_float(0.999), C.c_float(1e-8), it + 1, L.stream_ptr()))
    torch.cuda.synchronize()
    for got, want, p0 in zip(v2 + g2 + [b2], ref_p, vs + gs + [bias]):
        upd = (want.detach() - p0).abs().max()
        assert float((got - want.detach()).abs().max()) <= 2e-3 * float(upd) + 1e-9


def test_fused_step_gradients_match_the_torch_path():
    """one forward + backward of the same batch on both paths.  The fused path's leaves are the EFFECTIVE weights (their gradients
    are written by the weight-gradient GEMMs straight into the flat bucket); pushing them through torch's own weight-norm backward
    must reproduce the (weight_g, weight_v) gradients of the torch path; biases, NeRF++ weights and the variance directly."""
    from nero_amd.train import ShapeTrainStep
    tu = ShapeTrainStep(CFG, rays_per_rank=192, pool_rays=768, device='cuda:0', variance=0.4, prime_fraction=0.0, fused=False)
    tf = ShapeTrainStep(CFG, rays_per_rank=192, pool_rays=768, device='cuda:0', variance=0.4, prime_fraction=0.0, fused=True)
    iu, if_ = tu.forward_backward(25000), tf.forward_backward(25000)
    torch.cuda.synchronize()
    assert iu['n_in'] == if_['n_in'] and abs(float(iu['loss']) - float(if_['loss'])) < 1e-6
    ref = {k: p.grad.detach().clone() for k, p in tu.net.named_parameters()}
    mods = dict(tf.net.named_modules())
    got = {}
    fo = tf.fopt
    for lin, w, _ in fo._wn:
        name = next(k for k, m in mods.items() if m is lin)
        v, g = lin.weight_v.detach().clone().requires_grad_(True), lin.weight_g.detach().clone().requires_grad_(True)
        torch._weight_norm(v, g, 0).backward(w.grad)
        got[name + '.weight_v'], got[name + '.weight_g'] = v.grad, g.grad
    for p in fo._plain:
        name = next(k for k, q in tf.net.named_parameters() if q is p)
        got[name] = p.grad
    assert set(got) == set(ref)
    # The two paths differ in the last bit of the effective weights (two weight-norm arithmetics).  Two amplifiers act on that:
    # the SDF's beta = 100 second-order terms (a few 1e-5 on its first layers) and ReLU units whose pre-activation sits within
    # that last bit of zero -- a flipped unit moves a gradient by one row's share, 1 / n_in ~ 4e-4 of the MLP's gradient scale at
    # this size (the same arithmetic-independent effect tests/helpers.py::assert_grads_fp32_grade accounts for).  Hence: every
    # tensor within a handful of row shares of its MLP's gradient scale, and the bulk at rounding level.
    group = lambda k: '.'.join(k.split('.')[:2])
    gscale = {}
    for k, r in ref.items():
        gscale[group(k)] = max(gscale.get(group(k), 0.0), float(r.abs().max()))
    row_share = 1.0 / max(int(iu['n_in']), 1)
    errs = []
    for k, r in ref.items():
        scale = float(r.abs().max())
        if scale < 1e-12:
            assert float(got[k].abs().max()) < 1e-12, k
            continue
        d = float((got[k] - r).abs().max())
        assert d <= 8 * row_share * gscale[group(k)] + 1e-4 * scale, (k, d / scale, d / gscale[group(k)], row_share)
        errs.append(d / scale)
    assert len(errs) > 100 and float(np.median(errs)) < 1e-5, float(np.median(errs))


def test_fused_training_steps_match_the_torch_path():
    """three optimisation steps on both paths.  Adam's update m / (sqrt(v) + eps) is sign-like per element, so an element whose
    gradient is ~eps may legitimately move by a whole step in the other direction; everything else must agree tightly."""
    from nero_amd.train import ShapeTrainStep
    runs = {}
    for fused in (False, True):
        ts = ShapeTrainStep(CFG, rays_per_rank=192, pool_rays=768, device='cuda:0', variance=0.4, prime_fraction=0.0, fused=fused)
        p0 = {k: v.detach().clone() for k, v in ts.net.state_dict().items()}
        losses = [float(ts.step(25000 + i)['loss']) for i in range(3)]
        torch.cuda.synchronize()
        runs[fused] = (p0, {k: v.detach().clone() for k, v in ts.net.state_dict().items()}, losses)
    (p0, pr, lr_), (q0, pf, lf) = runs[False], runs[True]
    assert all(torch.equal(p0[k], q0[k]) for k in p0)
    # (step 1 sees identical weights; afterwards the last-bit / ReLU-tie differences of the gradients feed back through Adam)
    assert abs(lr_[0] - lf[0]) < 1e-6 and np.allclose(lr_, lf, rtol=0, atol=1e-3), (lr_, lf)
    for k in pr:
        if k.endswith('FG_LUT'):
            continue
        upd = float((pr[k] - p0[k]).abs().max())
        d = (pf[k] - pr[k]).abs()
        assert float(d.mean()) <= 5e-3 * upd + 1e-9, (k, float(d.mean()), upd)
        assert float((d > 0.02 * upd + 1e-9).float().mean()) <= 0.02, (k, float((d > 0.02 * upd).float().mean()))
    assert upd > 0


def test_adam_job_without_gradient_is_skipped_like_torch():
    """nero_adam_job.step: a tensor with no gradient for the first steps (torch: .grad is None -> skipped, its step counter does not
    advance) must, once it has one, take torch.optim.Adam's FIRST-step update (bias corrections of step 1), while the other jobs keep
    counting.  ADVICE r2: deviation_network.variance across freeze_inv_s_step."""
    from nero_amd import _lib as L
    g0 = torch.Generator().manual_seed(1)
    p_a, p_b = torch.randn(5, generator=g0).cuda(), torch.randn(300, generator=g0).cuda()
    grads_a = [torch.randn(5, generator=g0).cuda() * 1e-2 for _ in range(5)]
    grads_b = [torch.randn(300, generator=g0).cuda() * 1e-2 for _ in range(5)]
    # torch: parameter a has no gradient during the first three steps
    ra, rb = p_a.clone().requires_grad_(True), p_b.clone().requires_grad_(True)
    opt = torch.optim.Adam([ra, rb], lr=3e-4)
    for it in range(5):
        ra.grad = None if it < 3 else grads_a[it].clone()
        rb.grad = grads_b[it].clone()
        opt.step()
    # kernels
    a, b = p_a.clone(), p_b.clone()
    z = torch.zeros_like
    ma, va, mb, vb = z(a), z(a), z(b), z(b)
    ga, gb = z(a), z(b)
    jobs = (L.AdamJob * 2)()
    jobs[0].p, jobs[0].grad, jobs[0].m, jobs[0].v, jobs[0].n = a.data_ptr(), ga.data_ptr(), ma.data_ptr(), va.data_ptr(), 5
    jobs[1].p, jobs[1].grad, jobs[1].m, jobs[1].v, jobs[1].n = b.data_ptr(), gb.data_ptr(), mb.data_ptr(), vb.data_ptr(), 300
    wn = (L.WnJob * 1)()
    own = 0
    for it in range(5):
        gb.copy_(grads_b[it])
        if it < 3:
            jobs[0].step = -1
        else:
            ga.copy_(grads_a[it])
            own += 1
            jobs[0].step = own
        jobs[1].step = it + 1
        L.check(L.lib.nero_wn_adam_batch(wn, 0, jobs, 2, C.c_float(3e-4), C.c_float(0.9), C.c_float(0.999), C.c_float(1e-8), it + 1, L.stream_ptr()))
        if it == 2:
            torch.cuda.synchronize()
            assert torch.equal(a, p_a) and float(ma.abs().max()) == 0.0 and float(va.abs().max()) == 0.0      # untouched while absent
    torch.cuda.synchronize()
    for got, want, p0 in ((a, ra, p_a), (b, rb, p_b)):
        upd = float((want.detach() - p0).abs().max())
        assert upd > 0 and float((got - want.detach()).abs().max()) <= 2e-3 * upd + 1e-9
    # and a job with step 0 follows the call's global step (the round-2 ABI)
    c1, c2 = p_b.clone(), p_b.clone()
    m1, v1, m2, v2 = z(c1), z(c1), z(c2), z(c2)
    j = (L.AdamJob * 2)()
    j[0].p, j[0].grad, j[0].m, j[0].v, j[0].n, j[0].step = c1.data_ptr(), gb.data_ptr(), m1.data_ptr(), v1.data_ptr(), 300, 0
    j[1].p, j[1].grad, j[1].m, j[1].v, j[1].n, j[1].step = c2.data_ptr(), gb.data_ptr(), m2.data_ptr(), v2.data_ptr(), 300, 7
    L.check(L.lib.nero_wn_adam_batch(wn, 0, j, 2, C.c_float(3e-4), C.c_float(0.9), C.c_float(0.999), C.c_float(1e-8), 7, L.stream_ptr()))
    torch.cuda.synchronize()
    assert float((c1 - c2).abs().max()) <= 1e-9


def test_variance_crosses_the_freeze_boundary_like_the_torch_path():
    """freeze_inv_s_step inside the run: two frozen steps, then three live ones.  The fused optimiser must leave the variance alone
    while frozen and then move it like torch.optim.Adam does from ITS first step (a global step counter made the first
    live updates 3-6x too large)."""
    from nero_amd.train import ShapeTrainStep
    cfg = {**CFG, 'freeze_inv_s_step': 25002}
    traj = {}
    for fused in (False, True):
        ts = ShapeTrainStep(cfg, rays_per_rank=192, pool_rays=768, device='cuda:0', variance=0.4, prime_fraction=0.0, fused=fused, prime_passes=0)
        v = []
        for i in range(5):
            ts.step(25000 + i)
            v.append(float(ts.net.deviation_network.variance))
        traj[fused] = v
    tu, tf = traj[False], traj[True]
    assert abs(tu[0] - 0.4) < 1e-7 and abs(tu[1] - 0.4) < 1e-7                       # torch path: .grad is None while frozen
    assert abs(tf[0] - 0.4) < 1e-7 and abs(tf[1] - 0.4) < 1e-7                       # frozen: untouched
    first_u, first_f = tu[2] - tu[1], tf[2] - tf[1]
    assert abs(first_u) > 1e-6
    # Adam's first update is lr * sign(g) (bias-corrected m / sqrt(v) = 1): both paths move by the learning rate of that step
    assert abs(first_f - first_u) <= 0.02 * abs(first_u), (first_u, first_f)
    assert abs((tf[4] - tf[1]) - (tu[4] - tu[1])) <= 0.05 * abs(tu[4] - tu[1]), (tu, tf)


def test_inference_cache_sees_fused_updates():
    """the no-grad cache of packed operand images is keyed on torch's version counters, which the fused Adam kernels do
    not bump -- forward_only / render_image / extract_fields after fused steps must use the NEW weights"""
    from nero_amd.train import ShapeTrainStep
    ts = ShapeTrainStep(CFG, rays_per_rank=128, pool_rays=256, device='cuda:0', variance=0.4, prime_fraction=0.0, fused=True, prime_passes=0)
    ts.cursor = 0
    a = ts.forward_only(25000).clone()
    ts.cursor = 0
    a2 = ts.forward_only(25000).clone()
    assert torch.equal(a, a2)                                   # cache hit, same weights
    for i in range(3):
        ts.step(25000 + i)
    ts.cursor = 0
    b = ts.forward_only(25000).clone()
    assert float((a - b).abs().max()) > 1e-6                    # three Adam steps later the render must have moved
    ts.net._kern_cache = None                                   # a freshly packed render of the same batch is the ground truth
    ts.cursor = 0
    c = ts.forward_only(25000)
    assert torch.equal(b, c)


@pytest.mark.gpu
def test_batched_weight_norm_node_matches_torch_per_linear(monkeypatch):
    """nero_amd/wn_fused.py (round 6): ALL weight-normed Linears of the drop-in renderer through ONE autograd node (nero_wn_forward_batch /
    nero_wn_backward_batch) against torch._weight_norm per Linear (NERO_WN_BATCH=0, the reference's own formulation): effective weights,
    the loss of a render step and every parameter gradient (weight_g, weight_v, bias -- the reference's state_dict names) agree to fp32
    rounding; gradients ACCUMULATE across two backward passes without zero_grad; the state_dict is untouched."""
    from nero_amd.renderer import NeROShapeRenderer
    from nero_amd.shape_step import flatten_effective
    from nero_amd.synthetic import perturb_state, synthetic_rays
    from nero_amd.train import shape_training_loss
    cfg = {'n_samples': 16, 'n_importance': 16, 'n_bg_samples': 8, 'freeze_inv_s_step': 15000, 'apply_occ_loss': True, 'occ_loss_step': 20000,
           'shader_config': {'human_light': True}}
    torch.manual_seed(6033)
    net = NeROShapeRenderer(cfg, training=False)
    perturb_state(net, 0.4)
    net = net.cuda()
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    o, d, poses_img, gt = synthetic_rays(192, seed=1)
    o, d, gt = o.cuda(), d.cuda(), gt.cuda()
    near, far = net.near_far_from_sphere(o, d)
    hp = torch.zeros(192, 3, 4, device='cuda')
    zv = net.sample_ray(o, d, near, far, 0.0)

    def run(batch, passes=1):
        monkeypatch.setenv('NERO_WN_BATCH', '1' if batch else '0')
        net.zero_grad(set_to_none=True)
        names, eff = flatten_effective(net)
        for _ in range(passes):
            out = net.render(o, d, near, far, hp, -1, 0.5, is_train=True, step=25000, z_vals=zv)
            loss = shape_training_loss(net, out, gt, 25000)
            loss.backward()
        torch.cuda.synchronize()
        return [e.detach().clone() for e in eff], float(loss), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    e0, l0, g0 = run(False)
    e1, l1, g1 = run(True)
    assert len(e0) == len(e1) and all(a.shape == b.shape for a, b in zip(e0, e1))
    worst_w = max(float((a - b).abs().max() / (a.abs().max() + 1e-30)) for a, b in zip(e0, e1))
    assert worst_w < 1e-6, worst_w
    assert abs(l0 - l1) < 1e-6 * max(1.0, abs(l0)), (l0, l1)
    assert set(g0) == set(g1) and len(g0) >= 3 * 37
    worst_g = max(float((g0[k] - g1[k]).abs().max() / (g0[k].abs().max() + 1e-30)) for k in g0 if float(g0[k].abs().max()) > 0)
    assert worst_g < 2e-4, worst_g                      # (two evaluations of the step with effective weights 1e-7 apart: ReLU ties aside, 1e-6)
    _, _, g2 = run(True, passes=2)                       # accumulation: twice the gradient
    worst_acc = max(float((g2[k] - 2 * g1[k]).abs().max() / (g1[k].abs().max() + 1e-30)) for k in g1 if float(g1[k].abs().max()) > 0)
    assert worst_acc < 1e-5, worst_acc
    sd1 = net.state_dict()
    assert all(torch.equal(sd0[k], sd1[k]) for k in sd0)
