"""GPU tier: the HIP glue of the Stage-I training step (nero_amd/csrc/step_glue.hip, nero_amd.stage1.ShapeStepGlue) against the tensor
glue it replaces -- NeROShapeRenderer.near_far_from_sphere, the candidate subset of compute_occ_loss (network/renderer.py:528-541),
compute_rgb_loss + the means of the three loss terms (network/renderer.py:332-343, train/trainer.py:127-137) and autograd's seeds --
kernel by kernel, and the whole step gradient by gradient on the same batch and the same random draws."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from nero_amd import _lib as L
    from nero_amd import stage1
    return L, stage1._lib, stage1._p


def test_near_far_vs_tensor_expression():
    from nero_amd.renderer import NeROShapeRenderer
    from nero_amd.synthetic import synthetic_rays
    L, lib, p = _lib()
    o, d, _, _ = synthetic_rays(5000, seed=3)
    o = o.cuda()
    near, far = torch.empty(5000, 1, device='cuda'), torch.empty(5000, 1, device='cuda')
    for scale, exact in ((1.0, True), (1.7, False)):          # (unit directions: what every caller passes; scaled: a != 1)
        d_ = (d * scale).cuda()
        L.check(lib.nero_near_far_sphere(p(o), p(d_), 5000, p(near), p(far), L.stream_ptr()))
        n_ref, f_ref = NeROShapeRenderer.near_far_from_sphere(o, d_)
        if exact:                                             # the kernel adds the three products in ATen's order: bit-identical
            assert torch.equal(near, n_ref) and torch.equal(far, f_ref)
        assert float((near - n_ref).abs().max()) <= 1e-6 and float((far - f_ref).abs().max()) <= 1e-6
    # a ray that starts inside the sphere: the clamp
    o2 = torch.tensor([[0.1, 0.0, 0.0]], device='cuda')
    d2 = torch.tensor([[0.0, 0.0, 1.0]], device='cuda')
    L.check(lib.nero_near_far_sphere(p(o2), p(d2), 1, p(near), p(far), L.stream_ptr()))
    assert float(near[0]) == pytest.approx(1e-3) and float(far[0]) == pytest.approx(1.0)


@pytest.mark.parametrize('n,frac,cap,ties', [(5000, 0.3, 256, False), (5000, 0.02, 256, False), (70000, 0.5, 2048, False), (3000, 0.0, 64, False),
                                             (1, 1.0, 4, False), (9000, 0.4, 100, True), (1025, 1.0, 1024, False)])
def test_occ_select_vs_nonzero_argsort_sort(n, frac, cap, ties):
    """counts, order and content of the kept candidate list = torch.nonzero + argsort(keys[:Pn], stable)[:cap] + sort"""
    L, lib, p = _lib()
    g = torch.Generator().manual_seed(n + cap)
    flag = (torch.rand(n, generator=g) < frac).to(torch.uint8).cuda()
    keys = torch.rand(n, generator=g)
    if ties:
        keys = (keys * 16).floor() / 16                       # many equal keys: the stable order decides
    keys = keys.cuda()
    cand, counts = torch.full((cap,), -7, dtype=torch.int32, device='cuda'), torch.zeros(2, dtype=torch.int32, device='cuda')
    ws = torch.empty(lib.nero_occ_select_workspace(n), dtype=torch.uint8, device='cuda')
    L.check(lib.nero_occ_select(p(flag), n, p(keys), cap, p(cand), p(counts), ws.data_ptr(), ws.numel(), L.stream_ptr()))
    ref = torch.nonzero(flag)[:, 0]
    Pn = ref.numel()
    if Pn > cap:
        ref = ref[torch.sort(torch.argsort(keys[:Pn], stable=True)[:cap])[0]]
    kept = ref.numel()
    assert counts.tolist() == [kept, Pn]
    assert torch.equal(cand[:kept].long(), ref) and bool((cand[kept:] == -1).all())


@pytest.mark.parametrize('kind', ['l2', 'l1', 'smooth_l1', 'charbonier'])
@pytest.mark.parametrize('weighted', [False, True])
def test_shape_loss_and_seeds_vs_autograd(kind, weighted):
    from nero_amd.renderer import NeROShapeRenderer
    from nero_amd.stage1 import RGB_LOSS_KIND
    L, lib, p = _lib()
    R, n_in, cap = 700, 9000, 300
    g = torch.Generator().manual_seed(5)
    rgb, gt = torch.rand(R, 3, generator=g).cuda(), torch.rand(R, 3, generator=g).cuda()
    rgb[:5] = gt[:5]                                         # exact hits: the sub-gradient conventions
    gerr, occ = torch.rand(n_in, generator=g).cuda(), torch.rand(n_in, generator=g).cuda()
    kept = 173
    cand = torch.full((cap,), -1, dtype=torch.int32, device='cuda')
    cand[:kept] = torch.sort(torch.randperm(n_in, generator=g)[:kept])[0].int().cuda()
    counts = torch.tensor([kept, 5 * kept], dtype=torch.int32, device='cuda')
    gt_occ = torch.rand(cap, generator=g).cuda()
    gt_occ[3] = occ[cand[3].item()]                          # |0|
    w = torch.tensor([0.7, 1.9], device='cuda') if weighted else None
    eik_w = 0.1
    losses = torch.zeros(4, device='cuda')
    d_rgb, d_gerr, d_occ = torch.empty(R, 3, device='cuda'), torch.empty(n_in, device='cuda'), torch.full((n_in,), 3.0, device='cuda')
    part = torch.empty(lib.nero_shape_loss_partials(R, n_in), device='cuda')
    L.check(lib.nero_shape_loss(R, RGB_LOSS_KIND[kind], p(rgb), p(gt), n_in, p(gerr), eik_w, p(occ), p(cand), p(counts), p(gt_occ), p(w), p(losses),
                                p(d_rgb), p(d_gerr), p(d_occ), p(part), L.stream_ptr()))
    a, b, c = rgb.clone().requires_grad_(True), gerr.clone().requires_grad_(True), occ.clone().requires_grad_(True)

    class Net:
        cfg = {'rgb_loss': kind}
    l_rgb = NeROShapeRenderer.compute_rgb_loss(Net, a, gt).mean()
    l_eik = (b * eik_w).mean() * (w[0] if weighted else 1.0)
    l_occ = torch.nn.functional.l1_loss(c[cand[:kept].long()], gt_occ[:kept]) * (w[1] if weighted else 1.0)
    (l_rgb + l_eik + l_occ).backward()
    ref = torch.stack([l_rgb + l_eik + l_occ, l_rgb, l_eik, l_occ]).detach()
    assert float((losses - ref).abs().max()) < 2e-6, (losses, ref)
    assert float((d_rgb - a.grad).abs().max()) <= 1e-8 + 1e-5 * float(a.grad.abs().max())
    assert float((d_gerr - b.grad).abs().max()) <= 1e-6 * float(b.grad.abs().max())
    assert float((d_occ - c.grad).abs().max()) <= 1e-6 * float(c.grad.abs().max())
    # without the occlusion term
    L.check(lib.nero_shape_loss(R, RGB_LOSS_KIND[kind], p(rgb), p(gt), n_in, p(gerr), eik_w, None, None, None, None, p(w), p(losses),
                                p(d_rgb), p(d_gerr), None, p(part), L.stream_ptr()))
    assert float((losses[0] - (l_rgb + l_eik)).abs()) < 2e-6 and float(losses[3]) == 0.0


CASES = {
    'bell_occ_capped': ({'occ_loss_max_pn': 200}, 25000),
    'bell_occ_default_cap': ({}, 25000),
    'bell_no_occ_frozen_variance': ({'freeze_inv_s_step': 15000}, 5000),
    'bear_human_light_l1': ({'shader_config': {'human_light': True}, 'rgb_loss': 'l1', 'occ_loss_max_pn': 333}, 25000),
    'bell_l2_no_perturb': ({'rgb_loss': 'l2', 'perturb': 0.0}, 25000),
}


@pytest.mark.parametrize('case', sorted(CASES))
def test_glued_step_matches_the_tensor_glue(case, monkeypatch):
    """the whole step on the same batch, the same random draws and the same occlusion keys: loss and every parameter gradient of the
    flat bucket (and d loss / d variance) from ShapeStepGlue against net.render + shape_training_loss + autograd"""
    from nero_amd.train import ShapeTrainStep
    cfg, step = CASES[case]
    R = 384
    ts = ShapeTrainStep(cfg, rays_per_rank=R, pool_rays=4 * R, device='cuda', variance=0.5, prime_fraction=0.0, prime_passes=0)
    c = ts.net.cfg
    g = torch.Generator().manual_seed(11)
    rands = (torch.rand(R, 1, generator=g).cuda(), torch.rand(R, c['n_bg_samples'], generator=g).cuda(), torch.rand(R * 160, generator=g).cuda())

    def run(mode):
        monkeypatch.setenv('NERO_STEP_GLUE', mode)
        ts.cursor = 0
        info = ts.forward_backward(step, rands)
        torch.cuda.synchronize()
        return float(info['loss']), ts.bucket.flat.clone(), info

    l_t, g_t, i_t = run('torch')
    l_h, g_h, i_h = run('hip')
    assert ts._glue_obj is not None and 'loss_terms' in i_h and 'loss_terms' not in i_t          # the second run really took the HIP glue
    assert (i_t['n_in'], i_t['n_out']) == (i_h['n_in'], i_h['n_out'])
    assert abs(l_t - l_h) <= 2e-6 * max(1.0, abs(l_t)), (l_t, l_h)
    # (observed: the two paths agree bit for bit -- same near / far, same draws, same seeds up to the rounding of 1 / R)
    scale = float(g_t.abs().max())
    assert scale > 0 and float((g_t - g_h).abs().max()) <= 2e-5 * scale, (float((g_t - g_h).abs().max()), scale)
    # per tensor: nothing hides under the bucket's largest entry
    for name, view in ts.fopt.grad_views.items():
        off = view.data_ptr() - ts.bucket.flat.data_ptr()
        a = g_t[off // 4: off // 4 + view.numel()]
        b = g_h[off // 4: off // 4 + view.numel()]
        s = float(a.abs().max())
        assert float((a - b).abs().max()) <= 1e-4 * s + 1e-9, (name, float((a - b).abs().max()), s)
    # and the optimiser takes it from there
    ts.step(step + 1)
    torch.cuda.synchronize()


def test_returned_loss_does_not_alias_the_glue_buffer():
    """two steps: the first step's `loss` keeps its value after the second step rewrote the glue's loss buffer (scripts/soak.py stacks
    the losses of hundreds of steps)"""
    from nero_amd.train import ShapeTrainStep
    ts = ShapeTrainStep({}, rays_per_rank=128, pool_rays=512, device='cuda', variance=0.5, prime_fraction=0.0, prime_passes=0)
    a = ts.step(25000)
    va = float(a['loss'])
    b = ts.step(25001)
    torch.cuda.synchronize()
    assert 'loss_terms' in a and float(a['loss']) == va and float(b['loss']) != va


@pytest.mark.parametrize('case', ['all_rays_miss_the_sphere', 'ragged_65_rays_some_missing'])
def test_glued_step_edge_batches(case, monkeypatch):
    """no inner sample at all (every ray passes the unit sphere at distance > 1: no SDF rows, no eikonal term, no occlusion candidates)
    and a ragged batch mixing such rays with ordinary ones: the glued step equals the tensor-glued step there too"""
    from nero_amd.train import ShapeTrainStep
    R = 64 if case.startswith('all') else 65
    ts = ShapeTrainStep({'occ_loss_max_pn': 50}, rays_per_rank=R, pool_rays=2 * R, device='cuda', variance=0.5, prime_fraction=0.0, prime_passes=0)
    o, d = ts.pool['o'], ts.pool['d']
    miss = torch.ones(o.shape[0], dtype=torch.bool, device='cuda') if case.startswith('all') else (torch.arange(o.shape[0], device='cuda') % 3 == 0)
    # move the origin of the chosen rays sideways by 3: their closest approach to the centre is then > 1
    side = torch.nn.functional.normalize(torch.cross(d, torch.tensor([[0.3, -0.5, 0.8]], device='cuda').expand_as(d), dim=-1), dim=-1)
    ts.pool['o'] = torch.where(miss[:, None], o + 3.0 * side, o).contiguous()
    g = torch.Generator().manual_seed(4)
    c = ts.net.cfg
    rands = (torch.rand(R, 1, generator=g).cuda(), torch.rand(R, c['n_bg_samples'], generator=g).cuda(), torch.rand(R * 160, generator=g).cuda())
    res = {}
    for mode in ('torch', 'hip'):
        monkeypatch.setenv('NERO_STEP_GLUE', mode)
        ts.cursor = 0
        info = ts.forward_backward(25000, rands)
        torch.cuda.synchronize()
        res[mode] = (float(info['loss']), info['n_in'], info['n_out'], ts.bucket.flat.clone())
    (lt, nit, not_, gt_), (lh, nih, noh, gh) = res['torch'], res['hip']
    assert (nit, not_) == (nih, noh) and (nit == 0) == case.startswith('all') and not_ > 0
    assert abs(lt - lh) <= 2e-6 * max(1.0, abs(lt)) and lt == lt
    assert float((gt_ - gh).abs().max()) <= 1e-5 * float(gt_.abs().max()) + 1e-12



@pytest.mark.parametrize('n_in,cap,kept', [(5000, 256, 256), (5000, 256, 100), (70000, 2048, 2048), (300, 64, 0), (4097, 4096, 4096)])
def test_occ_l1_node_vs_tensor_expression(n_in, cap, kept):
    """nero_occ_l1 / _backward (the drop-in renderer's occlusion term) = the tensor expression it replaced, with its autograd gradient:
    mean over the kept slots of |occ_prob[cand] - gt|, sign(0) = 0, zero gradient away from the candidates"""
    from nero_amd.shape_step import OccL1
    g = torch.Generator().manual_seed(n_in + cap + kept)
    occ = torch.rand(n_in, generator=g).cuda()
    idx = torch.sort(torch.randperm(n_in, generator=g)[:kept])[0].int()
    cand = torch.full((cap,), -1, dtype=torch.int32)
    cand[:kept] = idx
    cand = cand.cuda()
    counts = torch.tensor([kept, kept + 5], dtype=torch.int32).cuda()
    gt = torch.rand(cap, generator=g).cuda()
    if kept > 2:
        gt[1] = occ[cand[1].long()]                              # an exact tie: sub-gradient 0
    a = occ.clone().requires_grad_(True)
    loss = OccL1.apply(a, cand, counts, gt)
    (loss * 0.7).backward()
    b = occ.clone().requires_grad_(True)
    valid = cand >= 0
    ref = ((b[cand.clamp(min=0).long()] - gt).abs() * valid).sum() / counts[0].clamp(min=1).float()
    (ref * 0.7).backward()
    assert loss.shape == () and float((loss - ref).abs()) <= 2e-6 * max(1.0, float(ref.abs()))
    assert torch.equal(a.grad != 0, b.grad != 0)
    assert float((a.grad - b.grad).abs().max()) <= 1e-7 * max(1e-30, float(b.grad.abs().max())) + 1e-12
