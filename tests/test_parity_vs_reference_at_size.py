"""GPU tier: the HIP step against the UNMODIFIED REFERENCE at the benchmarked shape (VERDICT r5 missing 3 / weak 2).

tests/test_parity_at_size.py compares HIP with the ORACLE at 4096 rays, and the oracle is pinned to the reference at 48 rays x (16+16+8): sound,
but indirect.  Here the checker is the reference itself: tests/golden/at_size_{bell,bear}_1024.npz hold what the unmodified reference computes
on 1024 rays x (64+64+32) samples -- its own sampler with every up-sampling round recorded, then render + loss + backward on those z_vals in
float32 AND float64 (oracle/gen_golden_at_size.py, run in the build container where /root/reference exists; the reference is Python and
does not travel, its outputs do).  1024 rays is one rank's shard of BASELINE configs[2]; the round widths 64 / 80 / 96 / 112 and the 160-sample
rays are those of configs[1].

  * weights: rebuilt from the seed, checked against the stored checksums -- the HIP model IS the reference's model;
  * sampler: every round fed the REFERENCE's (z, sdf) of that round: section weights within 2e-5 of the reference's, searchsorted indices and the merge permutation BIT-EQUAL (the permutation up to exact ties, which the reference's unstable sort leaves open)
    through the exact-contract entry points (nero_sample_pdf on the reference's weights, nero_merge_sorted on its new z), at all four widths;
  * render step teacher-forced on the reference's z_vals: ray_rgb, gradient_error, std, loss_occ, loss within 1e-4 of the reference's float32
    run AND of its float64 run (north_star tolerance);
  * Stage II (test_stage2_step_on_the_references_hits): MCShadingNetwork on 1024 surface points x (128 + 128) directions, the HIP step answering
    its ONE trace call with the hits the reference's run obtained (its own secondary rays must equal the reference's to 2e-5: measured 5e-7);
  * parameter gradients on the stored 1024-entry sample of every tensor, errors relative to the tensor's largest float64 entry: <= 1e-4 of the
    reference-float64 gradient, or (a) <= 3 x the reference's OWN float32-vs-float64 distance on the tensors of the same MLP (ReLU ties the
    float64 run resolves differently), or (b) absolute error <= 1e-4 of the MLP's gradient scale; how often (a) / (b) were needed is recorded
    and bounded.
Everything measured lands in gpurun_out/parity_at_size.json under `ref_at_size[...]` (copied to profiles/r06_parity_vs_reference_at_size.json)."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

from tests.helpers import _mlp_of, parity_report

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _load(case):
    path = os.path.join(GOLDEN, f'at_size_{case}_1024.npz')
    if not os.path.exists(path):
        pytest.skip(f'{path} not generated (python oracle/gen_golden_at_size.py in the build container)')
    z = np.load(path)
    return z, json.loads(str(z['meta']))


def _model(meta, z):
    from nero_amd.renderer import NeROShapeRenderer
    from nero_amd.synthetic import perturb_state
    from oracle.golden_util import state_checksums
    from tests.helpers import ref_fg_lut
    torch.manual_seed(meta['seed'])
    net = NeROShapeRenderer(meta['cfg'], training=False)
    perturb_state(net, meta['variance'])
    net.color_network.FG_LUT.copy_(ref_fg_lut())          # the reference's table (its asset; the fixture holds it bit for bit)
    ck = state_checksums({k: v.detach().clone() for k, v in net.state_dict().items()})
    for k, v in ck.items():
        assert np.allclose(v, z['ck/' + k], rtol=1e-9, atol=1e-9), k
    net.train()
    return net.cuda()


def _t(z, k, dev='cuda', dtype=None):
    t = torch.from_numpy(np.asarray(z[k]))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(dev).contiguous()


def _rel(a, b):
    a, b = a.detach().double().cpu(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize('case', ['bell', 'bear'])
def test_sampler_rounds_on_the_references_own_inputs(case):
    from nero_amd import _lib as L
    z, meta = _load(case)
    nt = meta['n_trace']
    o, d = _t(z, 'o')[:nt].contiguous(), _t(z, 'd')[:nt].contiguous()
    st, P = L.stream_ptr(), C.c_void_p
    widths, worst_w, flips, n_idx, plain, ties = [], 0.0, 0, 0, 0, 0
    for i in range(4):
        zc, sc = _t(z, f'tr/z{i}'), _t(z, f'tr/sdf{i}')
        ref_new, ref_inds, ref_index = _t(z, f'tr/z_new{i}', 'cpu'), _t(z, f'tr/inds{i}', 'cpu').int(), _t(z, f'tr/index{i}', 'cpu').int()
        n, m = zc.shape[1], ref_new.shape[1]
        widths.append(n)
        z_new = torch.empty(nt, m, device='cuda'); w = torch.empty(nt, n - 1, device='cuda')
        inds = torch.empty(nt, m, dtype=torch.int32, device='cuda')
        L.check(L.lib.nero_upsample(P(o.data_ptr()), P(d.data_ptr()), P(zc.data_ptr()), n, P(sc.data_ptr()), n, n, P(None),
                                    C.c_float(float(z[f'tr/inv_s{i}'])), m, nt, P(z_new.data_ptr()), P(w.data_ptr()), P(inds.data_ptr()), st))
        # the kernel's own weights -> its own indices: identical to the reference's except where a 1-ulp weight difference crosses a cdf edge
        flips += int((inds.cpu() != ref_inds).sum())
        n_idx += ref_inds.numel()
        # exact contract 1: searchsorted (+ the new z) on the REFERENCE's weights (captured at its sample_pdf call, network/renderer.py:384)
        wt = _t(z, f'tr/weights{i}', 'cpu')
        worst_w = max(worst_w, _rel(w, wt))
        out = torch.empty(nt, m, device='cuda'); inds2 = torch.empty(nt, m, dtype=torch.int32, device='cuda')
        wtc = wt.cuda().contiguous()
        L.check(L.lib.nero_sample_pdf(P(zc.data_ptr()), n, P(wtc.data_ptr()), n - 1, n, m, nt, P(out.data_ptr()), P(inds2.data_ptr()), st))
        assert torch.equal(inds2.cpu(), ref_inds), (case, i, int((inds2.cpu() != ref_inds).sum()))
        # the new z = bins[below] + (u - cdf[below]) / (cdf[above] - cdf[below]) * bin width: where the cdf interval is tiny (2e-5 occurs at this
        # size; the 48-ray fixtures never get there) one ulp of a float32 cdf entry (6e-8) moves z by 6e-8 * width / interval, so the bound is
        # conditioned per sample; 99.9 % of the samples sit inside the plain 2e-6
        cdf = torch.cat([torch.zeros(nt, 1, dtype=torch.float64), torch.cumsum((wt.double() + 1e-5) / (wt.double() + 1e-5).sum(-1, keepdim=True), -1)], -1)
        above = ref_inds.long().clamp(max=n - 1)
        below = (ref_inds.long() - 1).clamp(min=0)
        denom = torch.gather(cdf, 1, above) - torch.gather(cdf, 1, below)
        denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)                     # network/field.py:421-422
        width = torch.gather(zc.cpu().double(), 1, above) - torch.gather(zc.cpu().double(), 1, below)
        dz = (out.cpu().double() - ref_new.double()).abs()
        assert bool((dz <= 2e-6 + 4e-7 * width / denom).all()), (case, i, float((dz / (2e-6 + 4e-7 * width / denom)).max()))
        plain += int((dz <= 2e-6).sum())
        # exact contract 2: the merge permutation on the reference's new z
        zt = torch.empty(nt, n + m, device='cuda'); zt[:, :n] = zc
        index = torch.empty(nt, n + m, dtype=torch.int32, device='cuda')
        zn = ref_new.cuda().contiguous()
        L.check(L.lib.nero_merge_sorted(P(zt.data_ptr()), n + m, n, P(None), 0, P(zn.data_ptr()), m, P(None), 0, nt, P(index.data_ptr()), st))
        # ... up to EXACT ties: the reference sorts with torch.sort (network/renderer.py:398, not stable), so where a new z equals an existing one
        # bit for bit (sample_pdf returns bins[below] itself when u sits on a cdf edge; the bear fixture has three such pairs, the bell fixture
        # none) the order of the two equal entries is whatever the sort implementation leaves.  The kernel's rule is old before new, lower
        # index first (a stable sort of the concatenation); everything else must be the reference's permutation, and the sorted values are
        # identical either way
        zall = torch.cat([zc.cpu(), ref_new], 1)
        mine, ref = index.cpu().long(), ref_index.long()
        assert torch.equal(torch.gather(zall, 1, mine), torch.gather(zall, 1, ref)), (case, i)
        diff = mine != ref
        ties += int(diff.sum())
        assert int(diff.sum()) <= 8 and torch.equal(mine, torch.sort(zall, dim=1, stable=True)[1]), (case, i, int(diff.sum()))
    assert widths == [64, 80, 96, 112]
    assert worst_w < 2e-5 and flips / n_idx < 2e-3, (worst_w, flips, n_idx)
    parity_report(f'ref_at_size[{case}].sampler', rays=nt, widths=widths, indices_compared=n_idx, bit_equal_on_reference_weights=True,
                  own_weights_index_flips=flips, worst_weight_rel_err=worst_w, new_z_within_2e_6=plain / n_idx,
                  merge_entries_differing_at_exact_ties=ties)
    assert plain / n_idx > 0.995


@pytest.mark.parametrize('case', ['bell', 'bear'])
def test_free_running_sampler_against_the_references_z_vals(case):
    """No teacher forcing: the product's own hierarchical sampler (all four rounds, its own SDF evaluations) on the reference's rays and draws,
    against the z_vals the unmodified reference produced.  The sampler is ill-conditioned end to end (one SDF value within rounding of a section
    boundary moves a sample into the neighbouring section: SURVEY.md 0.2; tests/test_parity_at_size.py measures the same between two fp32
    evaluations of the oracle), so this MEASURES and bounds: the background z (no SDF involved) exact to rounding, the fraction of inner z equal
    to 1e-5, the fraction of rays with every z equal, and the colours rendered on the product's own z against the reference's float32 colours."""
    z, meta = _load(case)
    net = _model(meta, z)
    R, nb = meta['R'], int(z['rand_bg'].shape[1])
    with torch.no_grad():
        zg = net.sample_ray(_t(z, 'o'), _t(z, 'd'), _t(z, 'near'), _t(z, 'far'), 1.0, _t(z, 'rand1'), _t(z, 'rand_bg'))
        out = net.render(_t(z, 'o'), _t(z, 'd'), _t(z, 'near'), _t(z, 'far'), _t(z, 'human_poses'), -1, meta['anneal'], is_train=True,
                         step=meta['step'], z_vals=zg)
    zg, zr, zr64 = zg.cpu(), torch.from_numpy(z['z_vals']), torch.from_numpy(z['z_vals64'])
    assert zg.shape == zr.shape == zr64.shape

    def stats(a, b_):
        dz = (a[:, :-nb] - b_[:, :-nb]).abs()
        return dict(frac_z_equal_1e5=float((dz < 1e-5).float().mean()), frac_rays_all_z_equal_1e5=float((dz.max(-1)[0] < 1e-5).float().mean()),
                    worst_dz=float(dz.max()))
    e = (out['ray_rgb'].cpu() - torch.from_numpy(z['ray_rgb32'])).abs().max(-1)[0]
    rec = dict(rays=R, hip_vs_reference_fp32=stats(zg, zr), hip_vs_reference_fp64=stats(zg, zr64), reference_fp32_vs_fp64=stats(zr, zr64),
               bg_z_rel=float((zg[:, -nb:] / zr[:, -nb:] - 1).abs().max()),
               rgb_abs_err_median=float(e.median()), rgb_abs_err_p99=float(e.kthvalue(int(0.99 * R))[0]), rgb_abs_err_worst=float(e.max()))
    parity_report(f'ref_at_size[{case}].free_running_sampler', **rec)
    assert rec['bg_z_rel'] < 1e-6
    # measured (MI355X, round 6): the reference's OWN float32 sampler keeps 91 % of the z (46 % of the rays entirely) of its float64 sampler on
    # the same draws -- the floor; the HIP sampler is at least as close to the float64 z_vals as that, and closer to the reference's float32
    # z_vals than those are to the exact ones
    floor = rec['reference_fp32_vs_fp64']
    assert rec['hip_vs_reference_fp64']['frac_z_equal_1e5'] >= floor['frac_z_equal_1e5'] - 0.02, rec
    assert rec['hip_vs_reference_fp64']['frac_rays_all_z_equal_1e5'] >= floor['frac_rays_all_z_equal_1e5'] - 0.05, rec
    assert rec['hip_vs_reference_fp32']['frac_z_equal_1e5'] >= floor['frac_z_equal_1e5'] - 0.02, rec
    # the rendering integral barely notices a moved sample: colours on the product's own z against the reference's float32 colours
    assert rec['rgb_abs_err_median'] < 2e-6 and rec['rgb_abs_err_p99'] < 2e-5 and rec['rgb_abs_err_worst'] < 2e-3, rec


# measured on MI355X (round 6): clause (a) -- the reference's own fp32 run is equally far from its fp64 run -- see profiles/r06_parity_vs_reference_at_size.json
MAX_CLAUSE_A, MAX_CLAUSE_B = 40, 2


@pytest.mark.parametrize('case', ['bell', 'bear'])
def test_render_step_on_the_references_z_vals(case):
    from nero_amd.train import shape_training_loss
    z, meta = _load(case)
    net = _model(meta, z)
    keys = torch.rand(meta['R'] * 160, generator=torch.Generator().manual_seed(meta['occ_keys_seed'])).cuda()
    out = net.render(_t(z, 'o'), _t(z, 'd'), _t(z, 'near'), _t(z, 'far'), _t(z, 'human_poses'), -1, meta['anneal'], is_train=True,
                     step=meta['step'], z_vals=_t(z, 'z_vals'), occ_keys=keys)
    loss = shape_training_loss(net, out, _t(z, 'gt'), meta['step'])
    loss.backward()
    torch.cuda.synchronize()
    rep = {'rays': meta['R'], 'n_in': int(out['gradient_error'].shape[0])}
    assert out['gradient_error'].shape[0] == z['gradient_error32'].shape[0]        # the same inner / outer split, sample for sample
    for tag in ('32', '64'):
        e = dict(ray_rgb=_rel(out['ray_rgb'], z['ray_rgb' + tag]), gradient_error=_rel(out['gradient_error'], z['gradient_error' + tag]),
                 std=abs(float(out['std']) - float(z['std' + tag])) / abs(float(z['std' + tag])),
                 loss_occ=abs(float(out['loss_occ']) - float(z['loss_occ' + tag])) / max(abs(float(z['loss_occ' + tag])), 1e-12),
                 loss=abs(float(loss) - float(z['loss' + tag])) / abs(float(z['loss' + tag])))
        rep['vs_reference_fp' + tag] = e
        assert max(e.values()) < 1e-4, (case, tag, e)
    rep['reference_fp32_vs_fp64'] = dict(ray_rgb=_rel(torch.from_numpy(z['ray_rgb32']), z['ray_rgb64']),
                                         loss=abs(float(z['loss32']) - float(z['loss64'])) / abs(float(z['loss64'])))
    # ---- gradients on the stored sample
    from oracle.gen_golden_at_size import sample_index
    names = [k[4:] for k in z.files if k.startswith('g64/')]
    grads = {k: (p.grad.detach().double().reshape(-1).cpu() if p.grad is not None else torch.zeros(p.numel(), dtype=torch.float64))
             for k, p in net.named_parameters()}
    assert set(names) == set(grads)
    err, floor_t, abs_err, gscale, floor = {}, {}, {}, {}, {}
    for k in names:
        g64, g32, mx = torch.from_numpy(z['g64/' + k]), torch.from_numpy(z['g32/' + k]).double(), float(z['max64/' + k])
        gh = grads[k][torch.from_numpy(sample_index(grads[k].numel()))]
        grp = _mlp_of(k)
        gscale[grp] = max(gscale.get(grp, 0.0), mx)
        if mx < 1e-12:
            assert float(gh.abs().max()) < 1e-12, k
            continue
        err[k], abs_err[k] = float((gh - g64).abs().max()) / mx, float((gh - g64).abs().max())
        floor_t[k] = float((g32 - g64).abs().max()) / mx
        floor[grp] = max(floor.get(grp, 0.0), floor_t[k])
    plain = [k for k in err if err[k] <= 1e-4]
    a = [k for k in err if err[k] > 1e-4 and err[k] <= 3.0 * floor[_mlp_of(k)]]
    b = [k for k in err if k not in plain and k not in a and abs_err[k] <= 1e-4 * gscale[_mlp_of(k)]]
    bad = {k: (err[k], floor[_mlp_of(k)]) for k in err if k not in plain and k not in a and k not in b}
    vals = np.array(list(err.values()))
    rep['gradients'] = dict(n_tensors=len(err), n_plain=len(plain), n_clause_a=len(a), n_clause_b=len(b), clause_a=a, clause_b=b,
                            median_err=float(np.median(vals)), max_err=float(vals.max()),
                            median_reference_fp32_floor=float(np.median(list(floor_t.values()))), max_reference_fp32_floor=float(max(floor_t.values())),
                            worst=sorted(((k, err[k], floor[_mlp_of(k)]) for k in err), key=lambda t: -t[1])[:5])
    parity_report(f'ref_at_size[{case}].render_step', **rep)
    assert not bad, bad
    assert len(a) <= MAX_CLAUSE_A and len(b) <= MAX_CLAUSE_B, (len(a), len(b))
    assert np.median(vals) < max(2e-5, 3.0 * rep['gradients']['median_reference_fp32_floor'])


# ---- Stage II: MCShadingNetwork on 1024 surface points x (128 + 128) directions, the reference's own hits replayed --------------------------------
class _FixtureTracer:
    """RayTracer-shaped: answers the step's ONE trace call (all P x D secondary rays, network/field.py:856-880) with the hits the reference's
    float32 run obtained from the float64 brute-force tracer, slot by slot, and checks every 64th incoming ray against the reference's ray of
    that slot (the directions are computed by the code under test: nero_mc_dirs)"""

    def __init__(self, z, meta):
        self.depth = np.asarray(z['hit_depth'])
        self.hit = self.depth < 10
        self.pos_hit, self.nrm_hit = np.asarray(z['hit_pos']), np.asarray(z['hit_nrm'])
        self.ro, self.rd, self.stride = np.asarray(z['ray_o']), np.asarray(z['ray_d']), meta['ray_stride']
        self.calls, self.max_ray_dev = 0, 0.0

    def trace(self, o, d):
        assert self.calls == 0 and o.shape[0] == self.depth.shape[0], (self.calls, o.shape)
        self.calls += 1
        on, dn = o.detach().cpu().numpy(), d.detach().cpu().numpy()
        self.max_ray_dev = max(float(np.abs(on[::self.stride] - self.ro).max()), float(np.abs(dn[::self.stride] - self.rd).max()))
        assert self.max_ray_dev <= 2e-5, self.max_ray_dev
        pos = on + dn * self.depth[:, None]                       # (misses: never read behind the hit mask)
        nrm = np.zeros_like(on)
        nrm[:, 2] = 1.0
        pos[self.hit], nrm[self.hit] = self.pos_hit, self.nrm_hit
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(o.device)
        return f(pos), f(nrm), f(self.depth)


@pytest.mark.parametrize('case', ['mat_bell', 'mat_bear'])
def test_stage2_step_on_the_references_hits(case):
    """mat_bell: configs/material/syn/bell.yaml's light model; mat_bear: configs/material/real/bear.yaml's (the photographer's light on the
    camera plane + the sphere-direction outer light), its own weights and surface points"""
    from nero_amd.renderer import NeROMaterialRenderer
    from nero_amd.synthetic import perturb_state
    from oracle.gen_golden_at_size import sample_index
    from oracle.golden_util import state_checksums
    from tests.helpers import MatHolder, golden_mesh
    path = os.path.join(GOLDEN, f'at_size_{case}_1024.npz')
    if not os.path.exists(path):
        pytest.skip(f'{path} not generated (python oracle/gen_golden_at_size.py {case} in the build container)')
    z = np.load(path)
    meta = json.loads(str(z['meta']))
    cfg = meta['shader_cfg']
    torch.manual_seed(meta['seed'])
    ref = MatHolder(cfg)
    perturb_state(ref, None)
    for k, v in state_checksums({k: v.detach().clone() for k, v in ref.state_dict().items()}).items():
        assert np.allclose(v, z['ck/' + k], rtol=1e-9, atol=1e-9), k
    net = NeROMaterialRenderer({'shader_cfg': cfg, 'database_name': 'real/bear' if cfg['human_lights'] else 'syn/bell'}, mesh=golden_mesh())
    net.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    net = net.cuda()
    tr = net.ray_tracer = _FixtureTracer(z, meta)
    c = lambda k: _t(z, k)
    out = net.shade_train(c('pts'), c('view'), c('normals'), c('human_poses'), c('gt'), meta['step'], c('rand_d'), c('rand_s'), c('reg_ang'), c('reg_eps'))
    loss = out['loss_rgb'].mean() + out['loss_mat_reg'].mean() + out['loss_diffuse_light'].mean()
    loss.backward()
    torch.cuda.synchronize()
    assert tr.calls == 1
    rep = {'points': meta['P'], 'directions': cfg['diffuse_sample_num'] + cfg['specular_sample_num'], 'rays': meta['n_rays'], 'hit_fraction': meta['n_hit'] / meta['n_rays'],
           'max_secondary_ray_deviation': tr.max_ray_dev}
    for tag in ('32', '64'):
        e = {'rgb_pr': _rel(out['rgb_pr'], z['rgb' + tag]), 'loss': abs(float(loss) - float(z['loss' + tag])) / abs(float(z['loss' + tag])),
             'loss_diffuse_light': _rel(out['loss_diffuse_light'], z['loss_white' + tag])}
        for k in ('albedo', 'roughness', 'metallic', 'diffuse_light', 'specular_light', 'specular_color'):
            e[k] = _rel(out[k], z[f'out{tag}/{k}'])
        rep['vs_reference_fp' + tag] = e
        assert max(v for k, v in e.items() if k != 'loss_diffuse_light') < 1e-4, (tag, e)
    # the per-point diffuse-light regulariser sum|dl - mean(dl)| is a cancellation of three nearly equal channels (measured 7e-5 ... 1e-4, the
    # reference's own float32 run as far from its float64 run): like loss_mat_reg below it is held to the float64 reference or 3 x that floor
    w64 = torch.from_numpy(z['loss_white64'])
    rep['loss_diffuse_light_vs_fp64'] = dict(hip=_rel(out['loss_diffuse_light'], w64), reference_fp32=_rel(torch.from_numpy(z['loss_white32']), w64))
    assert rep['loss_diffuse_light_vs_fp64']['hip'] <= max(1e-4, 3.0 * rep['loss_diffuse_light_vs_fp64']['reference_fp32']), rep['loss_diffuse_light_vs_fp64']
    # loss_mat_reg = |m(p) - m(p + eps)| of two nearly equal predictions: held to the float64 reference, or 3 x the reference's own fp32 distance
    reg64 = torch.from_numpy(z['loss_mat_reg64'])
    rep['loss_mat_reg_vs_fp64'] = dict(hip=_rel(out['loss_mat_reg'], reg64), reference_fp32=_rel(torch.from_numpy(z['loss_mat_reg32']), reg64))
    assert rep['loss_mat_reg_vs_fp64']['hip'] <= max(1e-4, 3.0 * rep['loss_mat_reg_vs_fp64']['reference_fp32']), rep['loss_mat_reg_vs_fp64']
    names = [k[4:] for k in z.files if k.startswith('g64/')]
    grads = {k: (p.grad.detach().double().reshape(-1).cpu() if p.grad is not None else torch.zeros(p.numel(), dtype=torch.float64)) for k, p in net.named_parameters()}
    assert set(names) == set(grads), sorted(set(names) ^ set(grads))[:6]
    err, floor_t, abs_err, gscale, floor = {}, {}, {}, {}, {}
    for k in names:
        g64, g32, mx = torch.from_numpy(z['g64/' + k]), torch.from_numpy(z['g32/' + k]).double(), float(z['max64/' + k])
        gh = grads[k][torch.from_numpy(sample_index(grads[k].numel()))]
        grp = _mlp_of(k)
        gscale[grp] = max(gscale.get(grp, 0.0), mx)
        if mx < 1e-12:
            assert float(gh.abs().max()) < 1e-12, k
            continue
        err[k], abs_err[k] = float((gh - g64).abs().max()) / mx, float((gh - g64).abs().max())
        floor_t[k] = float((g32 - g64).abs().max()) / mx
        floor[grp] = max(floor.get(grp, 0.0), floor_t[k])
    plain = [k for k in err if err[k] <= 1e-4]
    a = [k for k in err if err[k] > 1e-4 and err[k] <= 3.0 * floor[_mlp_of(k)]]
    b = [k for k in err if k not in plain and k not in a and abs_err[k] <= 1e-4 * gscale[_mlp_of(k)]]
    bad = {k: (err[k], floor[_mlp_of(k)]) for k in err if k not in plain and k not in a and k not in b}
    vals = np.array(list(err.values()))
    rep['gradients'] = dict(n_tensors=len(err), n_plain=len(plain), n_clause_a=len(a), n_clause_b=len(b), clause_a=a, clause_b=b,
                            median_err=float(np.median(vals)), max_err=float(vals.max()), median_reference_fp32_floor=float(np.median(list(floor_t.values()))),
                            worst=sorted(((k, err[k], floor[_mlp_of(k)]) for k in err), key=lambda t: -t[1])[:5])
    parity_report(f'ref_at_size[{case}].stage2_step', **rep)
    assert not bad, bad
    assert len(a) <= MAX_CLAUSE_A and len(b) <= MAX_CLAUSE_B, (len(a), len(b))
