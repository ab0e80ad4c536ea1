"""The C-level Stage-II shading driver (include/nero_hip.h: nero_stage2_*; SURVEY.md 8b "nero_mc_shade_fwd/bwd").  CPU tier: struct
layouts and the host-only workspace query.  GPU tier: the device-side hit / miss split against torch.nonzero, and one material training
step sequenced by the C driver against the Python-sequenced step, bit for bit (reference boundary: NeROMaterialRenderer.shade,
network/renderer.py:810-813)."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stage2_struct_layouts_match_header():
    from nero_amd import stage2 as S2
    src = ('#include <stdio.h>\n#include "nero_hip.h"\nint main(){printf("%zu %zu %zu\\n", sizeof(nero_stage2_cfg), sizeof(nero_stage2_weights), '
           'sizeof(nero_stage2_grads));}')
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, 's.c')
        open(c, 'w').write(src)
        exe = os.path.join(td, 's')
        subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), c, '-o', exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    assert sizes == [C.sizeof(S2.Cfg), C.sizeof(S2.Weights), C.sizeof(S2.Weights)], sizes


def test_stage2_workspace_query_without_a_gpu():
    from nero_amd import _lib as L
    from nero_amd import stage2 as S2
    lib = S2._lib
    hs = []
    for human in (0, 1):
        c = S2.Cfg(128, 128, human, human, 0, 5.0, 5.0, 2, 2, 2)
        h = C.c_void_p()
        L.check(lib.nero_stage2_create(C.byref(c), C.byref(h)))
        hs.append(h)
    w_bell, w_bear = lib.nero_stage2_workspace_bytes(hs[0], 8192, 4096), lib.nero_stage2_workspace_bytes(hs[1], 8192, 4096)
    # 1.05 M light rows x (3 saved layers + 3 deltas + encodings) ~ 10 GB; the human-light MLP adds its share
    assert 4e9 < w_bell < 30e9 and w_bear > w_bell, (w_bell, w_bear)
    assert lib.nero_stage2_workspace_bytes(hs[0], 1024, 512) < w_bell / 5          # (fixed part: three 272 MB partial-sum buffers)
    assert lib.nero_stage2_pack_bytes(hs[1]) > lib.nero_stage2_pack_bytes(hs[0]) > 8e6
    bad = S2.Cfg(128, 128, 0, 0, 7, 5.0, 5.0, 2, 2, 2)
    h = C.c_void_p()
    assert lib.nero_stage2_create(C.byref(bad), C.byref(h)) == -3
    for h in hs:
        lib.nero_stage2_destroy(h)


@pytest.mark.gpu
@pytest.mark.parametrize('n', [1, 63, 1024, 1025, 300001])
def test_mc_split_matches_torch_nonzero(n):
    from nero_amd import _lib as L
    g = torch.Generator().manual_seed(n)
    depth = torch.where(torch.rand(n, generator=g) < 0.3, torch.rand(n, generator=g) * 2.0, torch.full((n,), 10.0)).cuda()
    i32 = dict(dtype=torch.int32, device='cuda')
    slot, mi, hi, counts = torch.full((n,), 12345, **i32), torch.empty(n, **i32), torch.empty(n, **i32), torch.empty(2, **i32)
    tmp = torch.empty(L.lib.nero_mc_split_tmp_ints(n), **i32)
    L.check(L.lib.nero_mc_split(C.c_void_p(depth.data_ptr()), n, C.c_void_p(slot.data_ptr()), C.c_void_p(mi.data_ptr()), C.c_void_p(hi.data_ptr()),
                                C.c_void_p(counts.data_ptr()), C.c_void_p(tmp.data_ptr()), L.stream_ptr()))
    hit = depth < 10
    want_m, want_h = torch.nonzero(~hit)[:, 0].int(), torch.nonzero(hit)[:, 0].int()
    n_miss, n_hit = (int(v) for v in counts.cpu())
    assert (n_miss, n_hit) == (want_m.numel(), want_h.numel())
    assert torch.equal(mi[:n_miss], want_m) and torch.equal(hi[:n_hit], want_h)
    want_slot = torch.empty(n, **i32)
    want_slot[want_m.long()] = torch.arange(n_miss, **i32)
    want_slot[want_h.long()] = -torch.arange(n_hit, **i32) - 1
    assert torch.equal(slot, want_slot)


SCFG = dict(diffuse_sample_num=32, specular_sample_num=32, human_lights=True, outer_light_version='sphere_direction')


@pytest.mark.gpu
@pytest.mark.parametrize('scfg', [SCFG, dict(diffuse_sample_num=64, specular_sample_num=32, human_lights=False, outer_light_version='direction',
                                             geometry_type='ggx_smith')], ids=['bear', 'bell_smith'])
def test_c_driven_material_step_equals_python_driven_step_bit_for_bit(scfg, monkeypatch):
    from nero_amd.synthetic import icosphere
    from nero_amd.train import MaterialTrainStep
    v, f = icosphere(4, 0.5, 0.2)
    mesh = (v, np.ascontiguousarray(f[:, ::-1]))
    P = 160
    g = torch.Generator().manual_seed(11)
    rands = {'rand_d': torch.rand(P, 1, 1, generator=g).cuda(), 'rand_s': torch.rand(P, 1, 1, generator=g).cuda(),
             'reg_ang': torch.rand(P, 1, generator=g).cuda(), 'reg_eps': torch.normal(mean=0.0, std=0.05, size=[P, 1], generator=g).cuda()}
    res = {}
    for drv in ('py', 'c'):
        monkeypatch.setenv('NERO_STEP_DRIVER', drv)
        # (fused_glue=False: the tensor-op loss glue on both sides, so that the comparison isolates the launch sequencing; the HIP glue
        # has its own test in tests/test_material_train.py)
        ts = MaterialTrainStep({'shader_cfg': scfg, 'database_name': 'real/bear'}, mesh, points_per_rank=P, pool_points=4 * P, device='cuda:0',
                               fused_glue=False)
        assert (ts.drv is not None) == (drv == 'c')
        info = ts.forward_backward(5000, rands)
        torch.cuda.synchronize()
        res[drv] = (float(info['loss']), info['out']['rgb_pr'].detach().clone(), ts.bucket.flat.clone())
    (lp, rp, fp), (lc, rc, fc) = res['py'], res['c']
    assert torch.equal(rp, rc)
    assert lp == lc
    assert torch.equal(fp, fc), float((fp - fc).abs().max())
    assert float(fp.abs().max()) > 0


@pytest.mark.gpu
def test_c_driven_material_training_steps(monkeypatch):
    from nero_amd.synthetic import icosphere
    from nero_amd.train import MaterialTrainStep
    monkeypatch.setenv('NERO_STEP_DRIVER', 'c')
    v, f = icosphere(4, 0.5, 0.2)
    ts = MaterialTrainStep({'shader_cfg': SCFG, 'database_name': 'real/bear'}, (v, np.ascontiguousarray(f[:, ::-1])), points_per_rank=128,
                           pool_points=512, device='cuda:0')
    losses = [float(ts.step(5000 + i)['loss']) for i in range(4)]
    assert all(l == l for l in losses) and ts.drv is not None


@pytest.mark.gpu
@pytest.mark.parametrize('n', [1, 64, 1000, 1025, 300001])
def test_mc_split_with_dead_rays_matches_torch_nonzero(n):
    """nero_mc_split_dead (round 6): flagged rays enter neither list and get slot INT_MIN; the other two classes keep torch.nonzero's order"""
    from nero_amd import _lib as L
    g = torch.Generator().manual_seed(n + 7)
    depth = torch.where(torch.rand(n, generator=g) < 0.4, torch.rand(n, generator=g) * 2.0, torch.full((n,), 10.0)).cuda()
    dead = (torch.rand(n, generator=g) < 0.15).to(torch.uint8).cuda()
    i32 = dict(dtype=torch.int32, device='cuda')
    slot, mi, hi, counts = torch.full((n,), 12345, **i32), torch.empty(n, **i32), torch.empty(n, **i32), torch.empty(2, **i32)
    tmp = torch.empty(L.lib.nero_mc_split_tmp_ints(n), **i32)
    P = C.c_void_p
    L.check(L.lib.nero_mc_split_dead(P(depth.data_ptr()), P(dead.data_ptr()), n, P(slot.data_ptr()), P(mi.data_ptr()), P(hi.data_ptr()),
                                     P(counts.data_ptr()), P(tmp.data_ptr()), L.stream_ptr()))
    live = dead == 0
    hit = (depth < 10) & live
    want_m, want_h = torch.nonzero(~hit & live)[:, 0].int(), torch.nonzero(hit)[:, 0].int()
    n_miss, n_hit = (int(v) for v in counts.cpu())
    assert (n_miss, n_hit) == (want_m.numel(), want_h.numel()) and n_miss + n_hit == int(live.sum())
    assert torch.equal(mi[:n_miss], want_m) and torch.equal(hi[:n_hit], want_h)
    want_slot = torch.full((n,), -2 ** 31, **i32)
    want_slot[want_m.long()] = torch.arange(n_miss, **i32)
    want_slot[want_h.long()] = -torch.arange(n_hit, **i32) - 1
    assert torch.equal(slot, want_slot)


@pytest.mark.gpu
@pytest.mark.parametrize('drv', ['py', 'c'])
def test_skipping_the_zero_weight_rays_changes_nothing(drv, monkeypatch):
    """Round 6: GGX directions below the shading horizon carry an estimator weight of exactly zero under the Schlick geometry term
    (network/field.py:892-903, 979-987); the step neither traces nor shades them.  Against NERO_MC_SKIP_DEAD=0 (every ray traced and shaded,
    rounds 1-5): the same outputs and loss to fp32 rounding of the sums that lost zero terms -- here: bit-equal outputs -- and the same
    gradients to the rounding of weight-gradient sums whose row blocks moved (1e-5 of each tensor's largest entry); fewer MLP rows."""
    from nero_amd.synthetic import icosphere
    from nero_amd.train import MaterialTrainStep
    v, f = icosphere(4, 0.5, 0.2)
    mesh = (v, np.ascontiguousarray(f[:, ::-1]))
    scfg = dict(diffuse_sample_num=64, specular_sample_num=64, human_lights=True, outer_light_version='sphere_direction')
    P_ = 192
    g = torch.Generator().manual_seed(5)
    rands = {'rand_d': torch.rand(P_, 1, 1, generator=g).cuda(), 'rand_s': torch.rand(P_, 1, 1, generator=g).cuda(),
             'reg_ang': torch.rand(P_, 1, generator=g).cuda(), 'reg_eps': torch.normal(mean=0.0, std=0.05, size=[P_, 1], generator=g).cuda()}
    monkeypatch.setenv('NERO_STEP_DRIVER', drv)
    res = {}
    for skip in ('0', '1'):
        monkeypatch.setenv('NERO_MC_SKIP_DEAD', skip)
        ts = MaterialTrainStep({'shader_cfg': scfg, 'database_name': 'real/bear'}, mesh, points_per_rank=P_, pool_points=4 * P_, device='cuda:0',
                               fused_glue=False)
        info = ts.forward_backward(5000, rands)
        torch.cuda.synchronize()
        res[skip] = (float(info['loss']), {k: v.detach().clone() for k, v in info['out'].items() if torch.is_tensor(v)}, ts.bucket.flat.clone(),
                     [p.numel() for p in ts.bucket.params])
    (l0, o0, f0, sizes), (l1, o1, f1, _) = res['0'], res['1']
    assert abs(l0 - l1) <= 1e-6 * max(1.0, abs(l0))
    for k in o0:
        if o0[k].dtype.is_floating_point:
            assert float((o0[k] - o1[k]).abs().max()) <= 1e-6 * max(1.0, float(o0[k].abs().max())), k
    off = 0
    for n_ in sizes:
        a, b = f0[off:off + n_], f1[off:off + n_]
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()) + 1e-12
        off += n_
    assert float(f0.abs().max()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize('n', [1, 64, 1000, 1025, 300001])
def test_mc_split_classes_partitions_the_miss_list_by_the_human_mask(n):
    """nero_mc_split_classes (round 6): dead rays in neither list, hits in ray order, the miss list = the misses that reach the photographer's
    region (ray order) followed by the other misses (ray order); counts = (n_miss, n_hit, n_hum)"""
    from nero_amd import _lib as L
    g = torch.Generator().manual_seed(n + 11)
    depth = torch.where(torch.rand(n, generator=g) < 0.4, torch.rand(n, generator=g) * 2.0, torch.full((n,), 10.0)).cuda()
    dead = (torch.rand(n, generator=g) < 0.15).to(torch.uint8).cuda()
    hum = (torch.rand(n, generator=g) < 0.5).to(torch.uint8).cuda()
    i32 = dict(dtype=torch.int32, device='cuda')
    slot, mi, hi, counts = torch.full((n,), 12345, **i32), torch.empty(n, **i32), torch.empty(n, **i32), torch.empty(3, **i32)
    tmp = torch.empty(L.lib.nero_mc_split_tmp_ints(n), **i32)
    P = C.c_void_p
    L.check(L.lib.nero_mc_split_classes(P(depth.data_ptr()), P(dead.data_ptr()), P(hum.data_ptr()), n, P(slot.data_ptr()), P(mi.data_ptr()),
                                        P(hi.data_ptr()), P(counts.data_ptr()), P(tmp.data_ptr()), L.stream_ptr()))
    live = dead == 0
    hit = (depth < 10) & live
    miss = ~hit & live
    want_h = torch.nonzero(hit)[:, 0].int()
    want_m = torch.cat([torch.nonzero(miss & (hum != 0))[:, 0], torch.nonzero(miss & (hum == 0))[:, 0]]).int()
    n_miss, n_hit, n_hum = (int(v) for v in counts.cpu())
    assert (n_miss, n_hit, n_hum) == (want_m.numel(), want_h.numel(), int((miss & (hum != 0)).sum()))
    assert torch.equal(mi[:n_miss], want_m) and torch.equal(hi[:n_hit], want_h)
    want_slot = torch.full((n,), -2 ** 31, **i32)
    want_slot[want_m.long()] = torch.arange(n_miss, **i32)
    want_slot[want_h.long()] = -torch.arange(n_hit, **i32) - 1
    assert torch.equal(slot, want_slot)
