"""GPU tier: run-to-run bit reproducibility.  No kernel of the path uses floating-point atomics and every reduction has a fixed order,
so the same batch with the same random draws must give the same loss and the same gradients BIT FOR BIT, however the workgroups are
scheduled.  This is the test that found the fault of the two-workgroups-per-CU forward kernel in round 3 (one launch in three returned one column
of 16 rows with another partial sum -- 1e-7 of a gradient, below every parity tolerance, invisible to a comparison of single runs) and
the same fault in `sdf_alpha_bwd` in round 5, where it was traced to packed fp32 arithmetic beside another wave's MFMAs.
With no packed fp32 left in the library the two-workgroup kernels are back and these tests run on them (tests/test_paired_engine.py
compares them with the 512-thread kernels bit for bit)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BELL = {'freeze_inv_s_step': 15000, 'apply_occ_loss': True, 'occ_loss_step': 20000}


@pytest.mark.parametrize('kind,rays', [('bell', 2048), ('bear', 512)])
def test_stage1_training_step_is_bit_reproducible(kind, rays):
    from nero_amd.train import ShapeTrainStep
    cfg = dict(BELL) if kind == 'bell' else {**BELL, 'shader_config': {'human_light': True}}
    ts = ShapeTrainStep(cfg, rays_per_rank=rays, pool_rays=4 * rays, device='cuda:0', variance=0.5, prime_fraction=0.0, prime_passes=0)
    ref = None
    for k in range(6):
        ts.cursor = 0
        torch.manual_seed(1234)                               # the sampler's perturbation draws, the occlusion-loss keys
        info = ts.forward_backward(25000)
        torch.cuda.synchronize()
        cur = (float(info['loss']), info['n_in'], ts.bucket.flat.clone())
        if ref is None:
            ref = cur
            assert float(ref[2].abs().max()) > 0
            continue
        assert cur[0] == ref[0] and cur[1] == ref[1], (k, cur[0], ref[0])
        assert torch.equal(cur[2], ref[2]), (k, int((cur[2] != ref[2]).sum()), float((cur[2] - ref[2]).abs().max()))


def test_stage2_training_step_is_bit_reproducible():
    from nero_amd.synthetic import icosphere
    from nero_amd.train import MaterialTrainStep
    v, f = icosphere(5, 0.5, 0.2)
    P = 1024
    ts = MaterialTrainStep({'shader_cfg': dict(diffuse_sample_num=64, specular_sample_num=64, human_lights=True, outer_light_version='sphere_direction'),
                            'database_name': 'real/bear'}, (v, np.ascontiguousarray(f[:, ::-1])), points_per_rank=P, pool_points=2 * P, device='cuda:0')
    g = torch.Generator().manual_seed(11)
    rands = {'rand_d': torch.rand(P, 1, 1, generator=g).cuda(), 'rand_s': torch.rand(P, 1, 1, generator=g).cuda(),
             'reg_ang': torch.rand(P, 1, generator=g).cuda(), 'reg_eps': torch.normal(mean=0.0, std=0.05, size=[P, 1], generator=g).cuda()}
    ref = None
    for k in range(6):
        ts.cursor = 0
        info = ts.forward_backward(5000, rands)
        torch.cuda.synchronize()
        cur = (float(info['loss']), ts.bucket.flat.clone())
        if ref is None:
            ref = cur
            continue
        assert cur[0] == ref[0], (k, cur[0], ref[0])
        assert torch.equal(cur[1], ref[1]), (k, int((cur[1] != ref[1]).sum()))


def test_forward_chain_launches_are_bit_reproducible():
    """a NeRF++-head shaped chain (256 -> 256 -> 128 with a 27-column aux operand -> 3) over 300 k rows, forty launches on the default
    forward engine (at this size the two-workgroups-per-CU kernel): every saved activation and the head identical to the first launch
    (the shape and size at which the round-3 build of that kernel failed in a third of its launches)"""
    from nero_amd import _lib as L
    from nero_amd import chain as CH
    from nero_amd.chain import Chain, Dense, Head
    assert CH.GEMM_MODE['fwd'] == L.GEMM_F16X3, 'the default forward arithmetic is f16x3 (nero_amd/chain.py)'
    g = torch.Generator(device='cuda').manual_seed(2)
    rn = lambda *s: torch.randn(*s, device='cuda', generator=g)
    n = 300000
    rp = (n + 63) // 64 * 64
    W0, b0 = rn(256, 256) / 16, rn(256) * 0.1
    W1, b1 = rn(128, 283) / 16, rn(128) * 0.1
    Wh, bh = rn(3, 128) / 8, rn(3) * 0.1
    ch = Chain([(Dense(W0, b0, L.ACT_NONE, 256), None), (Dense(W1, b1, L.ACT_RELU, 256, 0, 27, 256), None), (None, Head(Wh, bh))],
               k_init=256, k_aux=32).pack()
    init, aux = rn(rp, 256), rn(rp, 32)
    ref = None
    for k in range(40):
        o = ch.forward(init, aux, n, save=True)
        cur = (o['saves'][0][:n].clone(), o['saves'][1][:n, :128].clone(), o['heads'][2][:n, :3].clone())
        if ref is None:
            ref = cur
            continue
        for a, b in zip(cur, ref):
            assert torch.equal(a, b), (k, int((a != b).sum()), float((a - b).abs().max()))
