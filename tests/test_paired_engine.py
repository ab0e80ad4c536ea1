"""GPU tier: the two-workgroups-per-CU organisation of the chain passes (nero_amd/csrc/mlp_f16p.hip, nero_f16_paired) against the
512-thread kernels (mlp_f16x3.hip): same operands, same descriptors, identical results BIT FOR BIT, and identical from launch to launch
at the size and shape at which the round-2 version of this engine failed in a third of its launches (the fault traced in round 5 to packed
fp32 arithmetic beside another wave's MFMAs: DESIGN.md 9.3; the library holds no packed fp32 now, tests/test_no_packed_fp32.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def paired():
    from nero_amd import chain as CH
    prev = CH.f16_paired()
    yield CH.f16_paired
    CH.f16_paired(prev)


def _head_chain(n):
    from nero_amd import _lib as L
    from nero_amd.chain import Chain, Dense, Head
    g = torch.Generator(device='cuda').manual_seed(2)
    rn = lambda *s: torch.randn(*s, device='cuda', generator=g)
    rp = (n + 63) // 64 * 64
    W0, b0 = rn(256, 256) / 16, rn(256) * 0.1
    W1, b1 = rn(128, 283) / 16, rn(128) * 0.1
    Wh, bh = rn(3, 128) / 8, rn(3) * 0.1
    ch = Chain([(Dense(W0, b0, L.ACT_NONE, 256), None), (Dense(W1, b1, L.ACT_RELU, 256, 0, 27, 256), None), (None, Head(Wh, bh))],
               k_init=256, k_aux=32).pack()
    return ch, rn(rp, 256), rn(rp, 32)


def _fwd(ch, init, aux, n):
    o = ch.forward(init, aux, n, save=True)
    return o['saves'][0][:n].clone(), o['saves'][1][:n, :128].clone(), o['heads'][2][:n, :3].clone()


@pytest.mark.parametrize('n', [1000, 300000])
def test_paired_forward_equals_the_wide_kernel_and_itself(paired, n):
    ch, init, aux = _head_chain(n)
    paired(0)
    ref = _fwd(ch, init, aux, n)
    paired(8 | 1)
    for k in range(40 if n > 100000 else 3):
        cur = _fwd(ch, init, aux, n)
        for a, b in zip(cur, ref):
            assert torch.equal(a, b), (k, int((a != b).sum()), float((a - b).abs().max()))


BELL = {'freeze_inv_s_step': 15000, 'apply_occ_loss': True, 'occ_loss_step': 20000}


@pytest.mark.parametrize('kind,rays', [('bell', 2048), ('bear', 512)])
def test_stage1_step_is_identical_under_every_organisation(paired, kind, rays):
    """the whole Stage-I step (forward, tangent and reverse chains of the SDF, colour, NeRF++ networks) under mask 0, 1, 2, 4, 7 (bit 3: whatever the launch size):
    the same loss and gradient bucket bit for bit; mask 7 six times over"""
    from nero_amd.train import ShapeTrainStep
    cfg = dict(BELL) if kind == 'bell' else {**BELL, 'shader_config': {'human_light': True}}
    ts = ShapeTrainStep(cfg, rays_per_rank=rays, pool_rays=4 * rays, device='cuda:0', variance=0.5, prime_fraction=0.0, prime_passes=0)

    def step():
        ts.cursor = 0
        torch.manual_seed(1234)
        info = ts.forward_backward(25000)
        torch.cuda.synchronize()
        return float(info['loss']), info['n_in'], ts.bucket.flat.clone()

    paired(0)
    ref = step()
    assert float(ref[2].abs().max()) > 0
    for mask in (1, 2, 4, 7, 7, 7, 7, 7, 7):
        paired(8 | mask)
        cur = step()
        assert cur[0] == ref[0] and cur[1] == ref[1], (mask, cur[0], ref[0])
        assert torch.equal(cur[2], ref[2]), (mask, int((cur[2] != ref[2]).sum()), float((cur[2] - ref[2]).abs().max()))


def test_stage2_step_is_identical_under_every_organisation(paired):
    import numpy as np
    from nero_amd.synthetic import icosphere
    from nero_amd.train import MaterialTrainStep
    v, f = icosphere(5, 0.5, 0.2)
    P = 1024
    ts = MaterialTrainStep({'shader_cfg': dict(diffuse_sample_num=64, specular_sample_num=64, human_lights=True, outer_light_version='sphere_direction'),
                            'database_name': 'real/bear'}, (v, np.ascontiguousarray(f[:, ::-1])), points_per_rank=P, pool_points=2 * P, device='cuda:0')
    g = torch.Generator().manual_seed(11)
    rands = {'rand_d': torch.rand(P, 1, 1, generator=g).cuda(), 'rand_s': torch.rand(P, 1, 1, generator=g).cuda(),
             'reg_ang': torch.rand(P, 1, generator=g).cuda(), 'reg_eps': torch.normal(mean=0.0, std=0.05, size=[P, 1], generator=g).cuda()}

    def step():
        ts.cursor = 0
        info = ts.forward_backward(5000, rands)
        torch.cuda.synchronize()
        return float(info['loss']), ts.bucket.flat.clone()

    paired(0)
    ref = step()
    for mask in (1, 4, 5, 5, 5):
        paired(8 | mask)
        cur = step()
        assert cur[0] == ref[0], (mask, cur[0], ref[0])
        assert torch.equal(cur[1], ref[1]), (mask, int((cur[1] != ref[1]).sum()))
