"""GPU tier: the row-owner forward kernel (nero_amd/csrc/mlp_f16r.hip, nero_f16_rowowner: a wave owns 32 rows and all features, activation
planes in registers, weights through an LDS-DMA ring, the epilogue of a feature tile under the next tile's MFMAs) against the 512-thread
kernel of mlp_f16x3.hip: same packed operands, same descriptors -- identical results BIT FOR BIT, for every launch size (1 row ... several
groups per CU), skip / aux inputs of both widths, heads, ragged last groups; and identical from launch to launch."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def rowowner():
    from nero_amd import chain as CH
    prev, prev_p = CH.f16_rowowner(), CH.f16_paired()
    yield CH.f16_rowowner
    CH.f16_rowowner(prev)
    CH.f16_paired(prev_p)


def _sdf_chain():
    """the SDF network's value-only chain of every shipped YAML: PE-6 input (39 of 40 columns), 8 x 256 softplus layers, the input re-injected
    in front of layer 4 (217 + 39 columns, 1/sqrt(2)), a one-output head -- what the sampler and the occlusion march evaluate"""
    from nero_amd.sdf import SDFField
    g = torch.Generator().manual_seed(4)
    dims = [39] + [256] * 8 + [257]
    eff = []
    for l in range(9):
        n_out = dims[l + 1] - (39 if l + 1 == 4 else 0)
        eff.append(((torch.randn(n_out, dims[l], generator=g) * 1.2 / math.sqrt(dims[l])).cuda(), (torch.randn(n_out, generator=g) * 0.05).cuda()))
    return SDFField(eff).pack()


@pytest.mark.parametrize('n', [1, 100, 129, 4096, 70001])
def test_sdf_value_chain_is_bit_identical(rowowner, n):
    from nero_amd import chain as CH
    from nero_amd.chain import row_pad
    f = _sdf_chain()
    g = torch.Generator(device='cuda').manual_seed(n)
    pe = torch.zeros(row_pad(n), 40, device='cuda')
    pe[:n, :39] = torch.randn(n, 39, device='cuda', generator=g)
    CH.f16_paired(0)
    rowowner(0)
    ref = f.sdf_from_pe(pe, n)[:n].clone()
    rowowner(3)
    for k in range(3):
        cur = f.sdf_from_pe(pe, n)[:n].clone()
        assert torch.equal(cur[:, 0], ref[:, 0]), (n, k, int((cur[:, 0] != ref[:, 0]).sum()), float((cur[:, 0] - ref[:, 0]).abs().max()))
    assert float(ref[:, 0].abs().max()) > 0


def _generic_chain(k_aux, act_mid):
    from nero_amd import _lib as L
    from nero_amd.chain import Chain, Dense, Head
    g = torch.Generator(device='cuda').manual_seed(2 + k_aux)
    rn = lambda *s: torch.randn(*s, device='cuda', generator=g)
    W0, b0 = rn(256, 128) / 11, rn(256) * 0.1
    W1, b1 = rn(224, 256 + k_aux) / 16, rn(224) * 0.1
    W2, b2 = rn(128, 224) / 15, rn(128) * 0.1
    Wh, bh = rn(3, 128) / 8, rn(3) * 0.1
    ch = Chain([(Dense(W0, b0, L.ACT_RELU, 128), None), (Dense(W1, b1, act_mid, 256, 0, k_aux, 256), None), (Dense(W2, b2, L.ACT_NONE, 224), None),
                (None, Head(Wh, bh))], k_init=128, k_aux=(k_aux + 7) // 8 * 8, aux_wide=k_aux > 40).pack()
    return ch, rn


@pytest.mark.parametrize('k_aux,n', [(27, 1000), (27, 33000), (88, 1000), (88, 40000)])
def test_generic_chain_with_aux_and_head_is_bit_identical(rowowner, k_aux, n):
    from nero_amd import _lib as L
    from nero_amd import chain as CH
    ch, rn = _generic_chain(k_aux, L.ACT_SOFTPLUS100 if k_aux == 27 else L.ACT_RELU)
    rp = (n + 63) // 64 * 64
    init, aux = rn(rp, 128), rn(rp, (k_aux + 7) // 8 * 8)
    CH.f16_paired(0)
    rowowner(0)
    ref = ch.forward(init, aux, n, save=False)['heads'][3][:n, :3].clone()
    rowowner(3)
    cur = ch.forward(init, aux, n, save=False)['heads'][3][:n, :3].clone()
    assert torch.equal(cur, ref), (int((cur != ref).sum()), float((cur - ref).abs().max()))
    # a saving launch keeps the 512-thread / paired kernels (the row-owner kernel writes no activations): same answer either way
    sv = ch.forward(init, aux, n, save=True)
    assert torch.equal(sv['heads'][3][:n, :3], ref)


def test_sampler_and_render_are_unchanged_with_the_row_owner_kernel(rowowner):
    """the whole Stage-I render of a golden case with the sampler's SDF evaluations on the row-owner kernel: z_vals, ray_rgb and the loss
    gradient bit for bit those of the default organisation"""
    from tests.helpers import T, build_case_model, load_golden
    z, meta = load_golden('bell_s25000')
    net = build_case_model(meta).cuda()
    cu = lambda k: T(z, k, 'cuda')

    def run():
        net.zero_grad(set_to_none=True)
        out = net.render(cu('o'), cu('d'), cu('near'), cu('far'), cu('human_poses'), -1, meta['anneal'], is_train=True, step=meta['step'],
                         rand1=cu('rand1'), rand_bg=cu('rand_bg'))
        loss = net.compute_rgb_loss(out['ray_rgb'], cu('gt')).mean() + (out['gradient_error'] * 0.1).mean()
        loss.backward()
        return out['ray_rgb'].detach().clone(), torch.cat([p.grad.reshape(-1) for p in net.parameters() if p.grad is not None]).clone()
    rowowner(0)
    a = run()
    rowowner(3)
    b = run()
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
