"""GPU tier: the Stage-II training step driver (nero_amd.train.MaterialTrainStep; NeROMaterialRenderer.train_step,
network/renderer.py:829-844 + MaterialRegLoss, network/loss.py:45-55 + Trainer.run's inner loop, train/trainer.py:120-140):
the fused trainer loop against the torch path, and two data-parallel ranks against the single-process big batch (SURVEY.md 8e,
BASELINE configs[4])."""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

SCFG = dict(diffuse_sample_num=32, specular_sample_num=32, human_lights=True, outer_light_version='sphere_direction')
P = 96


def _mesh():
    from nero_amd.synthetic import icosphere
    v, f = icosphere(4, 0.5, 0.2)
    return v, np.ascontiguousarray(f[:, ::-1])


def _rands(n, lo, hi, dev):
    """per-point random draws of the GLOBAL batch (rows lo:hi of a fixed table), so that ranks and the big batch see the same ones"""
    g = torch.Generator().manual_seed(11)
    t = {'rand_d': torch.rand(n, 1, 1, generator=g), 'rand_s': torch.rand(n, 1, 1, generator=g), 'reg_ang': torch.rand(n, 1, generator=g),
         'reg_eps': torch.normal(mean=0.0, std=0.05, size=[n, 1], generator=g)}
    return {k: v[lo:hi].to(dev) for k, v in t.items()}


def test_fused_material_step_matches_the_torch_path():
    """three optimisation steps of the bear material model on both trainer loops (same criterion as the Stage-I test)"""
    from nero_amd.train import MaterialTrainStep
    runs = {}
    for fused in (False, True):
        ts = MaterialTrainStep({'shader_cfg': SCFG, 'database_name': 'real/bear'}, _mesh(), points_per_rank=P, pool_points=4 * P, device='cuda:0',
                               fused=fused)
        p0 = {k: v.detach().clone() for k, v in ts.net.state_dict().items()}
        losses = []
        for i in range(3):
            lr = 1e-4
            info = ts.forward_backward(5000 + i, _rands(P, 0, P, 'cuda:0'))
            losses.append(float(info['loss']))
            if fused:
                ts.fopt.step(lr, 1)
            else:
                for g in ts.opt.param_groups:
                    g['lr'] = lr
                ts.opt.step()
        torch.cuda.synchronize()
        runs[fused] = (p0, {k: v.detach().clone() for k, v in ts.net.state_dict().items()}, losses)
    (p0, pr, lr_), (q0, pf, lf) = runs[False], runs[True]
    assert all(torch.equal(p0[k], q0[k]) for k in p0)
    assert abs(lr_[0] - lf[0]) < 1e-6 and np.allclose(lr_, lf, rtol=0, atol=1e-3), (lr_, lf)
    n = 0
    for k in pr:
        if k.endswith('light_pts'):
            continue
        upd = float((pr[k] - p0[k]).abs().max())
        d = (pf[k] - pr[k]).abs()
        assert float(d.mean()) <= 5e-3 * upd + 1e-9, (k, float(d.mean()), upd)
        assert float((d > 0.02 * upd + 1e-9).float().mean()) <= 0.02, (k, float((d > 0.02 * upd).float().mean()))
        n += 1
    assert n == 96                                                  # every tensor of the bear material model


def test_no_grad_cache_follows_the_fused_material_optimiser():
    """FusedMaterialOptimizer updates the parameters with raw kernels, which torch's version counters do not see; the
    renderer's no-grad cache of packed operand images must be dropped by the optimiser (a material extraction after fused steps
    must see the new weights)."""
    from nero_amd.train import MaterialTrainStep
    ts = MaterialTrainStep({'shader_cfg': SCFG, 'database_name': 'real/bear'}, _mesh(), points_per_rank=P, pool_points=4 * P, device='cuda:0',
                           fused=True)
    g = torch.Generator().manual_seed(5)
    pts = (torch.nn.functional.normalize(torch.randn(500, 3, generator=g), dim=-1) * 0.45).cuda()
    with torch.no_grad():
        k0 = ts.net._kernels()
        before = [t.clone() for t in ts.net.predict_materials(pts, k0)]
        assert ts.net._kernels()[2] is k0[2]                         # unchanged weights: the cached pack is served
    for i in range(2):
        ts.forward_backward(5000 + i, _rands(P, 0, P, 'cuda:0'))
        ts.fopt.step(1e-2, 1)
    assert ts.net._kern_cache is None                                # dropped by FusedMaterialOptimizer._after_step
    with torch.no_grad():
        k1 = ts.net._kernels()
        assert k1[2] is not k0[2]
        after = ts.net.predict_materials(pts, k1)
    assert max(float((a - b).abs().max()) for a, b in zip(after, before)) > 1e-5


BELL_SCFG = dict(diffuse_sample_num=64, specular_sample_num=32, human_lights=False, outer_light_version='direction', geometry_type='ggx_smith')


@pytest.mark.parametrize('step', [5000, 500], ids=['step5000', 'step500_hinge'])
@pytest.mark.parametrize('kind', ['bear', 'bell_smith', 'bell_l1_constant'])
def test_hip_loss_glue_matches_the_tensor_glue(kind, step):
    """nero_amd/csrc/mat_loss.hip (perturbed points, sigmoid heads, loss_rgb + loss_mat_reg + loss_diffuse_light and their gradients as five launches)
    against the tensor-op glue of NeROMaterialRenderer.shade_train, which the oracle tests pin: same pool, same random draws, same
    C-level shading calls.  Loss and colours to 1e-6; every gradient tensor to 1e-5 of its own maximum.  The heads must reproduce the
    tensor ops BIT FOR BIT: one ulp of roughness (a contracted fma in the affine map) moves the specular sample directions and with them
    the light-MLP gradients by 1e-3 of their maximum -- which is how the first version of the kernel failed this test."""
    from nero_amd.train import MaterialTrainStep
    cfg = {'shader_cfg': SCFG if kind == 'bear' else BELL_SCFG, 'database_name': 'real/bear'}
    if kind == 'bell_l1_constant':
        cfg = {'shader_cfg': {**BELL_SCFG, 'change_type': 'constant', 'change_eps': 0.03}, 'database_name': 'syn/bell', 'rgb_loss': 'l1'}
    Pn = 192
    res = {}
    for glue in (False, True):
        ts = MaterialTrainStep(cfg, _mesh(), points_per_rank=Pn, pool_points=4 * Pn, device='cuda:0', fused_glue=glue)
        assert ts.drv is not None
        info = ts.forward_backward(step, _rands(Pn, 0, Pn, 'cuda:0'))
        torch.cuda.synchronize()
        res[glue] = (float(info['loss']), info['out']['rgb_pr'].detach().clone(), info['out']['roughness'].detach().clone(), ts.bucket.flat.clone(), ts)
        if glue:
            terms = info['loss_terms'].cpu()
            assert abs(float(terms[1] + terms[2] + terms[3]) - float(terms[0])) < 1e-6
    (lt, rt, qt, ft, ts), (lh, rh, qh, fh, _) = res[False], res[True]
    assert abs(lt - lh) <= 2e-6 * max(1.0, abs(lt)), (lt, lh)
    assert float((rt - rh).abs().max()) < 1e-6 and float((qt - qh).abs().max()) < 1e-6
    off, worst, n = 0, 0.0, 0
    for p in ts.bucket.params:
        a, b = fh[off:off + p.numel()], ft[off:off + p.numel()]
        off += p.numel()
        scale = float(b.abs().max())
        if scale < 1e-12:
            assert float(a.abs().max()) < 1e-12
            continue
        worst = max(worst, float((a - b).abs().max()) / scale)
        n += 1
    assert n >= 50 and worst < 1e-5, (n, worst)


def test_hip_reg_points_match_the_tensor_glue():
    from nero_amd import stage2 as S2
    from nero_amd.renderer import NeROMaterialRenderer
    net = NeROMaterialRenderer({'shader_cfg': BELL_SCFG, 'database_name': 'syn/bell'}, mesh=_mesh()).cuda()
    g = torch.Generator().manual_seed(3)
    n = 1000
    pts = (torch.rand(n, 3, generator=g) - 0.5).cuda()
    nrm = torch.randn(n, 3, generator=g).cuda()
    nrm[:3] = torch.tensor([[0.0, 0.0, 1.0], [1.0, 0.0, 0.0], [0.0, -2.0, 0.0]]).cuda()            # axis-aligned normals: both frame branches
    ang, eps = torch.rand(n, 1, generator=g).cuda(), (torch.randn(n, 1, generator=g) * 0.05).cuda()
    want = net.regularization_points(pts, nrm, ang, eps)
    got = S2.reg_points(pts, nrm, ang, eps)
    assert torch.equal(got[:n], pts)
    assert float((got[n:] - want).abs().max()) < 2e-7


def _rank(rank, world, port, ret, step, backend='gloo', one_device=True):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dev = 'cuda:0' if one_device else f'cuda:{rank}'
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    from nero_amd.train import MaterialTrainStep
    ts = MaterialTrainStep({'shader_cfg': SCFG, 'database_name': 'real/bear'}, _mesh(), points_per_rank=P, pool_points=4 * P, device=dev,
                           rank=rank, world=world)
    info = ts.forward_backward(step, _rands(world * P, rank * P, (rank + 1) * P, dev))
    ts.bucket.all_reduce_mean(world)
    torch.cuda.synchronize()
    if rank == 0:
        ret['flat'] = ts.bucket.flat.cpu()
    ret[f'loss{rank}'] = float(info['loss'])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('step', [5000, 500], ids=['step5000', 'step500_hinge'])
def test_two_ranks_reproduce_the_big_batch_gradient(step):
    """Both processes share the test box's one device (RCCL refuses duplicate GPUs: gloo with nero_amd.parallel's host hop);
    rank-strided point shards, replicated BVH, ONE flat all-reduce.  step 500: the reg_min_max hinge is a SUM over the batch
    (network/field.py:1079-1084) and needs its `world` weight."""
    _two_rank_material(step, 'gloo', True)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='RCCL needs one device per rank: a box with >= 2 GPUs')
def test_rccl_one_rank_per_device_material_step_reproduces_the_big_batch_gradient():
    """Stage II over backend `nccl` (= RCCL), rank r on cuda:r (bench.py --stage 2 --gpus N): skipped on the one-GPU test box"""
    _two_rank_material(5000, 'nccl', False)


def _two_rank_material(step, backend, one_device):
    from nero_amd.train import MaterialTrainStep
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 35500 + os.getpid() % 2000 + (11 if backend == 'nccl' else 0)
    mp.spawn(_rank, args=(2, port, ret, step, backend, one_device), nprocs=2, join=True)
    ts = MaterialTrainStep({'shader_cfg': SCFG, 'database_name': 'real/bear'}, _mesh(), points_per_rank=2 * P, pool_points=4 * P, device='cuda:0')
    info = ts.forward_backward(step, _rands(2 * P, 0, 2 * P, 'cuda:0'))
    torch.cuda.synchronize()
    ref, got = ts.bucket.flat.cpu(), ret['flat']
    off, worst, n = 0, 0.0, 0
    for p in ts.bucket.params:
        a, b = got[off:off + p.numel()], ref[off:off + p.numel()]
        off += p.numel()
        scale = float(b.abs().max())
        if scale < 1e-12:
            continue
        worst = max(worst, float((a - b).abs().max()) / scale)
        n += 1
    assert n >= 60 and worst < 1e-5, (n, worst)
    # the rank losses carry the world-weighted hinge: their mean exceeds the big-batch loss by exactly (world - 1) x mean hinge share,
    # which is zero at step >= 2000
    if step >= 2000:
        assert abs(0.5 * (ret['loss0'] + ret['loss1']) - float(info['loss'])) < 1e-6
