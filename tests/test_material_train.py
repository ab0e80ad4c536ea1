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


def _rank(rank, world, port, ret, step):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from nero_amd.train import MaterialTrainStep
    ts = MaterialTrainStep({'shader_cfg': SCFG, 'database_name': 'real/bear'}, _mesh(), points_per_rank=P, pool_points=4 * P, device='cuda:0',
                           rank=rank, world=world)
    info = ts.forward_backward(step, _rands(world * P, rank * P, (rank + 1) * P, 'cuda:0'))
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
    from nero_amd.train import MaterialTrainStep
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 35500 + os.getpid() % 2000
    mp.spawn(_rank, args=(2, port, ret, step), nprocs=2, join=True)
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
