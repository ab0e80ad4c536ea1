"""GPU tier: the HIP training step under TWO ranks.  Both processes share the one device of the test box (RCCL refuses two ranks
on one GPU, so the collective runs over gloo through nero_amd.parallel's host hop); everything else -- rank-strided ray shards,
the global-count weight of the eikonal mean, the persistent flat gradient bucket, the per-rank occlusion-loss cap -- is exactly
the code path bench.py / ShapeTrainStep run under `torch.distributed.run` with the nccl backend.  The averaged gradient must
equal the single-process gradient of the 2R-ray batch."""
import os

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

CFG = {'n_samples': 16, 'n_importance': 16, 'n_bg_samples': 8, 'freeze_inv_s_step': 15000, 'perturb': 0.0, 'apply_occ_loss': True,
       'occ_loss_step': 20000}
R, STEP = 96, 25000


def _rank(rank, world, port, ret):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from nero_amd.train import ShapeTrainStep
    ts = ShapeTrainStep(CFG, rays_per_rank=R, pool_rays=4 * R, device='cuda:0', variance=0.4, rank=rank, world=world, prime_fraction=0.0)
    assert ts.net.cfg['occ_loss_max_pn'] == 2048 // world
    info = ts.forward_backward(STEP)
    ts.bucket.all_reduce_mean(world)
    torch.cuda.synchronize()
    if rank == 0:
        ret['flat'] = ts.bucket.flat.cpu()
    ret[f'n_in{rank}'] = info['n_in']
    ret[f'loss{rank}'] = float(info['loss'])
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_device_reproduce_the_big_batch_gradient():
    from nero_amd.train import ShapeTrainStep
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 33500 + os.getpid() % 2000
    mp.spawn(_rank, args=(2, port, ret), nprocs=2, join=True)
    ts = ShapeTrainStep(CFG, rays_per_rank=2 * R, pool_rays=4 * R, device='cuda:0', variance=0.4, rank=0, world=1, prime_fraction=0.0)
    info = ts.forward_backward(STEP)
    torch.cuda.synchronize()
    assert ret['n_in0'] + ret['n_in1'] == info['n_in'] and ret['n_in0'] != ret['n_in1']        # unequal shards: the count weight matters
    ref, got = ts.bucket.flat.cpu(), ret['flat']
    off = 0
    worst = 0.0
    for p in ts.bucket.params:                                # the leaves of the flat bucket (fused trainer: effective weights)
        a, b = got[off:off + p.numel()], ref[off:off + p.numel()]
        off += p.numel()
        scale = float(b.abs().max())
        if scale < 1e-12:
            continue
        worst = max(worst, float((a - b).abs().max()) / scale)
    # identical kernels on identical rows, different row tiling / reduction order between a 2R launch and two R launches
    assert worst < 1e-5, worst
    assert abs(0.5 * (ret['loss0'] + ret['loss1']) - float(info['loss'])) < 1e-6
