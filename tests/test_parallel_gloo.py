"""CPU tier, world_size 2 over gloo: the N>1 path (rank-strided ray shards + one flat gradient all-reduce) reproduces the
single-process big-batch gradient.  The local compute is the CPU oracle (tests may use it); the product's GPU step uses the
same two helpers from nero_amd/parallel.py."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _small_net():
    from nero_amd.renderer import NeROShapeRenderer
    from nero_amd.synthetic import perturb_state
    torch.manual_seed(6033)
    net = NeROShapeRenderer(dict(n_samples=8, n_importance=8, n_bg_samples=4, up_sample_steps=2, apply_occ_loss=False), training=False)
    perturb_state(net, 0.4)
    return net


def _grads(net, o, d, gt, weight_fn=None):
    from oracle import nero_oracle as O
    sd = {k: v for k, v in net.named_parameters()}
    sd.update({k: v for k, v in net.named_buffers()})
    P = O.effective_params(sd)
    cfg = {**O.DEFAULT_CFG, **net.cfg}
    near, far = O.near_far_from_sphere(o, d)
    out = O.render(P, cfg, o, d, near, far, torch.zeros(o.shape[0], 3, 4), 5000, 0.1)
    loss = O.rgb_loss(cfg, out['ray_rgb'], gt).mean()          # ray-level mean: exact under equal shards
    if weight_fn is not None:                                  # sample-level mean (eikonal): needs the global-count weight
        loss = loss + (out['gradient_error'] * 0.1).mean() * weight_fn(out['n_inner'])
    for p in net.parameters():
        p.grad = None
    loss.backward()
    return [p for p in net.parameters()]


def _worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    from nero_amd.parallel import allreduce_mean_grads, global_count_weight, rank_slice
    from nero_amd.synthetic import synthetic_rays
    R = 6
    o, d, _, gt = synthetic_rays(world * R, seed=1, window=200)
    s = rank_slice(0, R, rank)
    net = _small_net()
    params = _grads(net, o[s], d[s], gt[s], lambda n: global_count_weight(n, world, 'cpu'))
    allreduce_mean_grads(params, world)
    if rank == 0:
        ret['dp'] = [p.grad.clone() if p.grad is not None else None for p in params]
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allreduce_matches_big_batch():
    from nero_amd.parallel import rank_slice
    from nero_amd.synthetic import synthetic_rays
    assert rank_slice(12, 6, 1) == slice(18, 24)
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    o, d, _, gt = synthetic_rays(12, seed=1, window=200)
    torch.set_num_threads(4)
    net = _small_net()
    params = _grads(net, o, d, gt, lambda n: 1.0)
    n = 0
    for p, g in zip(params, ret['dp']):
        ref = p.grad if p.grad is not None else torch.zeros_like(p)
        g = g if g is not None else torch.zeros_like(p)
        scale = float(ref.abs().max())
        if scale < 1e-12:
            continue
        assert float((g - ref).abs().max()) / scale < 2e-4
        n += 1
    assert n > 50


def _bucket_worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import nero_oracle as O
    from nero_amd.parallel import GradBucket, global_count_weight, rank_slice
    from nero_amd.synthetic import synthetic_rays
    R = 6
    o, d, _, gt = synthetic_rays(world * R, seed=1, window=200)
    s = rank_slice(0, R, rank)
    net = _small_net()
    bucket = GradBucket(net.parameters())
    views = [p.grad for p in net.parameters()]
    for it in range(2):                                        # two steps: the views must survive zero() / backward / all-reduce
        bucket.zero()
        sd = {k: v for k, v in net.named_parameters()}
        sd.update({k: v for k, v in net.named_buffers()})
        cfg = {**O.DEFAULT_CFG, **net.cfg}
        near, far = O.near_far_from_sphere(o[s], d[s])
        out = O.render(O.effective_params(sd), cfg, o[s], d[s], near, far, torch.zeros(R, 3, 4), 5000, 0.1)
        w = global_count_weight(out['n_inner'], world, 'cpu')
        (O.rgb_loss(cfg, out['ray_rgb'], gt[s]).mean() + (out['gradient_error'] * 0.1).mean() * w).backward()
        bucket.all_reduce_mean(world)
    assert all(p.grad is v for p, v in zip(net.parameters(), views))
    assert all(p.grad.data_ptr() >= bucket.flat.data_ptr() for p in net.parameters())
    if rank == 0:
        ret['dp'] = [p.grad.clone() for p in net.parameters()]
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_persistent_grad_bucket_matches_big_batch():
    from nero_amd.parallel import per_rank_occ_cap
    from nero_amd.synthetic import synthetic_rays
    assert per_rank_occ_cap(2048, 8) == 256 and per_rank_occ_cap(2048, 1) == 2048 and per_rank_occ_cap(3, 8) == 1
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 31500 + os.getpid() % 2000
    mp.spawn(_bucket_worker, args=(2, port, ret), nprocs=2, join=True)
    o, d, _, gt = synthetic_rays(12, seed=1, window=200)
    torch.set_num_threads(4)
    net = _small_net()
    params = _grads(net, o, d, gt, lambda n: 1.0)
    n = 0
    for p, g in zip(params, ret['dp']):
        ref = p.grad if p.grad is not None else torch.zeros_like(p)
        scale = float(ref.abs().max())
        if scale < 1e-12:
            continue
        assert float((g - ref).abs().max()) / scale < 2e-4
        n += 1
    assert n > 50


# ----------------------------------------------------------------------------------------------------------------------
# Stage II (material) data parallelism: point shards, replicated tracer, the `world` weight of the summed hinge
# ----------------------------------------------------------------------------------------------------------------------
MAT_SCFG = dict(diffuse_sample_num=8, specular_sample_num=8, human_lights=True, outer_light_version='sphere_direction')


def _mat_inputs(n):
    from tests.helpers import golden_mesh
    from oracle.tracer_oracle import trace_bruteforce
    from nero_amd.synthetic import synthetic_rays
    v, f = golden_mesh()
    o, d, poses, gt = synthetic_rays(8 * n, seed=5, window=120)
    pos, nrm, depth, tri = trace_bruteforce(v, f, o.numpy(), d.numpy())
    sel = [i for i in range(o.shape[0]) if tri[i] >= 0][:n]
    assert len(sel) == n
    t = lambda a: torch.from_numpy(a[sel]).float()
    g = torch.Generator().manual_seed(3)
    return dict(pts=t(pos), view=-d[sel], normals=torch.nn.functional.normalize(-t(nrm), dim=-1), poses=poses[sel], gt=gt[sel],
                rand_d=torch.rand(n, 1, 1, generator=g), rand_s=torch.rand(n, 1, 1, generator=g), reg_ang=torch.rand(n, 1, generator=g),
                reg_eps=torch.normal(mean=0.0, std=0.05, size=[n, 1], generator=g))


def _mat_grads(I, s, step, world):
    from oracle import nero_oracle as O, nero_oracle_mat as M
    from tests.helpers import MatHolder, oracle_trace_fn
    from nero_amd.synthetic import perturb_state
    from nero_amd.train import material_training_loss
    from nero_amd.renderer import NeROShapeRenderer
    torch.manual_seed(6033)
    net = MatHolder(MAT_SCFG)
    perturb_state(net, None)
    with torch.no_grad():                          # push some points into the sigmoid's saturation so that the summed hinge is live
        net.shader_network.roughness_predictor[6].bias.add_(6.0)
    sd = {k: v for k, v in net.named_parameters()}
    sd.update({k: v for k, v in net.named_buffers()})
    hp = NeROShapeRenderer.get_human_coordinate_poses(type('c', (), {'cfg': {'fixed_camera': False}})(), I['poses'])
    out = M.material_train_outputs(O.effective_params(sd), {'shader_cfg': MAT_SCFG}, oracle_trace_fn(), I['pts'][s], I['view'][s], I['normals'][s],
                                   hp[s], I['gt'][s], step, I['rand_d'][s], I['rand_s'][s], I['reg_ang'][s], I['reg_eps'][s])
    scfg = {**M.DEFAULT_SHADER_CFG, **MAT_SCFG}
    loss = material_training_loss(scfg, out, step, world)          # the PRODUCT's loss assembly incl. the data-parallel hinge weight
    loss.backward()
    return [p for p in net.parameters()], out


def _mat_worker(rank, world, port, ret, step):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    from nero_amd.parallel import allreduce_mean_grads, rank_slice
    n = 6
    I = _mat_inputs(world * n)
    params, _ = _mat_grads(I, rank_slice(0, n, rank), step, world)
    allreduce_mean_grads(params, world)
    if rank == 0:
        ret['dp'] = [p.grad.clone() if p.grad is not None else None for p in params]
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('step', [500, 5000])
def test_two_rank_material_step_matches_big_batch(step):
    """Stage II (BASELINE configs[4]): rank-strided point shards + one flat all-reduce reproduce the single-process gradient of the
    2P-point batch -- at step 500 only with the `world` weight on the reg_min_max hinge, which the reference SUMS over the batch
    (network/field.py:1079-1084) while every other term is a mean"""
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 37500 + os.getpid() % 2000 + step % 7
    mp.spawn(_mat_worker, args=(2, port, ret, step), nprocs=2, join=True)
    torch.set_num_threads(4)
    I = _mat_inputs(12)
    params, out = _mat_grads(I, slice(0, 12), step, 1)
    if step < 2000:
        from nero_amd.renderer import material_hinge
        from oracle import nero_oracle_mat as M
        assert float(material_hinge({**M.DEFAULT_SHADER_CFG, **MAT_SCFG}, out['roughness'], out['metallic'], step)) > 1e-3     # the hinge is live
    n = 0
    for p, g in zip(params, ret['dp']):
        ref = p.grad if p.grad is not None else torch.zeros_like(p)
        g = g if g is not None else torch.zeros_like(p)
        scale = float(ref.abs().max())
        if scale < 1e-12:
            continue
        assert float((g - ref).abs().max()) / scale < 2e-4, float((g - ref).abs().max()) / scale
        n += 1
    assert n > 50
