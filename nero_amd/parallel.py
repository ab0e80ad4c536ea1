"""Data-parallel plumbing for the render step: ray sharding and the single flat gradient all-reduce (RCCL over xGMI on
MI355X: backend 'nccl'; 'gloo' on CPU for tests).  The reference is single-GPU (train/trainer.py:68-72 raises for
multi_gpus); rays are independent given the weights, so this is the only communication the path needs."""
import torch
import torch.distributed as dist


def rank_slice(cursor, rays_per_rank, rank):
    """rows of the shared, identically shuffled pool that `rank` takes from the global batch starting at `cursor`"""
    lo = cursor + rank * rays_per_rank
    return slice(lo, lo + rays_per_rank)


def allreduce_mean_grads(params, world, group=None):
    """ONE collective per step: flatten every gradient into a single fp32 bucket (8.8 MB for the bell shape model),
    all-reduce (sum), divide by world, scatter back.  Parameters without a gradient contribute zeros."""
    if world <= 1:
        return
    grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in params]
    flat = torch._utils._flatten_dense_tensors(grads)
    dist.all_reduce(flat, group=group)
    flat.div_(world)
    for p, g in zip(params, torch._utils._unflatten_dense_tensors(flat, grads)):
        p.grad = g.contiguous()


def global_count_weight(local_count, world, device, group=None):
    """factor w such that  mean_over_ranks( w_r * sum_i v_ri / n_r )  ==  sum_ri v_ri / sum_r n_r :  the per-sample loss terms
    whose sample count differs between ranks (the eikonal term is a mean over each rank's inner samples, network/renderer.py:574,
    network/loss.py:42) then reproduce the single-process big-batch mean exactly.  One scalar all-reduce."""
    if world <= 1:
        return 1.0
    t = torch.tensor([float(local_count)], dtype=torch.float64, device=device)
    dist.all_reduce(t, group=group)
    total = float(t.item())
    return float(local_count) * world / total if total > 0 else 1.0
