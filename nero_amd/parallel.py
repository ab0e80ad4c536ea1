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
