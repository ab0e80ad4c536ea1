"""Data-parallel plumbing for the render step: ray sharding and the single flat gradient all-reduce (RCCL over xGMI on
MI355X: backend 'nccl'; 'gloo' on CPU / for several ranks sharing one device in tests).  The reference is single-GPU
(train/trainer.py:68-72 raises for multi_gpus); rays are independent given the weights, so this is the only communication
the path needs."""
import torch
import torch.distributed as dist


def rank_slice(cursor, rays_per_rank, rank):
    """rows of the shared, identically shuffled pool that `rank` takes from the global batch starting at `cursor`"""
    lo = cursor + rank * rays_per_rank
    return slice(lo, lo + rays_per_rank)


def _all_reduce_sum(t, group=None):
    """sum-all-reduce in place.  RCCL reduces device buffers directly; gloo (CPU tests, or two ranks on ONE device where RCCL
    refuses duplicate GPUs) goes through a host copy."""
    if t.is_cuda and dist.get_backend(group) == 'gloo':
        h = t.cpu()
        dist.all_reduce(h, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, group=group)


class GradBucket:
    """ONE persistent flat fp32 buffer holding every parameter gradient (8.8 MB for the bell shape model).  Each `p.grad` is a
    VIEW into it for the whole run, so autograd accumulates straight into the bucket: no per-step flatten / unflatten copies and
    no re-assignment of `p.grad` (the optimiser keeps seeing the same tensors).  A step is
        bucket.zero()  ->  backward  ->  bucket.all_reduce_mean(world)  ->  optimiser
    i.e. one memset and one collective.  The payload crosses xGMI in ~0.1 ms against a ~30 ms step, so it is not split into
    overlapped sub-buckets: the whole backward is a single autograd node and there is nothing to hide it behind."""

    def __init__(self, params):
        self.params = [p for p in params]
        dev, n = self.params[0].device, sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def reattach(self):
        """make every p.grad a view of the flat buffer again.  Something outside may have severed one (module.zero_grad() /
        optimizer.zero_grad() default to set_to_none=True; `p.grad = fresh` re-binds): a severed .grad would silently drop out of
        the collective and of the fused Adam kernel.  A gradient found in a foreign tensor is copied into its slice first."""
        off = 0
        for p in self.params:
            n = p.numel()
            g = p.grad
            if g is None or g.data_ptr() != self.flat.data_ptr() + 4 * off:
                view = self.flat[off:off + n].view_as(p)
                if g is not None:
                    view.copy_(g)
                p.grad = view
            off += n

    def zero(self):
        self.reattach()
        self.flat.zero_()

    def all_reduce_mean(self, world, group=None):
        self.reattach()
        if world <= 1 and not (FORCE_COLLECTIVES and dist.is_available() and dist.is_initialized()):
            return
        _all_reduce_sum(self.flat, group)
        self.flat.mul_(1.0 / world)


def allreduce_mean_grads(params, world, group=None):
    """bucket-less variant (kept for callers that own their .grad tensors): flatten, all-reduce (sum), divide, scatter back."""
    if world <= 1:
        return
    grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in params]
    flat = torch._utils._flatten_dense_tensors(grads)
    _all_reduce_sum(flat, group)
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
    _all_reduce_sum(t, group)
    total = float(t.item())
    return float(local_count) * world / total if total > 0 else 1.0


def global_count_weights(local_counts, world, device, group=None):
    """global_count_weight for several per-sample loss terms at once (eikonal: inner-sample count; occlusion loss: candidate
    count) with ONE small all-reduce"""
    if world <= 1:
        return [1.0] * len(local_counts)
    t = torch.tensor([float(c) for c in local_counts], dtype=torch.float64, device=device)
    _all_reduce_sum(t, group)
    tot = t.tolist()
    return [float(c) * world / g if g > 0 else 1.0 for c, g in zip(local_counts, tot)]


def device_count_weights(local_counts, world, device, group=None):
    """global_count_weights WITHOUT the host stall: the weights stay on the device (a float32 tensor, one element per term) and
    multiply the loss terms there -- the all-reduce is enqueued on the stream like every other launch of the step and nothing waits
    for it on the host (the blocking .tolist() cost one host round trip per step and rank on the DP path)."""
    if world <= 1 and not (group is None and FORCE_COLLECTIVES and dist.is_available() and dist.is_initialized()):
        return torch.ones(len(local_counts), dtype=torch.float32, device=device)
    if any(torch.is_tensor(c) for c in local_counts):          # (a count that lives on the device -- the occlusion-loss candidates of the HIP-glued step)
        loc = torch.stack([c.reshape(()).to(device=device, dtype=torch.float64) if torch.is_tensor(c)
                           else torch.tensor(float(c), dtype=torch.float64, device=device) for c in local_counts])
    else:
        loc = torch.tensor([float(c) for c in local_counts], dtype=torch.float64, device=device)
    tot = loc.clone()
    _all_reduce_sum(tot, group)
    w = torch.where(tot > 0, loc * float(world) / tot.clamp(min=1.0), torch.ones_like(loc))
    return w.to(torch.float32)


# set by bench.py when it runs as ONE rank under torch.distributed.run: the collectives of the data-parallel path (count weights, the
# flat gradient all-reduce) are then issued although world == 1, so that the N = 1 point of a scaling run exercises the RCCL code path
FORCE_COLLECTIVES = False


def parallel_forced():
    """True when the collectives of the data-parallel path are issued although world == 1 (FORCE_COLLECTIVES under an initialised group)"""
    return bool(FORCE_COLLECTIVES and dist.is_available() and dist.is_initialized())


def per_rank_occ_cap(max_pn, world):
    """the reference caps the occlusion-loss candidate set at occ_loss_max_pn per PROCESS (network/renderer.py:535-541); with
    `world` ranks each takes max_pn / world so that the global candidate budget stays what the single-process step uses
    (SURVEY.md §8e)."""
    return max(1, int(max_pn) // max(1, int(world)))
