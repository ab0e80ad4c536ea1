"""which gradient tensors differ between repeats of the same Stage-I step (NERO_STREAMS=3 race hunt)"""
import os, sys
sys.path.insert(0, '.')
import torch
from nero_amd.train import ShapeTrainStep
kind = sys.argv[1] if len(sys.argv) > 1 else 'bear'
rays = int(sys.argv[2]) if len(sys.argv) > 2 else 512
BELL = {'freeze_inv_s_step': 15000, 'apply_occ_loss': True, 'occ_loss_step': 20000}
cfg = dict(BELL) if kind == 'bell' else {**BELL, 'shader_config': {'human_light': True}}
ts = ShapeTrainStep(cfg, rays_per_rank=rays, pool_rays=4 * rays, device='cuda:0', variance=0.5, prime_fraction=0.0, prime_passes=0)
names = ts.fopt.names + ['variance']
ref = None
for k in range(10):
    ts.cursor = 0
    torch.manual_seed(1234)
    info = ts.forward_backward(25000)
    torch.cuda.synchronize()
    cur = ts.bucket.flat.clone()
    if ref is None:
        ref = cur
        continue
    if not torch.equal(cur, ref):
        off = 0
        bad = []
        for n, p in zip(names, ts.bucket.params):
            a, b = cur[off:off + p.numel()], ref[off:off + p.numel()]
            off += p.numel()
            if not torch.equal(a, b):
                bad.append((n, int((a != b).sum()), p.numel(), float((a - b).abs().max() / (b.abs().max() + 1e-30))))
        print(f'repeat {k}: streams={os.environ.get("NERO_STREAMS")} differing tensors:', bad, flush=True)
    else:
        print(f'repeat {k}: identical', flush=True)
