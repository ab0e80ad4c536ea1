"""the training step twice on the same batch with the same random draws: loss and every gradient must be identical bit for bit
(no atomics anywhere, fixed reduction orders).  usage: python scripts/determinism.py [rays] [repeats]"""
import sys
sys.path.insert(0, '.')
import torch
from nero_amd.train import ShapeTrainStep
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ts = ShapeTrainStep({'freeze_inv_s_step': 15000, 'apply_occ_loss': True, 'occ_loss_step': 20000}, rays_per_rank=R, device='cuda:0', variance=0.5,
                    prime_fraction=0.0, prime_passes=0)
ref = None
bad = 0
for k in range(N):
    ts.cursor = 0
    torch.manual_seed(1234)
    info = ts.forward_backward(25000)
    torch.cuda.synchronize()
    cur = (float(info['loss']), ts.bucket.flat.clone())
    if ref is None:
        ref = cur
        continue
    d = (cur[1] - ref[1]).abs()
    if cur[0] != ref[0] or float(d.max()) > 0:
        bad += 1
        print(f'run {k}: loss {cur[0]!r} vs {ref[0]!r}, gradient entries differing {int((d > 0).sum())}, max {float(d.max()):.3e} (of {float(ref[1].abs().max()):.3e})')
print(f'determinism R={R}: {bad} of {N - 1} repeats differ from the first run')
