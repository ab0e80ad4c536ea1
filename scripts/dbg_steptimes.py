"""per-step wall times of the Stage-I training step (debug aid: finds one-off stalls inside bench.py's timed region)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nero_amd.train import ShapeTrainStep

frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.35
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
cfg = {'freeze_inv_s_step': 15000, 'apply_occ_loss': True, 'occ_loss_step': 20000}
t0 = time.time()
ts = ShapeTrainStep(cfg, rays_per_rank=4096, device='cuda:0', variance=0.5, prime_fraction=frac)
times = []
for i in range(n):
    torch.cuda.synchronize(); a = time.time()
    ts.step(25000 + i)
    torch.cuda.synchronize(); times.append((time.time() - a) * 1e3)
print(f'frac={frac} setup+all {time.time()-t0:.1f}s steps(ms):', ' '.join(f'{t:.0f}' for t in times))
print('reserved GB', torch.cuda.memory_reserved() / 2**30, 'malloc retries', torch.cuda.memory_stats().get('num_alloc_retries'), 'segments', torch.cuda.memory_stats().get('segment.all.allocated'))
