"""per-step wall times of the first optimisation steps after construction (allocator / clock settling): python scripts/step_times.py [n]"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from nero_amd.train import ShapeTrainStep
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ts = ShapeTrainStep(bench.BELL, rays_per_rank=4096, device='cuda:0', variance=bench.VARIANCE)
torch.cuda.synchronize()
t = []
for i in range(n):
    t0 = time.perf_counter()
    ts.step(25000 + i)
    torch.cuda.synchronize()
    t.append((time.perf_counter() - t0) * 1e3)
print(' '.join(f'{x:.1f}' for x in t))
print('reserved GiB', torch.cuda.memory_reserved() / 2 ** 30, 'alloc retries', torch.cuda.memory_stats().get('num_alloc_retries'), 'segments', torch.cuda.memory_stats().get('segment.all.current'))
