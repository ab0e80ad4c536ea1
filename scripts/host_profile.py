"""host-side (Python) cost of the training step: cProfile over a few steps, top functions by own time (run on the GPU box)"""
import cProfile, io, os, pstats, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from nero_amd.train import ShapeTrainStep
ts = ShapeTrainStep(bench.BELL, rays_per_rank=4096, device='cuda:0', variance=bench.VARIANCE)
for i in range(5):
    ts.step(25000 + i)
torch.cuda.synchronize()
n = 10
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for i in range(n):
    ts.step(25010 + i)
pr.disable()
t_launch = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f'{n} steps: host returned after {t_launch / n * 1e3:.2f} ms/step, GPU done after {t_all / n * 1e3:.2f} ms/step (under cProfile)')
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
print('\n'.join(l[:150] for l in s.getvalue().split('\n')[:45]))
