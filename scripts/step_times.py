"""per-step wall time of the fused Stage-I training step.  usage: python scripts/step_times.py [rays] [steps] [bell|bear]
(NERO_STEP_DRIVER=py|c selects the Python-sequenced launches or the C-level driver)"""
import os, sys, time
sys.path.insert(0, '.')
import torch
from nero_amd.train import ShapeTrainStep
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
kind = sys.argv[3] if len(sys.argv) > 3 else 'bell'
cfg = {'freeze_inv_s_step': 15000, 'apply_occ_loss': True, 'occ_loss_step': 20000}
if kind == 'bear':
    cfg['shader_config'] = {'human_light': True}
ts = ShapeTrainStep(cfg, rays_per_rank=R, device='cuda:0', variance=0.5, prime_passes=2)
for i in range(5):
    ts.step(25000 + i)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
t0 = time.time()
ev[0].record()
for i in range(n):
    ts.step(25005 + i)
    ev[i + 1].record()
torch.cuda.synchronize()
wall = (time.time() - t0) / n
ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
print(f"driver={os.environ.get('NERO_STEP_DRIVER', 'c')} {kind} R={R}: wall {wall*1e3:.2f} ms/step ({R/wall:.0f} rays/s), median {ms[n//2]:.2f}, min {ms[0]:.2f}, max {ms[-1]:.2f}", flush=True)
