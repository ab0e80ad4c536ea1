"""drop-in trainer path against the fused trainer at the reference's batch (512 rays), interleaved, 5 repeats of 60 steps, medians:
python scripts/r06/dropin_ab2.py [rays]"""
import os, sys, time, json, statistics
sys.path.insert(0, '.')
import torch
import bench as B
from nero_amd.train import ShapeTrainStep
dev = 'cuda:0'
cfg = dict(B.BELL)
rays = int(sys.argv[1]) if len(sys.argv) > 1 else 512
res = {}
for rep in range(5):
    r = B.dropin_trainer_bench(dev, cfg, rays, B.VARIANCE, warmup=8, steps=60)
    res.setdefault('dropin', []).append(r['ms_per_step'])
    ts = ShapeTrainStep(cfg, rays_per_rank=rays, device=dev, variance=B.VARIANCE, prime_fraction=0.0)
    for i in range(8): ts.step(25000 + i)
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(60): ts.step(25008 + i)
    torch.cuda.synchronize()
    res.setdefault('fused', []).append(round((time.time() - t0) / 60 * 1e3, 3))
    del ts
    torch.cuda.empty_cache()
med = {k: statistics.median(v) for k, v in res.items()}
print(json.dumps(res), '\nmedians', med, 'drop-in / fused = %.3f' % (med['dropin'] / med['fused']))
