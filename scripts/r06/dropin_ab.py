"""drop-in trainer path (INTEGRATION.md option A: NeROShapeRenderer.forward + torch.optim.Adam) with the weight-normed Linears through
torch._weight_norm per Linear (NERO_WN_BATCH=0, rounds 1-5) and through the one batched autograd node of nero_amd/wn_fused.py, next to the
fused trainer step, at the reference's own batch (512 rays) and at 4096: python scripts/r06/dropin_ab.py"""
import os, sys, time, json
sys.path.insert(0, '.')
import torch
import bench as B
from nero_amd.train import ShapeTrainStep
dev = 'cuda:0'
cfg = dict(B.BELL)
res = {}
for rays in (512, 4096):
    for rep in range(2):
        for wn in ('0', '1'):
            os.environ['NERO_WN_BATCH'] = wn
            r = B.dropin_trainer_bench(dev, cfg, rays, B.VARIANCE)
            res.setdefault(f'dropin_r{rays}_wn_batch{wn}', []).append(r['ms_per_step'])
        ts = ShapeTrainStep(cfg, rays_per_rank=rays, device=dev, variance=B.VARIANCE, prime_fraction=0.0)
        for i in range(5): ts.step(25000 + i)
        torch.cuda.synchronize(); t0 = time.time()
        for i in range(20): ts.step(25005 + i)
        torch.cuda.synchronize()
        res.setdefault(f'fused_trainer_r{rays}', []).append(round((time.time() - t0) / 20 * 1e3, 3))
        del ts
        torch.cuda.empty_cache()
print(json.dumps(res, indent=1))
