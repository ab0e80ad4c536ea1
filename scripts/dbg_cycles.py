"""find the reference cycles that keep a step's workspaces alive until the cyclic GC runs (debug aid)"""
import os, sys, gc, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nero_amd.train import ShapeTrainStep

cfg = {'freeze_inv_s_step': 15000, 'apply_occ_loss': True, 'occ_loss_step': 20000}
ts = ShapeTrainStep(cfg, rays_per_rank=1024, device='cuda:0', variance=0.5, prime_fraction=0)
ts.step(25000)
gc.collect()
gc.disable()
torch.cuda.synchronize()
m0 = torch.cuda.memory_allocated()
ts.step(25001)
torch.cuda.synchronize()
m1 = torch.cuda.memory_allocated()
gc.set_debug(gc.DEBUG_SAVEALL)
n = gc.collect()
m2 = torch.cuda.memory_allocated()
print(f'live before {m0/2**30:.2f} GB after step {m1/2**30:.2f} GB; gc found {n} objects')
cnt = collections.Counter(type(o).__name__ for o in gc.garbage)
print(cnt.most_common(25))
garb_ids = {id(o) for o in gc.garbage}
big = [o for o in gc.garbage if isinstance(o, torch.Tensor) and o.is_cuda and o.numel() > 1 << 20]
print('big tensors in garbage:', len(big))
for o in gc.garbage:
    tn = type(o).__name__
    if tn.endswith('Backward') or 'Function' in tn or tn in ('dict',) and len(o) < 40 and any(isinstance(v, torch.Tensor) for v in o.values()):
        refs = [type(r).__name__ for r in gc.get_referents(o) if id(r) in garb_ids]
        keys = list(o.keys())[:30] if isinstance(o, dict) else ''
        print(tn, '->', collections.Counter(refs).most_common(6), keys)
