import sys, time, os
sys.path.insert(0,'.')
import torch
from nero_amd.train import ShapeTrainStep
cfg = {'freeze_inv_s_step': 15000, 'apply_occ_loss': True, 'occ_loss_step': 20000}
ts = ShapeTrainStep(cfg, rays_per_rank=4096, device='cuda:0', variance=0.5)
def sync(): torch.cuda.synchronize()
for i in range(6):
    sync(); t0=time.time()
    info = ts.step(5000+i)
    t1=time.time(); sync(); t2=time.time()
    st=torch.cuda.memory_stats()
    print(f'step {i}: host {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms, n_in {info["n_in"]}, segs {st["num_device_alloc"]} frees {st["num_device_free"]} retries {st["num_alloc_retries"]} reserved {st["reserved_bytes.all.current"]/2**30:.1f} GiB alloc-peak {st["allocated_bytes.all.peak"]/2**30:.1f}')
import cProfile, pstats
pr=cProfile.Profile(); pr.enable(); ts.step(5010); sync(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
