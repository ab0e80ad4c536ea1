"""per-step allocator statistics of the Stage-I training step (debug aid)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nero_amd.train import ShapeTrainStep

frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
cfg = {'freeze_inv_s_step': 15000, 'apply_occ_loss': True, 'occ_loss_step': 20000}
ts = ShapeTrainStep(cfg, rays_per_rank=4096, device='cuda:0', variance=0.5, prime_fraction=frac)
G = 2**30
for i in range(20):
    torch.cuda.reset_peak_memory_stats()
    s0 = torch.cuda.memory_stats()
    torch.cuda.synchronize(); a = time.time()
    ts.step(25000 + i)
    torch.cuda.synchronize(); dt = (time.time() - a) * 1e3
    s = torch.cuda.memory_stats()
    print(f"step {i}: {dt:7.0f} ms  live {s['allocated_bytes.all.current']/G:6.2f}  peak {s['allocated_bytes.all.peak']/G:6.2f}  "
          f"cum {(s['allocated_bytes.all.allocated']-s0['allocated_bytes.all.allocated'])/G:6.2f}  reserved {s['reserved_bytes.all.current']/G:7.2f}  "
          f"segs {s['segment.all.current']}  new_segs {s['segment.all.allocated']-s0['segment.all.allocated']} n_alloc {s['allocation.all.allocated']-s0['allocation.all.allocated']}")
