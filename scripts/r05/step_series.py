"""per-step event times IN ORDER over the first steps after construction (does the step settle, and when?): python scripts/r05/step_series.py [rays] [steps]"""
import sys
sys.path.insert(0, '.')
import torch
from nero_amd.train import ShapeTrainStep
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 90
ts = ShapeTrainStep({'freeze_inv_s_step': 15000, 'apply_occ_loss': True, 'occ_loss_step': 20000}, rays_per_rank=R, device='cuda:0', variance=0.5)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
info = []
ev[0].record()
for i in range(n):
    o = ts.step(25000 + i)
    info.append((o['n_in'], o['n_out']))
    ev[i + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
for i in range(0, n, 10):
    print(f'{i:3d}:', ' '.join(f'{m:6.2f}' for m in ms[i:i + 10]), '| n_in', info[i][0], 'n_out', info[i][1])
import subprocess
print(subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True).stdout[-1500:])
