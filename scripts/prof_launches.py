"""per-launch HIP-event timing of the chain / weight-gradient kernels of ONE training step (NERO_PROF_DUMP): kind, ms, TFLOP/s"""
import ctypes as C, os, sys
sys.path.insert(0, '.')
dump = '/tmp/nero_prof_dump.txt'
os.environ['NERO_PROF_DUMP'] = dump
import torch
from nero_amd import _lib as L
from nero_amd.train import ShapeTrainStep
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ts = ShapeTrainStep({'freeze_inv_s_step': 15000, 'apply_occ_loss': True, 'occ_loss_step': 20000}, rays_per_rank=R, device='cuda:0', variance=0.5, prime_passes=2)
for i in range(4):
    ts.step(25000 + i)
torch.cuda.synchronize()
if os.path.exists(dump):
    os.remove(dump)
L.lib.nero_prof_enable(1)
ts.step(25010)
torch.cuda.synchronize()
L.lib.nero_prof_enable(0)
rep = (C.c_double * 12)()
L.lib.nero_prof_report(rep)
names = {0: 'fwd', 1: 'tan', 2: 'bwd', 3: 'dw', 4: 'fwP', 5: 'taP', 6: 'bwP'}      # 4-6: the two-workgroups-per-CU kernels (mlp_f16p.hip)
tot = {}
for i, line in enumerate(open(dump)):
    k, ms, fl, rows, sig = (line.split() + ['0', '0x0'])[:5]
    k, ms, fl = int(k), float(ms), float(fl)
    tot[k] = tot.get(k, 0.0) + ms
    print(f'{i:3d} {names[k]:3s} {ms:8.4f} ms {fl / 1e9:10.2f} GFLOP {fl / ms / 1e9:8.1f} TFLOP/s rows {rows} sig {sig}')
print({names[k]: round(v, 3) for k, v in tot.items()})
