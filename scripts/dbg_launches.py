"""per-launch time / TFLOP/s of the MFMA kernels over ONE training step (tuning aid; uses NERO_PROF_DUMP)"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dump = '/tmp/nero_launches.txt'
os.environ['NERO_PROF_DUMP'] = dump
import torch
from nero_amd import _lib as L
from nero_amd.train import ShapeTrainStep
cfg = {'freeze_inv_s_step': 15000, 'apply_occ_loss': True, 'occ_loss_step': 20000}
ts = ShapeTrainStep(cfg, rays_per_rank=4096, device='cuda:0', variance=0.5)
for i in range(4): ts.step(25000 + i)
torch.cuda.synchronize()
if os.path.exists(dump): os.remove(dump)
L.lib.nero_prof_enable(1); ts.step(25010); torch.cuda.synchronize(); L.lib.nero_prof_enable(0)
rep = (C.c_double * 12)(); L.lib.nero_prof_report(rep)
names = ['fwd', 'tan', 'bwd', 'dW']
tot = [0, 0, 0, 0]
for i, ln in enumerate(open(dump)):
    k, ms, fl = ln.split(); k = int(k); ms = float(ms); fl = float(fl)
    tot[k] += ms
    print(f'{i:3d} {names[k]:4s} {ms:7.3f} ms  {fl/1e9:8.1f} GF  {fl/ms/1e9:6.1f} TF')
print('totals ms', dict(zip(names, [round(t, 2) for t in tot])))
