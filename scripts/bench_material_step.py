"""Stage-II training step through nero_amd.train.MaterialTrainStep (fused weight-norm / Adam kernels, flat gradient bucket) next to
the torch trainer loop on the same workload.  usage: python scripts/bench_material_step.py [P] [Dd] [Ds] [subdiv] [bell|bear]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from nero_amd.synthetic import icosphere
from nero_amd.train import MaterialTrainStep
P_ = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
Dd = int(sys.argv[2]) if len(sys.argv) > 2 else 128
Ds = int(sys.argv[3]) if len(sys.argv) > 3 else 128
sub = int(sys.argv[4]) if len(sys.argv) > 4 else 7
kind = sys.argv[5] if len(sys.argv) > 5 else 'bell'
only = sys.argv[6] if len(sys.argv) > 6 else 'both'          # 'fused' / 'torch' / 'both'
v, f = icosphere(sub, 0.5, 0.2)
f = np.ascontiguousarray(f[:, ::-1])
scfg = dict(diffuse_sample_num=Dd, specular_sample_num=Ds, human_lights=(kind == 'bear'),
            outer_light_version='sphere_direction' if kind == 'bear' else 'direction')
for fused in {'both': (False, True), 'fused': (True,), 'torch': (False,)}[only]:
    ts = MaterialTrainStep({'shader_cfg': scfg, 'database_name': 'real/bear' if kind == 'bear' else 'syn/bell'}, (v, f), points_per_rank=P_,
                           pool_points=4 * P_, device='cuda:0', fused=fused)
    for i in range(4):
        ts.step(5000 + i)
    torch.cuda.synchronize(); t = time.time(); n = 10
    for i in range(n):
        ts.step(5004 + i)
    torch.cuda.synchronize(); dt = (time.time() - t) / n
    print(f'{kind} P={P_} D={Dd}+{Ds} fused={fused}: {dt*1e3:.2f} ms/step, {P_/dt:.0f} pts/s, {P_*(Dd+Ds)/dt/1e6:.1f} M light-rays/s', flush=True)
    del ts
    torch.cuda.empty_cache()
