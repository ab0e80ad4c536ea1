"""soak run: a few hundred optimisation steps of every training configuration bench.py touches, checking that the loss stays finite
and goes down and that the parameters stay finite.  usage: python scripts/soak.py [steps]"""
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
from nero_amd.synthetic import icosphere
from nero_amd.train import MaterialTrainStep, ShapeTrainStep
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
BELL = {'freeze_inv_s_step': 15000, 'apply_occ_loss': True, 'occ_loss_step': 20000}


def run(name, ts, step0):
    losses = []
    for i in range(N):
        losses.append(ts.step(step0 + i)['loss'])
    torch.cuda.synchronize()
    l = torch.stack([x.reshape(()) for x in losses]).cpu().numpy()
    finite = bool(np.isfinite(l).all()) and all(bool(torch.isfinite(p).all()) for p in ts.net.parameters())
    a, b = float(l[:20].mean()), float(l[-20:].mean())
    print(f'{name}: {N} steps, loss {a:.5f} -> {b:.5f}, finite={finite}', flush=True)
    assert finite and b < a, name
    del ts
    torch.cuda.empty_cache()


run('stage1 bell 4096', ShapeTrainStep(BELL, rays_per_rank=4096, device='cuda:0', variance=0.5, prime_passes=1), 25000)
run('stage1 bell 512 (early schedule: init-SDF regulariser, frozen variance)', ShapeTrainStep(BELL, rays_per_rank=512, device='cuda:0', prime_fraction=0.0, prime_passes=0), 0)
run('stage1 bear 1024', ShapeTrainStep({**BELL, 'shader_config': {'human_light': True}}, rays_per_rank=1024, device='cuda:0', variance=0.5, prime_fraction=0.0,
                                      prime_passes=0), 25000)
v, f = icosphere(6, 0.5, 0.2)
mesh = (v, np.ascontiguousarray(f[:, ::-1]))
run('stage2 bell 4096 x 256 (hinge steps)', MaterialTrainStep({'shader_cfg': dict(diffuse_sample_num=128, specular_sample_num=128), 'database_name': 'syn/bell'}, mesh,
                                                             points_per_rank=4096, device='cuda:0'), 0)
run('stage2 bear 1024 x 128', MaterialTrainStep({'shader_cfg': dict(diffuse_sample_num=64, specular_sample_num=64, human_lights=True, outer_light_version='sphere_direction'),
                                                 'database_name': 'real/bear'}, mesh, points_per_rank=1024, device='cuda:0'), 3000)
print('soak ok')
