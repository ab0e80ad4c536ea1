import sys; sys.path.insert(0,'.')
import numpy as np, torch
from oracle import nero_oracle as O
from oracle import nero_oracle_mat as M
from tests.helpers import T, build_material_case, golden_mesh, load_golden, oracle_trace_fn
from nero_amd.renderer import NeROMaterialRenderer
def rel(a,b):
    a,b=a.detach().double().cpu(),b.detach().double().cpu(); return float((a-b).abs().max()/(b.abs().max()+1e-30))
z, meta = load_golden('mat_bell')
ref = build_material_case(meta)
sd = {k: v for k, v in ref.named_parameters()}; sd.update({k: v for k, v in ref.named_buffers()})
P = O.effective_params(sd)
tr = oracle_trace_fn()
oo = M.material_train_outputs(P, {'shader_cfg': meta['shader_cfg']}, tr, T(z,'pts'), T(z,'view'), T(z,'normals'), T(z,'human_poses'), T(z,'gt'), meta['step'], T(z,'rand_d'), T(z,'rand_s'), T(z,'reg_ang'), T(z,'reg_eps'))
net = NeROMaterialRenderer({'shader_cfg': meta['shader_cfg'], 'database_name': 'syn/bell'}, mesh=golden_mesh())
net.load_state_dict(ref.state_dict()); net=net.cuda()
c = lambda k: T(z,k,'cuda')
out = net.shade_train(c('pts'), c('view'), c('normals'), c('human_poses'), c('gt'), meta['step'], c('rand_d'), c('rand_s'), c('reg_ang'), c('reg_eps'))
for k in ('albedo','roughness','metallic','diffuse_light','specular_light','specular_color','rgb_pr'):
    print(k, rel(out[k], oo[k]))
print('hit frac oracle', float(oo['hit_fraction']))
# directions check: recompute oracle dirs
from oracle.nero_oracle_mat import *
import torch.nn.functional as F
v, n = F.normalize(T(z,'view'),dim=-1), F.normalize(T(z,'normals'),dim=-1)
refl = torch.sum(v*n,-1,keepdim=True)*n*2-v
dd = sample_diffuse_directions(n, 16, T(z,'rand_d'))
ss = sample_specular_directions(refl, oo['roughness'].detach(), 8, T(z,'rand_s'))
dirs_o = torch.cat([dd,ss],1).reshape(-1,3)
S = None
