import sys; sys.path.insert(0,'.')
import numpy as np, torch
from oracle import nero_oracle as O
from oracle import nero_oracle_mat as M
from tests.helpers import T, build_material_case, golden_mesh, load_golden, oracle_trace_fn
from tests.test_material_render import OracleTracer
from nero_amd.renderer import NeROMaterialRenderer
def rel(a,b):
    a,b=a.detach().double().cpu(),b.detach().double().cpu(); return float((a-b).abs().max()/(b.abs().max()+1e-30))
z, meta = load_golden('mat_bell')
res={}
tr0 = oracle_trace_fn()
for dt in (torch.float32, torch.float64):
    ref = build_material_case(meta).to(dt)
    sd = {k: v for k, v in ref.named_parameters()}; sd.update({k: v for k, v in ref.named_buffers()})
    P = O.effective_params(sd)
    def tr(o,d):
        a,b,c,h = tr0(o.float(), d.float()); return a.to(dt), b.to(dt), c.to(dt), h
    f=lambda k: T(z,k).to(dt)
    oo = M.material_train_outputs(P, {'shader_cfg': meta['shader_cfg']}, tr, f('pts'), f('view'), f('normals'), f('human_poses'), f('gt'), meta['step'], f('rand_d'), f('rand_s'), f('reg_ang'), f('reg_eps'))
    M.material_training_loss(oo).backward()
    res[dt]={k:(q.grad if q.grad is not None else torch.zeros_like(q)) for k,q in ref.named_parameters()}
ref = build_material_case(meta)
net = NeROMaterialRenderer({'shader_cfg': meta['shader_cfg'], 'database_name': 'syn/bell'}, mesh=golden_mesh())
net.load_state_dict(ref.state_dict()); net=net.cuda(); net.ray_tracer = OracleTracer(*golden_mesh())
c = lambda k: T(z,k,'cuda')
out = net.shade_train(c('pts'), c('view'), c('normals'), c('human_poses'), c('gt'), meta['step'], c('rand_d'), c('rand_s'), c('reg_ang'), c('reg_eps'))
(out['loss_rgb'].mean() + out['loss_mat_reg'].mean() + out['loss_diffuse_light'].mean()).backward()
for k,p in net.named_parameters():
    if 'roughness_predictor.6' in k or 'inner_light.2' in k or 'metallic_predictor.6' in k or 'outer_light.6' in k:
        gp = p.grad if p.grad is not None else torch.zeros_like(p)
        print(f'{k:50s} hip-vs-f64 {rel(gp,res[torch.float64][k]):.2e}  cpu32-vs-f64 {rel(res[torch.float32][k],res[torch.float64][k]):.2e} |g| {float(res[torch.float64][k].abs().max()):.2e}')
