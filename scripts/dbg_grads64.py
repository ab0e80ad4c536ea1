import sys; sys.path.insert(0,'.')
import torch, numpy as np
from oracle import nero_oracle as O
from tests.helpers import T, build_case_model, load_golden
def rel(a,b):
    a,b=a.detach().double().cpu(),b.detach().double().cpu(); return float((a-b).abs().max()/(b.abs().max()+1e-30))
name='bell_s25000'
z, meta = load_golden(name)
net = build_case_model(meta).cuda()
res={}
for dt in (torch.float32, torch.float64):
    ref = build_case_model(meta).to(dt)
    sd = {k: v for k, v in ref.named_parameters()}; sd.update({k: v for k, v in ref.named_buffers()})
    P = O.effective_params(sd)
    cfg = {**O.DEFAULT_CFG, **meta['cfg'], 'apply_occ_loss': False}
    f=lambda k: T(z,k).to(dt)
    oo = O.render_core(P, cfg, f('o'), f('d'), f('z_vals'), f('human_poses'), meta['anneal'], meta['step'])
    (O.rgb_loss(cfg, oo['ray_rgb'], f('gt')).mean() + (oo['gradient_error']*0.1).mean()).backward()
    res[dt]={k:(q.grad if q.grad is not None else torch.zeros_like(q)) for k,q in ref.named_parameters()}
out = net.render(T(z,'o','cuda'), T(z,'d','cuda'), T(z,'near','cuda'), T(z,'far','cuda'), T(z,'human_poses','cuda'), -1, meta['anneal'], is_train=True, step=meta['step'], z_vals=T(z,'z_vals','cuda'))
(net.compute_rgb_loss(out['ray_rgb'], T(z,'gt','cuda')).mean() + (out['gradient_error']*0.1).mean()).backward()
for k,p in net.named_parameters():
    gp = p.grad if p.grad is not None else torch.zeros_like(p)
    if 'metallic' in k or 'inner_weight' in k or 'lin4' in k:
        print(f'{k:48s} hip-vs-f64 {rel(gp,res[torch.float64][k]):.2e}  cpu32-vs-f64 {rel(res[torch.float32][k],res[torch.float64][k]):.2e}')
