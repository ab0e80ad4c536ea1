import sys; sys.path.insert(0,'.')
import torch, numpy as np
from oracle import nero_oracle as O
from tests.helpers import T, build_case_model, load_golden
def rel(a,b):
    a,b=a.detach().double().cpu(),b.detach().double().cpu(); return float((a-b).abs().max()/(b.abs().max()+1e-30))
for name in ['bell_s25000','bell_c1']:
    z, meta = load_golden(name)
    net = build_case_model(meta).cuda(); ref = build_case_model(meta)
    sd = {k: v for k, v in ref.named_parameters()}; sd.update({k: v for k, v in ref.named_buffers()})
    P = O.effective_params(sd)
    cfg = {**O.DEFAULT_CFG, **meta['cfg'], 'apply_occ_loss': False}
    oo = O.render_core(P, cfg, T(z,'o'), T(z,'d'), T(z,'z_vals'), T(z,'human_poses'), meta['anneal'], meta['step'])
    (O.rgb_loss(cfg, oo['ray_rgb'], T(z,'gt')).mean() + (oo['gradient_error']*0.1).mean()).backward()
    out = net.render(T(z,'o','cuda'), T(z,'d','cuda'), T(z,'near','cuda'), T(z,'far','cuda'), T(z,'human_poses','cuda'), -1, meta['anneal'], is_train=True, step=meta['step'], z_vals=T(z,'z_vals','cuda'))
    print(name, 'rgb', rel(out['ray_rgb'], oo['ray_rgb']), 'gerr', rel(out['gradient_error'], oo['gradient_error']))
    (net.compute_rgb_loss(out['ray_rgb'], T(z,'gt','cuda')).mean() + (out['gradient_error']*0.1).mean()).backward()
    for (k,p),(_,q) in zip(net.named_parameters(), ref.named_parameters()):
        gq = q.grad if q.grad is not None else torch.zeros_like(q); gp = p.grad if p.grad is not None else torch.zeros_like(p)
        print(f'  {k:50s} max|g| {float(gq.abs().max()):.3e} rel {rel(gp,gq):.2e}')
