"""diagnostic (round 4): where does the 1e-4 common-mode error of the Stage-II outer_light gradients (C4, D = 512+256, gate-forced
parity) come from?  Compares the per-row gradient at the outer-light head (d raw, before the exp) of the HIP step with the fp64 and
fp32 oracle runs on the same teacher-forced hits.   python scripts/r04/diag_outer_light.py [P] [Dd] [Ds]"""
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
from oracle import nero_oracle as O
from oracle import nero_oracle_mat as M
import tests.test_parity_at_size as T
from tests.helpers import CTracer as _CTracer, tracer_contract as _contract, golden_mesh, named_grads

Pn = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
Dd = int(sys.argv[2]) if len(sys.argv) > 2 else 512
Ds = int(sys.argv[3]) if len(sys.argv) > 3 else 256
shader_cfg = {**T.BELL2, 'diffuse_sample_num': Dd, 'specular_sample_num': Ds}
I = T._material_inputs(Pn)
mesh = golden_mesh()
rcfg = {'shader_cfg': shader_cfg}
from nero_amd.renderer import NeROMaterialRenderer, NeROShapeRenderer
hpl = NeROShapeRenderer.get_human_coordinate_poses(type('c', (), {'cfg': {'fixed_camera': False}})(), I['poses'])
step = 5000
ODEV = 'cuda'
tracers = {}
raws = {}
orig_exp_act = M._exp_act


def run_oracle(dtype, tag):
    cap = []

    def exp_act(mx):
        f = orig_exp_act(mx)

        def g(x):
            x.retain_grad()
            cap.append(x)
            return f(x)
        return g
    M._exp_act = exp_act
    ref = T._material_pair(shader_cfg, dtype, ODEV)
    sd = {k: v for k, v in ref.named_parameters()}
    sd.update({k: v for k, v in ref.named_buffers()})
    f = lambda a: a.to(ODEV).to(dtype)
    tr = _CTracer(*mesh, replay=tracers.get('src'))
    tracers.setdefault('src', tr)
    with torch.device(ODEV):
        oo = M.material_train_outputs(O.effective_params(sd), rcfg, _contract(tr), f(I['pts']), f(I['view']), f(I['normals']), f(hpl),
                                      f(I['gt']), step, f(I['rand_d']), f(I['rand_s']), f(I['reg_ang']), f(I['reg_eps']))
        loss = M.material_training_loss(oo)
        loss.backward()
    M._exp_act = orig_exp_act
    raws[tag] = (cap[0].detach().double().cpu(), cap[0].grad.detach().double().cpu())
    g = {k: v.double().cpu() for k, v in named_grads(ref).items()}
    state = {k: v.detach().cpu() for k, v in ref.state_dict().items()}
    return g, state


g32, state = run_oracle(torch.float32, 'f32')
g64, _ = run_oracle(torch.float64, 'f64')
net = NeROMaterialRenderer({'shader_cfg': shader_cfg, 'database_name': 'syn/bell'}, mesh=mesh)
net.load_state_dict({k: v.float() for k, v in state.items()})
net = net.cuda()
net.ray_tracer = _CTracer(*mesh, replay=tracers['src'], ray_tol=2e-5)
from nero_amd import chain as CH
hip = {}
orig_bwd = CH.Chain.backward


def bwd(self, fwd, n_rows, *a, **kw):
    if self.k_init in (72, 144) and kw.get('head_dys'):
        hip['d_raw'] = kw['head_dys'][3][:n_rows, :3].detach().double().cpu()
        hip['raw'] = fwd['heads'][3][:n_rows, :3].detach().double().cpu()
    return orig_bwd(self, fwd, n_rows, *a, **kw)


CH.Chain.backward = bwd
c = lambda k: I[k].cuda()
out = net.shade_train(c('pts'), c('view'), c('normals'), hpl.cuda(), c('gt'), step, c('rand_d'), c('rand_s'), c('reg_ang'), c('reg_eps'))
loss = out['loss_rgb'].mean() + out['loss_mat_reg'].mean() + out['loss_diffuse_light'].mean()
loss.backward()
gh = {k: v.double().cpu() for k, v in named_grads(net).items()}
r64, d64 = raws['f64']
r32, d32 = raws['f32']
dh, rh = hip['d_raw'], hip['raw']
print('rows', d64.shape, dh.shape)
rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
print('raw (head output): hip vs f64 %.2e   f32 vs f64 %.2e' % (rel(rh, r64), rel(r32, r64)))
print('d_raw per row, max-norm: hip vs f64 %.2e   f32 vs f64 %.2e' % (rel(dh, d64), rel(d32, d64)))
for name, a in (('hip', dh), ('f32', d32)):
    s, s64 = a.sum(0), d64.sum(0)
    print(f'{name}: column sums (= last-layer bias gradient) rel err', ((s - s64).abs() / s64.abs().max()).tolist())
    e = (a - d64)
    print(f'{name}: sum|err| / sum|d| = %.2e ; signed sum(err)/sum|d| = %s' % (float(e.abs().sum() / d64.abs().sum()), (e.sum(0) / d64.abs().sum(0)).tolist()))
    mag = d64.abs().max(1)[0]
    order = torch.argsort(mag, descending=True)
    for top in (10, 100, 1000, 10000, 100000):
        idx = order[:top]
        print(f'   top {top:6d} rows by |d|: share of sum|d| %.3f, rel err of their sum %.2e, max row rel err %.2e' % (
            float(d64[idx].abs().sum() / d64.abs().sum()), float((a[idx].sum(0) - d64[idx].sum(0)).abs().max() / d64[idx].sum(0).abs().max()),
            float(((a[idx] - d64[idx]).abs().max(1)[0] / mag[idx]).max())))
for k in ('shader_network.outer_light.6.bias', 'shader_network.outer_light.4.bias', 'shader_network.outer_light.0.bias'):
    print(k, 'hip %.2e f32 %.2e' % (rel(gh[k], g64[k]), rel(g32[k], g64[k])))
# ---- which rows carry the error? -------------------------------------------------------------------------------------------------
D = Dd + Ds
hit = np.asarray(tracers['src'].hit[0]).reshape(-1)
mi = np.nonzero(~hit)[0]
assert mi.shape[0] == d64.shape[0]
err = (dh - d64).abs().max(1)[0]
order = torch.argsort(err, descending=True)[:15]
print('worst rows by |hip - f64| (row, point, dir j (diffuse < %d), d64, hip, f32, raw64, rawhip):' % Dd)
for r in order.tolist():
    fi = int(mi[r])
    print(r, fi // D, fi % D, d64[r].tolist(), dh[r].tolist(), d32[r].tolist(), r64[r].tolist(), rh[r].tolist())
big = err > 1e-3 * d64.abs().max()
print('rows with err > 1e-3 of max|d|:', int(big.sum()), ' of which diffuse:', int(((torch.from_numpy(mi) % D) < Dd)[big].sum()))
pts = torch.from_numpy(mi[big.numpy()] // D)
print('distinct points among them:', int(pts.unique().numel()), pts.unique()[:20].tolist())
