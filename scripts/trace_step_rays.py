"""The secondary rays of one REAL Stage-II training step (bell 4096 x 128+128 on the benchmark mesh), captured at the tracer call:
both traversal kernels timed back to back on exactly those rays, statistics of the rays, and the rays themselves written to
gpurun_out/step_rays.npz for the host model (scripts/probe/trace_stats.cpp)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from nero_amd import _lib as L
    from nero_amd.synthetic import icosphere
    from nero_amd.train import MaterialTrainStep
    v, f = icosphere(7, 0.5, 0.2)
    f = np.ascontiguousarray(f[:, ::-1])
    scfg = dict(diffuse_sample_num=128, specular_sample_num=128, human_lights=False, outer_light_version='direction')
    ts = MaterialTrainStep({'shader_cfg': scfg, 'database_name': 'syn/bell'}, (v, f), points_per_rank=4096, pool_points=16384, device='cuda:0', fused=True)
    rt = ts.net.ray_tracer
    cap = {}
    inner = rt.trace

    def spy(o, d, *a, **k):
        cap['o'], cap['d'] = o.detach().clone(), d.detach().clone()
        return inner(o, d, *a, **k)
    rt.trace = spy
    for i in range(3):
        ts.step(5000 + i)
    torch.cuda.synchronize()
    rt.trace = inner
    o, d = cap['o'].reshape(-1, 3).contiguous(), cap['d'].reshape(-1, 3).contiguous()
    h = rt._handle()
    rec = {'rays': int(o.shape[0]), 'finite': bool(torch.isfinite(o).all() and torch.isfinite(d).all()),
           'dir_norm_min_max': [float(d.norm(dim=-1).min()), float(d.norm(dim=-1).max())],
           'origin_radius_min_max': [float(o.norm(dim=-1).min()), float(o.norm(dim=-1).max())]}
    from nero_amd.synthetic import secondary_rays
    so, sd = secondary_rays(v, f, 4096, 256)
    rad = o / o.norm(dim=-1, keepdim=True)
    perm = torch.randperm(o.shape[0], device=o.device)
    up = (d * rad).sum(-1, keepdim=True)
    variants = {'step rays': (o, d), 'step rays, origins lifted 1e-3 radially': (o + 1e-3 * rad, d), 'step rays in random order': (o[perm].contiguous(), d[perm].contiguous()),
                'step origins, synthetic directions': (o, sd), 'synthetic origins, step directions': (so, d),
                'step rays, inward directions mirrored outward': (o, torch.where(up < 0, d - 2 * up * rad, d).contiguous()), 'synthetic rays': (so, sd)}
    if os.environ.get('STEP_RAYS_ONLY'):
        variants = {k: variants[k] for k in os.environ['STEP_RAYS_ONLY'].split('|')}
    for name, (oo, dd) in variants.items():
        for mode in (0, 1):
            L.check(L.lib.nero_bvh_set_traversal(h, mode))
            for _ in range(3):
                r = rt.trace(oo, dd)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
            ev[0].record()
            for k in range(10):
                r = rt.trace(oo, dd)
                ev[k + 1].record()
            torch.cuda.synchronize()
            ms = sorted(ev[k].elapsed_time(ev[k + 1]) for k in range(10))
            rec['%s: ms mode%d' % (name, mode)] = round(ms[5], 4)
        rec['%s: hit fraction' % name] = round(float((r[2] < 10).float().mean()), 4)
    L.check(L.lib.nero_bvh_set_traversal(h, 1))
    print(json.dumps(rec), flush=True)
    os.makedirs('gpurun_out', exist_ok=True)
    np.savez_compressed('gpurun_out/step_rays.npz', o=o[:524288].cpu().numpy(), d=d[:524288].cpu().numpy())


if __name__ == '__main__':
    main()
