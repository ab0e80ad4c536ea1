"""Tracer microbenchmark (GPU box): the two traversal kernels of nero_amd/csrc/bvh.hip (nero_bvh_set_traversal) on the benchmark mesh, secondary rays shaped like
the Stage-II step's (P surface points x D directions: cosine hemisphere + a specular lobe), bit-for-bit comparison of the outputs
and the time per launch.  Usage: python scripts/trace_bench.py [out.json]"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from bench import _bench_mesh
    from nero_amd import _lib as L
    from nero_amd.raytracing import RayTracer
    from nero_amd.synthetic import camera_rays, secondary_rays
    v, f = _bench_mesh(7)
    rt = RayTracer(v, f)
    h = rt._handle()
    out = {'mesh_triangles': int(f.shape[0]), 'cases': []}
    cases = [('secondary 4096 x 256', lambda: secondary_rays(v, f, 4096, 256)), ('secondary 4096 x 768', lambda: secondary_rays(v, f, 4096, 768)),
             ('secondary 2048 x 512', lambda: secondary_rays(v, f, 2048, 512, seed=1)), ('secondary 190 x 256 (small launch)', lambda: secondary_rays(v, f, 190, 256, seed=2)),
             ('camera 1024 x 1024 (coherent)', lambda: camera_rays(1024))]
    sel = os.environ.get('TRACE_CASES')
    if sel:
        cases = [cases[int(k)] for k in sel.split(',')]
    iters = int(os.environ.get('TRACE_ITERS', '20'))
    for name, make in cases:
        o, d = make()
        res = {}
        rec = {'case': name, 'rays': int(o.shape[0])}
        modes = tuple(int(m) for m in os.environ.get('TRACE_MODES', '0,1').split(','))
        for mode in modes:
            L.check(L.lib.nero_bvh_set_traversal(h, mode))
            for _ in range(3):
                r = rt.trace(o, d)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
            ev[0].record()
            for k in range(iters):
                r = rt.trace(o, d)
                ev[k + 1].record()
            torch.cuda.synchronize()
            ms = sorted(ev[k].elapsed_time(ev[k + 1]) for k in range(iters))
            res[mode] = [x.clone() for x in r]
            rec['ms_median_mode%d' % mode] = round(ms[len(ms) // 2], 4)
            rec['grays_per_s_mode%d' % mode] = round(o.shape[0] / ms[len(ms) // 2] / 1e6, 3)
        rec['hit_fraction'] = round(float((res[modes[0]][2] < 10).float().mean()), 4)
        for mode in modes[1:]:
            a, b = res[modes[0]], res[mode]
            rec['mode%d_vs_mode%d' % (mode, modes[0])] = {
                'depth_and_position_bit_identical': torch.equal(a[2], b[2]) and torch.equal(a[0], b[0]),
                'rays_with_other_depth': int((a[2] != b[2]).sum()), 'max_depth_difference': float((a[2] - b[2]).abs().max()),
                'normals_bit_identical': torch.equal(a[1], b[1]), 'max_normal_difference': float((a[1] - b[1]).abs().max()),
                'speedup': round(rec['ms_median_mode%d' % modes[0]] / rec['ms_median_mode%d' % mode], 3)}
        print(json.dumps(rec), flush=True)
        out['cases'].append(rec)
    if len(sys.argv) > 1:
        with open(sys.argv[1], 'w') as fp:
            json.dump(out, fp, indent=1)


if __name__ == '__main__':
    main()
