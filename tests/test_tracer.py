"""GPU tier: HIP BVH tracer vs the brute-force Moeller-Trumbore oracle (oracle/tracer_oracle.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('subdiv,bumps', [(1, 0.0), (3, 0.15), (5, 0.2)])
def test_closest_hit_matches_bruteforce(subdiv, bumps):
    from nero_amd.raytracing import RayTracer
    from nero_amd.synthetic import icosphere
    from oracle.tracer_oracle import trace_bruteforce
    v, f = icosphere(subdiv, 0.5, bumps)
    rt = RayTracer(v, f)
    rg = np.random.default_rng(subdiv)
    n = 3000 if subdiv < 5 else 800
    # camera rays from outside, secondary rays from (just above) the surface, rays that miss
    o1 = rg.normal(size=(n, 3)); o1 = o1 / np.linalg.norm(o1, axis=1, keepdims=True) * 2.5
    d1 = -o1 + rg.normal(size=(n, 3)) * 0.35
    c = v[f[rg.integers(0, len(f), n)]].mean(1)
    o2 = c * 1.001
    d2 = rg.normal(size=(n, 3))
    o = np.concatenate([o1, o2]).astype(np.float32)
    d = np.concatenate([d1, d2]); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    pos, nrm, depth = rt.trace(torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda())
    pos_o, nrm_o, depth_o, _ = trace_bruteforce(v, f, o, d)
    depth = depth.cpu().numpy().astype(np.float64)
    hit, hit_o = depth < 10, depth_o < 10
    # rays that graze an edge may flip; demand agreement on all but a handful
    assert (hit != hit_o).sum() <= 2
    both = hit & hit_o
    assert 0.2 < both.mean() < 0.98
    dd = np.abs(depth[both] - depth_o[both])
    assert np.quantile(dd, 0.999) < 2e-5 and (dd > 1e-3).sum() <= 2
    good = both & (np.abs(depth - depth_o) < 1e-4)
    assert np.abs(nrm.cpu().numpy()[good] - nrm_o[good]).max() < 2e-4
    assert np.abs(pos.cpu().numpy()[good] - pos_o[good]).max() < 2e-4
    miss = ~hit
    assert np.all(depth[miss] == 10.0)


def test_reference_wrapper_contract():
    """shape handling / device handling of raytracing/raytracer.py:21-54"""
    from nero_amd.raytracing import RayTracer
    from nero_amd.synthetic import icosphere
    v, f = icosphere(2, 0.5)
    rt = RayTracer(torch.from_numpy(v), torch.from_numpy(f))
    o = torch.zeros(4, 5, 3); o[..., 2] = 2.0
    d = torch.zeros(4, 5, 3); d[..., 2] = -1.0
    pos, nrm, depth = rt.trace(o, d)
    assert pos.shape == (4, 5, 3) and nrm.shape == (4, 5, 3) and depth.shape == (4, 5) and pos.is_cuda
    assert torch.allclose(depth, torch.full_like(depth, 1.5), atol=2e-2)
    assert (nrm[..., 2] > 0.9).all()                   # outward winding -> +z at the north pole
    with pytest.raises(AssertionError):
        RayTracer(v, f[:4])
