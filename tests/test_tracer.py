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


@pytest.mark.parametrize('subdiv,n_pts,n_dir', [(2, 37, 33), (5, 300, 64), (7, 1024, 128)])
def test_both_traversal_kernels_agree_bit_for_bit(subdiv, n_pts, n_dir):
    """nero_bvh_set_traversal: the default kernel (memory requests of a step overlapped, LDS stack) visits the tree in the same order
    with the same arithmetic as the one-request-after-the-other kernel -- positions, normals and depths identical, on secondary rays
    leaving the surface (incoherent) and on camera rays; ray counts that are not a multiple of the workgroup size"""
    import ctypes as C
    from nero_amd import _lib as L
    from nero_amd.raytracing import RayTracer
    from nero_amd.synthetic import camera_rays, icosphere, secondary_rays
    v, f = icosphere(subdiv, 0.5, 0.2)
    f = np.ascontiguousarray(f[:, ::-1])
    rt = RayTracer(v, f)
    h = rt._handle()
    o1, d1 = secondary_rays(v, f, n_pts, n_dir, seed=subdiv)
    o2, d2 = camera_rays(61)
    o, d = torch.cat([o1, o2]), torch.cat([d1, d2])
    res = []
    for mode in (0, 1):
        L.check(L.lib.nero_bvh_set_traversal(h, mode))
        res.append([x.clone() for x in rt.trace(o, d)])
    L.check(L.lib.nero_bvh_set_traversal(h, 1))
    hit = res[0][2] < 10
    assert 0.05 < float(hit.float().mean()) < 0.95
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b), int((a != b).sum())
    with pytest.raises(RuntimeError):
        L.check(L.lib.nero_bvh_set_traversal(h, 2))


@pytest.mark.parametrize('n_pts,group,heavy_from', [(300, 256, 128), (97, 768, 512), (64, 128, 64), (10, 192, 64)])
def test_grouped_launch_order_is_bit_identical_to_the_natural_order(n_pts, group, heavy_from):
    """nero_bvh_trace_grouped (round 5): the heavy chunks of every group of rays are STARTED first -- same rays, same arithmetic, same
    outputs at the same addresses as nero_bvh_trace; arguments that do not fit the chunking fall back to the natural order"""
    from nero_amd.raytracing import RayTracer
    from nero_amd.synthetic import icosphere, secondary_rays
    v, f = icosphere(5, 0.5, 0.2)
    f = np.ascontiguousarray(f[:, ::-1])
    rt = RayTracer(v, f)
    o, d = secondary_rays(v, f, n_pts, group, seed=n_pts)
    assert o.shape[0] == n_pts * group
    plain = [x.clone() for x in rt.trace(o, d)]
    grouped = [x.clone() for x in rt.trace_grouped(o, d, group, heavy_from)]
    assert 0.05 < float((plain[2] < 10).float().mean()) < 0.95
    for a, b in zip(plain, grouped):
        assert torch.equal(a, b), int((a != b).sum())
    # a group size that is not a multiple of the 64-ray chunk, and a ray count that is not a multiple of the group: natural order, same result
    for g, hf, n in ((100, 50, o.shape[0]), (group, heavy_from, o.shape[0] - 64)):
        odd = rt.trace_grouped(o[:n], d[:n], g, hf)
        for a, b in zip(plain, odd):
            assert torch.equal(a[:n], b)


@pytest.mark.parametrize('mode', [0, 1])
def test_masked_trace_skips_the_flagged_rays_only(mode):
    """nero_bvh_trace_masked (round 6): a flagged ray is reported as a miss (depth 10, zero normal) without a node visit; every other ray's
    position, normal and depth are those of nero_bvh_trace bit for bit -- in both traversal kernels, with an odd ray count"""
    from nero_amd import _lib as L
    from nero_amd.raytracing import RayTracer
    from nero_amd.synthetic import icosphere, secondary_rays
    v, f = icosphere(5, 0.5, 0.2)
    f = np.ascontiguousarray(f[:, ::-1])
    rt = RayTracer(v, f)
    L.check(L.lib.nero_bvh_set_traversal(rt._handle(), mode))
    o, d = secondary_rays(v, f, 301, 64, seed=3)
    o, d = o[:-17].cuda().contiguous(), d[:-17].cuda().contiguous()
    n = o.shape[0]
    plain = [x.clone() for x in rt.trace(o, d)]
    skip = (torch.rand(n, generator=torch.Generator().manual_seed(1)) < 0.3).to(torch.uint8).cuda()
    pos, nrm, depth = rt.trace_masked(o, d, skip)
    keep = skip == 0
    assert 0.05 < float((plain[2] < 10).float().mean()) < 0.95 and int((plain[2][~keep] < 10).sum()) > 0
    for a, b in zip(plain, (pos, nrm, depth)):
        assert torch.equal(a[keep], b[keep])
    assert bool((depth[~keep] == 10.0).all()) and bool((nrm[~keep] == 0).all())
    none = rt.trace_masked(o, d, torch.zeros(n, dtype=torch.uint8, device='cuda'))
    for a, b in zip(plain, none):
        assert torch.equal(a, b)


@pytest.mark.parametrize('n_pts,k,order', [(300, 4, [3, 2, 1, 0]), (97, 12, list(range(11, -1, -1))), (64, 2, [1, 0]), (50, 4, [2, 0, 3, 1])])
def test_ordered_launch_is_bit_identical_to_the_natural_order(n_pts, k, order):
    """nero_bvh_trace_ordered (round 6): phase p of the launch holds chunk order[p] of every group of 64 k rays -- same rays, same arithmetic,
    same outputs at the same addresses as nero_bvh_trace / _masked, with and without a skip mask; an order that is not a permutation, or a ray
    count that is not a whole number of groups, falls back to the natural order"""
    from nero_amd.raytracing import RayTracer
    from nero_amd.synthetic import icosphere, secondary_rays
    v, f = icosphere(5, 0.5, 0.2)
    f = np.ascontiguousarray(f[:, ::-1])
    rt = RayTracer(v, f)
    o, d = secondary_rays(v, f, n_pts, 64 * k, seed=n_pts)
    o, d = o.cuda().contiguous(), d.cuda().contiguous()
    n = o.shape[0]
    plain = [x.clone() for x in rt.trace(o, d)]
    assert 0.05 < float((plain[2] < 10).float().mean()) < 0.95
    got = rt.trace_masked(o, d, None, chunk_order=order)
    for a, b in zip(plain, got):
        assert torch.equal(a, b), int((a != b).sum())
    skip = (torch.rand(n, generator=torch.Generator().manual_seed(k)) < 0.25).to(torch.uint8).cuda()
    masked = [x.clone() for x in rt.trace_masked(o, d, skip)]
    both = rt.trace_masked(o, d, skip, chunk_order=order)
    for a, b in zip(masked, both):
        assert torch.equal(a, b)
    for bad, m in (([0] * k, n), (order, n - 64)):                  # not a permutation / not whole groups: natural order, same result
        odd = rt.trace_masked(o[:m], d[:m], None, chunk_order=bad)
        for a, b in zip(plain, odd):
            assert torch.equal(a[:m], b)
