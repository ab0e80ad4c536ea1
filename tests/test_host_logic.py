"""CPU tier: host-side logic of the drop-in renderers against vectors dumped from the UNMODIFIED reference
(oracle/gen_golden_r2.py -> tests/golden/pools.npz, ref_state.json; oracle/gen_golden.py -> the render cases):
a1 helpers (get_human_coordinate_poses, _process_ray_batch, near_far_from_sphere; network/renderer.py:230-272), the Stage-I ray
pool (network/renderer.py:167-187), the state_dict surface, and the C restatement of the tracer oracle."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pools():
    return np.load(os.path.join(GOLDEN, 'pools.npz'))


def _shape_net(cfg=None):
    from nero_amd.renderer import NeROShapeRenderer
    torch.manual_seed(0)
    return NeROShapeRenderer(cfg or {}, training=False)


def test_human_coordinate_poses_vs_reference():
    z = _pools()
    net = _shape_net()
    poses = torch.from_numpy(z['poses'])
    assert np.abs(net.get_human_coordinate_poses(poses).numpy() - z['human_poses_img']).max() < 1e-6
    assert np.abs(_shape_net({'fixed_camera': True}).get_human_coordinate_poses(poses).numpy() - z['human_poses_img_fixed']).max() < 1e-6
    assert torch.equal(poses, torch.from_numpy(z['poses']))                      # the input is not modified
    # and the per-ray frames stored with every render case
    for name in ('bell_s25000', 'bear_s25000', 'bell_val'):
        g, _ = load_golden(name)
        if 'poses_img' in g.files:
            hp = net.get_human_coordinate_poses(torch.from_numpy(g['poses_img']))
            assert np.abs(hp.numpy() - g['human_poses']).max() < 1e-6, name


def test_process_ray_batch_vs_reference():
    z = _pools()
    net = _shape_net()
    sel = torch.from_numpy(z['s1/sel'])
    batch = {'dirs': torch.from_numpy(z['s1/dirs'])[sel], 'idxs': torch.from_numpy(z['s1/idxs'])[sel]}      # idxs [n,1] like the reference
    poses = torch.from_numpy(z['poses'])
    ro, rd, near, far, hp = net._process_ray_batch(batch, poses)
    for got, key in ((ro, 'rays_o'), (rd, 'rays_d'), (near, 'near'), (far, 'far'), (hp, 'human_poses')):
        assert got.shape == z['s1/' + key].shape, key
        assert np.abs(got.numpy() - z['s1/' + key]).max() < 2e-6, key
    # cached per-image frames give the same answer
    hp2 = net._process_ray_batch(batch, poses, net.get_human_coordinate_poses(poses))[4]
    assert torch.equal(hp, hp2)


def test_stage1_pool_contents_vs_reference_construct_ray_batch():
    """set_ray_pool (device-side pool build) == NeROShapeRenderer._construct_ray_batch of the reference, before the shuffle"""
    z = _pools()
    net = _shape_net()
    net._shuffle_train_batch = lambda: setattr(net, 'train_batch_i', 0)           # keep construction order
    net.set_ray_pool(torch.from_numpy(z['imgs']), torch.from_numpy(z['Ks']), torch.from_numpy(z['poses']), device='cpu')
    assert net.tbn == z['s1/dirs'].shape[0]
    assert np.abs(net.train_batch['dirs'].numpy() - z['s1/dirs']).max() < 2e-6
    assert np.array_equal(net.train_batch['rgbs'].numpy(), z['s1/rgbs'])
    assert np.array_equal(net.train_batch['idxs'].numpy(), z['s1/idxs'][:, 0])
    assert np.abs(net._train_human_poses.numpy() - z['human_poses_img']).max() < 1e-6


def test_shuffle_is_a_permutation_and_epoch_wraps():
    z = _pools()
    net = _shape_net({'train_ray_num': 100})
    net.set_ray_pool(torch.from_numpy(z['imgs']), torch.from_numpy(z['Ks']), torch.from_numpy(z['poses']), device='cpu')
    key = lambda a: np.sort((a * 1e4).round().astype(np.int64).view([('', np.int64)] * 3), axis=0)
    assert np.array_equal(key(net.train_batch['rgbs'].numpy()), key(z['s1/rgbs']))
    assert not np.array_equal(net.train_batch['rgbs'].numpy(), z['s1/rgbs'])


@pytest.mark.parametrize('which,cfg', [('shape_bell', {}), ('shape_bear', {'shader_config': {'human_light': True}}),
                                       ('shape_sphdir', {'shader_config': {'sphere_direction': True}})])
def test_state_dict_surface_matches_reference_constructor(which, cfg):
    """keys, shapes and dtypes of the reference constructor's state_dict; a dict with that surface loads with strict=True"""
    man = json.load(open(os.path.join(GOLDEN, 'ref_state.json')))[which]
    net = _shape_net(cfg)
    sd = net.state_dict()
    assert set(sd.keys()) == set(man.keys())
    for k, (shape, dtype) in man.items():
        assert list(sd[k].shape) == shape and str(sd[k].dtype) == dtype, k
    fake = {k: torch.full(shape, 0.25, dtype=getattr(torch, dtype.split('.')[1])) for k, (shape, dtype) in man.items()}
    net.load_state_dict(fake, strict=True)
    assert float(net.color_network.FG_LUT.mean()) == 0.25            # buffers are restored from the checkpoint too


@pytest.mark.parametrize('which,cfg', [('material_bell', {'human_lights': False, 'outer_light_version': 'direction'}),
                                       ('material_bear', {'human_lights': True, 'outer_light_version': 'sphere_direction'})])
def test_material_state_dict_surface(which, cfg):
    from tests.helpers import MatHolder
    man = json.load(open(os.path.join(GOLDEN, 'ref_state.json')))[which]
    net = MatHolder(cfg)
    sd = net.state_dict()
    assert set(sd.keys()) == set(man.keys())
    for k, (shape, dtype) in man.items():
        assert list(sd[k].shape) == shape and str(sd[k].dtype) == dtype, k
    net.load_state_dict({k: torch.zeros(shape, dtype=getattr(torch, dtype.split('.')[1])) for k, (shape, dtype) in man.items()}, strict=True)


@pytest.mark.skipif(not os.path.exists('/root/reference/network/renderer.py'), reason='reference tree not present (GPU box)')
def test_reference_constructed_state_dict_loads_strict():
    """a state_dict produced by the REFERENCE constructor (through the import shim, in a subprocess: the shim patches torch and
    changes the working directory) loads into nero_amd with strict=True and yields identical effective weights"""
    code = (
        "import sys, torch; sys.path.insert(0, %r)\n"
        "from oracle import ref_shim\n"
        "renderer, field = ref_shim.load_reference()\n"
        "torch.manual_seed(11)\n"
        "ref = renderer.NeROShapeRenderer({'shader_config': {'human_light': True}}, training=False)\n"
        "sd = ref.state_dict()\n"
        "from nero_amd.renderer import NeROShapeRenderer\n"
        "torch.manual_seed(99)\n"
        "net = NeROShapeRenderer({'shader_config': {'human_light': True}}, training=False)\n"
        "r = net.load_state_dict(sd, strict=True)\n"
        "assert not r.missing_keys and not r.unexpected_keys\n"
        "W_ref = ref.sdf_network.lin3.weight if hasattr(ref.sdf_network.lin3, 'weight') else None\n"
        "W = net.sdf_network.effective()[3][0]\n"
        "g, v = sd['sdf_network.lin3.weight_g'], sd['sdf_network.lin3.weight_v']\n"
        "assert torch.allclose(W, g * v / v.norm(dim=1, keepdim=True), atol=1e-7)\n"
        "assert torch.equal(net.color_network.FG_LUT, sd['color_network.FG_LUT'])\n"
        "print('STRICT_OK', len(sd))\n" % ROOT)
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
    assert 'STRICT_OK 137' in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_tracer_oracle_c_restatement_equals_numpy_oracle():
    from nero_amd.synthetic import icosphere
    from oracle.tracer_oracle import trace_bruteforce, trace_bruteforce_margins
    v, f = icosphere(3, 0.5, 0.15)
    f = np.ascontiguousarray(f[:, ::-1])
    rg = np.random.default_rng(0)
    o = rg.normal(size=(1500, 3))
    o = o / np.linalg.norm(o, axis=1, keepdims=True) * 2
    d = -o / 2 + rg.normal(size=(1500, 3)) * 0.2
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    a = trace_bruteforce(v, f, o, d)
    b = trace_bruteforce_margins(v, f, o, d)
    assert np.array_equal(a[3], b[3]) and np.abs(a[2] - b[2]).max() < 1e-12 and np.abs(a[1] - b[1]).max() < 1e-12
    assert np.abs(a[0] - b[0]).max() < 1e-12
    assert 0.3 < (b[3] >= 0).mean() < 1.0 and b[4].mean() < 0.01


def test_secondary_ray_generator_for_the_tracer_tests():
    """nero_amd.synthetic.secondary_rays / camera_rays (inputs of tests/test_tracer.py and scripts/trace_bench.py): unit directions, point-major
    layout, origins 1e-3 off the surface on the outside, every direction in the outer half space of its triangle; against the brute-force
    oracle: a bumpy mesh is hit by some of them and a camera ray through the centre hits the front of the mesh"""
    from nero_amd.synthetic import camera_rays, icosphere, secondary_rays
    from oracle.tracer_oracle import trace_bruteforce
    v, f = icosphere(3, 0.5, 0.2)
    f = np.ascontiguousarray(f[:, ::-1])
    P, D = 40, 16
    o, d = secondary_rays(v, f, P, D, seed=3, device='cpu')
    assert o.shape == (P * D, 3) and d.shape == (P * D, 3) and o.dtype == torch.float32
    assert float((d.norm(dim=-1) - 1).abs().max()) < 1e-5
    ob = o.reshape(P, D, 3)
    assert float((ob - ob[:, :1]).abs().max()) == 0.0                         # the D rays of a point share its origin
    on, dn = o.numpy().astype(np.float64), d.numpy().astype(np.float64)
    # the origin is 1e-3 from the nearest triangle plane on its outer side: a ray straight back down hits at t = 1e-3
    tri = v[f].astype(np.float64)
    c = tri.mean(1)
    k = np.argmin(np.linalg.norm(c[None] - on[::D, None], axis=-1), axis=1)
    n = np.cross(tri[k, 1] - tri[k, 0], tri[k, 2] - tri[k, 0])
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    n = np.where((n * c[k]).sum(1, keepdims=True) < 0, -n, n)
    assert np.abs(((on[::D] - c[k]) * n).sum(1) - 1e-3).max() < 1e-6
    assert ((dn.reshape(P, D, 3) * n[:, None]).sum(-1) > -1e-6).all()
    hit = trace_bruteforce(v, f, on, dn)[3] >= 0
    assert 0.02 < hit.mean() < 0.9
    co, cd = camera_rays(9, device='cpu')
    assert co.shape == (81, 3) and float((cd.norm(dim=-1) - 1).abs().max()) < 1e-6
    pos, nrm, depth, idx = trace_bruteforce(v, f, co.numpy().astype(np.float64), cd.numpy().astype(np.float64))
    assert idx[40] >= 0 and 1.0 < depth[40] < 2.0 and pos[40, 2] > 0               # the central ray: camera at z = 2, front of the mesh


def test_bench_gpus_flag_spawns_one_rank_per_gpu(monkeypatch):
    """`python bench.py --gpus N` without a torch.distributed environment re-launches itself under torch.distributed.run with
    N ranks on 127.0.0.1 (the flag used to be parsed and ignored)"""
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module('bench')
    calls = {}
    monkeypatch.setattr(bench.torch.cuda, 'device_count', lambda: 8)
    monkeypatch.setattr(bench.subprocess, 'call', lambda cmd, env=None: calls.update(cmd=cmd, env=env) or 0)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '8', '--steps', '3', '--warmup', '1'])
    assert bench.spawn_ranks(8) == 0
    cmd = calls['cmd']
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1'] and '--nproc-per-node=8' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[-6:] == ['--gpus', '8', '--steps', '3', '--warmup', '1']
    assert calls['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    monkeypatch.setattr(bench.torch.cuda, 'device_count', lambda: 1)
    assert bench.spawn_ranks(2) == 2                                   # refuses to fake ranks it has no GPU for
    # the result line of a distributed run carries one record per rank (device, UUID, its own ms per step) and the backend that ran
    src = open(os.path.join(ROOT, 'bench.py')).read()
    for field in ("'rccl_ranks'", "'backend'", "'ranks'", "'uuid'", "'ms_per_step'", 'all_gather_object', "init_process_group('nccl')",
                  'FORCE_COLLECTIVES = True'):
        assert field in src, field


def test_forced_collectives_run_at_world_one_and_device_count_weights_match_the_host_version():
    """bench.py under torch.distributed.run with ONE rank sets parallel.FORCE_COLLECTIVES: the flat all-reduce and the count weights
    are then issued (gloo here) although world == 1 and must be the identity; device_count_weights equals global_count_weights"""
    import torch.distributed as dist
    from nero_amd import parallel
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(29650 + os.getpid() % 300))
    dist.init_process_group('gloo', rank=0, world_size=1)
    try:
        p = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7))]
        b = parallel.GradBucket(p)
        b.flat.copy_(torch.arange(b.flat.numel(), dtype=torch.float32))
        want = b.flat.clone()
        parallel.FORCE_COLLECTIVES = True
        b.all_reduce_mean(1)
        assert torch.equal(b.flat, want)
        w = parallel.device_count_weights([1234, 0], 1, 'cpu')
        assert w.dtype == torch.float32 and w.tolist() == [1.0, 1.0]
        # a count that lives on the device (the occlusion-loss candidates of the HIP-glued step, an int32 element) beside a host count
        w = parallel.device_count_weights([1234, torch.tensor([7, 99], dtype=torch.int32)[0]], 1, 'cpu')
        assert w.dtype == torch.float32 and w.tolist() == [1.0, 1.0] and parallel.parallel_forced()
        parallel.FORCE_COLLECTIVES = False
        assert parallel.device_count_weights([10, 3], 1, 'cpu').tolist() == parallel.global_count_weights([10, 3], 1, 'cpu')
    finally:
        parallel.FORCE_COLLECTIVES = False
        dist.destroy_process_group()
