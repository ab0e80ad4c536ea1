"""Round-2 golden vectors, dumped from the UNMODIFIED reference under oracle/ref_shim.py (build container only):

  python oracle/gen_golden_r2.py

  tests/golden/pools.npz      ray-pool construction of both stages on a 3-view 12x10 toy database:
                               Stage I  NeROShapeRenderer._construct_ray_batch + _process_ray_batch (network/renderer.py:167-187, 258-272)
                               Stage II NeROMaterialRenderer._construct_ray_batch (network/renderer.py:756-802) behind the brute-force
                               tracer oracle (the third-party tracer is absent), + get_human_coordinate_poses of both classes
  tests/golden/ref_state.json  state_dict keys / shapes / dtypes of the reference constructors for the four shipped config families
  tests/golden/bell_noclip_l1.npz, bell_l2.npz, bell_smoothl1.npz
                               three more Stage-I render cases (oracle/gen_golden.py::run_case) for branches round 1 left unpinned:
                               clip_sample_variance = False (network/renderer.py:434-438) and the rgb_loss kinds l1 / l2 / smooth_l1
                               (network/renderer.py:332-344)

TEST INFRASTRUCTURE ONLY: /root/reference does not exist on the GPU box; the committed fixtures are what travels.
"""
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import ref_shim  # noqa: E402
from nero_amd.synthetic import icosphere, look_at_pose  # noqa: E402

OUT = os.path.join(REPO, 'tests', 'golden')


def toy_views(h=10, w=12):
    """3 views of the unit-sphere scene: imgs [3,h,w,3] in [0,1], Ks [3,3,3], poses [3,3,4] (float32 numpy)"""
    rg = np.random.default_rng(7)
    imgs = rg.uniform(0, 1, (3, h, w, 3)).astype(np.float32)
    Ks = np.stack([np.array([[14.0 + i, 0, w / 2], [0, 15.0 - i, h / 2], [0, 0, 1]], np.float32) for i in range(3)], 0)
    poses = np.stack([look_at_pose(np.array(c, dtype=np.float64)) for c in ([3, 0.2, 0.5], [0.3, 3, 1.0], [-2, -2, 1.5])], 0)
    return imgs, Ks, poses.astype(np.float32)


def pools():
    renderer, field = ref_shim.load_reference()
    from oracle.tracer_oracle import trace_bruteforce
    imgs, Ks, poses = toy_views()
    info = {'imgs': torch.from_numpy(imgs).permute(0, 3, 1, 2), 'Ks': torch.from_numpy(Ks), 'poses': torch.from_numpy(poses)}
    rec = dict(imgs=imgs, Ks=Ks, poses=poses)

    # ---- Stage I -------------------------------------------------------------------------------------------------
    s1 = renderer.NeROShapeRenderer.__new__(renderer.NeROShapeRenderer)
    torch.nn.Module.__init__(s1)
    s1.cfg = dict(renderer.NeROShapeRenderer.default_cfg)
    batch, bposes, rn, h, w = s1._construct_ray_batch(info)
    rec.update({'s1/dirs': batch['dirs'].numpy(), 's1/rgbs': batch['rgbs'].numpy(), 's1/idxs': batch['idxs'].numpy()})
    # get_human_coordinate_poses writes through an expanded tensor (renderer.py:249): legal per pose only on CPU
    hp_img = torch.cat([s1.get_human_coordinate_poses(bposes[i:i + 1].clone()) for i in range(3)], 0)
    rec['human_poses_img'] = hp_img.numpy()
    _orig = s1.get_human_coordinate_poses
    s1.get_human_coordinate_poses = lambda p: torch.cat([_orig(p[i:i + 1].clone()) for i in range(p.shape[0])], 0)
    sel = torch.from_numpy(np.random.default_rng(3).permutation(rn)[:64])
    ro, rd, near, far, hp = s1._process_ray_batch({k: v[sel] for k, v in batch.items()}, bposes.float())
    rec.update({'s1/sel': sel.numpy(), 's1/rays_o': ro.numpy(), 's1/rays_d': rd.numpy(), 's1/near': near.numpy(), 's1/far': far.numpy(),
                's1/human_poses': hp.numpy()})
    s1.cfg['fixed_camera'] = True
    rec['human_poses_img_fixed'] = torch.cat([_orig(bposes[i:i + 1].clone()) for i in range(3)], 0).numpy()

    # ---- Stage II ------------------------------------------------------------------------------------------------
    verts, tris = icosphere(3, 0.5, 0.15)
    tris = np.ascontiguousarray(tris[:, ::-1])                # inward winding, as in gen_golden.run_material_case

    class _Tracer:                                            # raytracing.RayTracer.trace contract (raytracing/raytracer.py:21-55)
        def trace(self, o, d):
            pos, nrm, depth, _ = trace_bruteforce(verts, tris, o.detach().numpy(), d.detach().numpy())
            return torch.from_numpy(pos).float(), torch.from_numpy(nrm).float(), torch.from_numpy(depth).float()
    s2 = renderer.NeROMaterialRenderer.__new__(renderer.NeROMaterialRenderer)
    torch.nn.Module.__init__(s2)
    s2.cfg = dict(renderer.NeROMaterialRenderer.default_cfg)
    s2.ray_tracer = _Tracer()
    s2.warned_normal = False
    _orig2 = s2.get_human_coordinate_poses
    s2.get_human_coordinate_poses = lambda p: torch.cat([_orig2(p[i:i + 1].clone()) for i in range(p.shape[0])], 0)
    b2 = s2._construct_ray_batch(info, 'cpu', True)
    for k, v in b2.items():
        rec['s2/' + k] = v.numpy()
    bt = s2._construct_ray_batch({k: v[1:2] for k, v in info.items()}, 'cpu', False)
    for k, v in bt.items():
        rec['s2t/' + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, 'pools.npz'), **rec)
    print('pools ok: stage I', rn, 'rays; stage II', b2['rays_o'].shape[0], 'hits of', rn)


def state_manifest():
    renderer, field = ref_shim.load_reference()
    out = {}

    def manifest(m):
        return {k: [list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()}
    torch.manual_seed(0)
    out['shape_bell'] = manifest(renderer.NeROShapeRenderer({}, training=False))
    out['shape_bear'] = manifest(renderer.NeROShapeRenderer({'shader_config': {'human_light': True}}, training=False))
    out['shape_sphdir'] = manifest(renderer.NeROShapeRenderer({'shader_config': {'sphere_direction': True}}, training=False))

    class Holder(torch.nn.Module):
        pass
    for name, cfg in (('material_bell', {'human_lights': False, 'outer_light_version': 'direction'}),
                      ('material_bear', {'human_lights': True, 'outer_light_version': 'sphere_direction'})):
        h = Holder()
        h.shader_network = field.MCShadingNetwork(cfg, lambda o, d: None)
        out[name] = manifest(h)
    with open(os.path.join(OUT, 'ref_state.json'), 'w') as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print('state manifest ok:', {k: len(v) for k, v in out.items()})


def extra_render_cases():
    from oracle.gen_golden import run_case
    small = dict(n_samples=16, n_importance=16, n_bg_samples=8, up_sample_steps=4)
    run_case('bell_noclip_l1', dict(small, clip_sample_variance=False, rgb_loss='l1'), R=48, step=25000, variance=0.45)
    run_case('bell_l2', dict(small, rgb_loss='l2'), R=48, step=25000, variance=0.3)
    run_case('bell_smoothl1', dict(small, rgb_loss='smooth_l1'), R=48, step=25000, variance=0.35)


if __name__ == '__main__':
    pools()
    state_manifest()
    extra_render_cases()
