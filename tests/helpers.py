"""Shared test helpers: rebuild the golden cases' weights / inputs (tests/golden/*.npz, made by oracle/gen_golden.py)."""
import json
import os

import numpy as np
import torch

from nero_amd.synthetic import perturb_state

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    meta = json.loads(str(z['meta']))
    return z, meta


def ref_fg_lut():
    return torch.from_numpy(np.load(os.path.join(GOLDEN, 'fg_lut_ref.npz'))['lut']).reshape(1, 256, 256, 2)


def build_case_model(meta, device='cpu'):
    """seed -> construct -> perturb: the same recipe oracle/gen_golden.py applied to the reference."""
    from nero_amd.renderer import NeROShapeRenderer
    torch.manual_seed(meta['seed'])
    net = NeROShapeRenderer(meta['cfg'], training=False)
    perturb_state(net, meta['variance'])
    with torch.no_grad():
        net.color_network.FG_LUT.copy_(ref_fg_lut())
    return net.to(device)


def T(z, k, device='cpu'):
    return torch.from_numpy(np.asarray(z[k])).to(device)


class MatHolder(torch.nn.Module):
    """state_dict prefix `shader_network.` like NeROMaterialRenderer (network/renderer.py:699-701)"""

    def __init__(self, shader_cfg):
        super().__init__()
        from nero_amd.fields import MCShadingNetwork
        self.shader_network = MCShadingNetwork(shader_cfg)


def build_material_case(meta, device='cpu'):
    torch.manual_seed(meta['seed'])
    net = MatHolder(meta['shader_cfg'])
    perturb_state(net, None)
    return net.to(device)


def golden_mesh():
    from nero_amd.synthetic import icosphere
    v, f = icosphere(3, 0.5, 0.15)
    return v, np.ascontiguousarray(f[:, ::-1])       # inward winding (see oracle/gen_golden.py::run_material_case)


def oracle_trace_fn():
    """NeROMaterialRenderer.trace contract (network/renderer.py:719-729) over the brute-force tracer oracle"""
    from oracle.tracer_oracle import trace_bruteforce
    verts, tris = golden_mesh()

    def trace(o, d):
        pos, nrm, depth, _ = trace_bruteforce(verts, tris, o.detach().cpu().numpy(), d.detach().cpu().numpy())
        nrm = torch.nn.functional.normalize(torch.from_numpy(-nrm).float(), dim=-1)
        depth = torch.from_numpy(depth).float().reshape(-1, 1)
        return torch.from_numpy(pos).float(), nrm, depth, (depth < 10)[:, 0]
    return trace
