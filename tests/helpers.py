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
