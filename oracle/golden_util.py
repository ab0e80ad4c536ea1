"""Helpers shared by oracle/gen_golden.py and tests/ (test infrastructure only)."""
import numpy as np
import torch

from nero_amd.synthetic import perturb_state, synthetic_rays  # noqa: F401  (re-export)


def state_checksums(sd):
    out = {}
    for k, v in sd.items():
        v = v.detach().double().reshape(-1)
        out[k] = np.array([float(v.sum()), float(v.abs().sum())])
    return out


def grad_digest(grads, n=64):
    """per tensor: [sum, abs-sum, l2] (float64) and a strided sample of <= n entries."""
    out = {}
    for k, g in grads.items():
        g = g.detach().double().reshape(-1)
        stride = max(1, g.numel() // n)
        out[k + '#s'] = np.array([float(g.sum()), float(g.abs().sum()), float(g.norm())])
        out[k + '#v'] = g[::stride][:n].numpy().astype(np.float32)
    return out
