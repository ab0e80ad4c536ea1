"""Import shim that lets the UNMODIFIED reference (/root/reference) execute on CPU in this container.

TEST INFRASTRUCTURE ONLY.  Used by oracle/gen_golden.py (in the build container, where /root/reference
exists) to dump golden vectors into tests/golden/.  Nothing in the product path, the -m gpu tests,
smoke() or bench.py imports this file: /root/reference does not exist on the GPU box.

What it does:
  1. stub packages for third-party imports that are absent here (cv2, open3d, trimesh, mcubes, raytracing,
     h5py, plyfile, skimage, transforms3d, tensorboardX, nvdiffrast);
  2. nvdiffrast.torch.texture restated as a pure-torch bilinear clamp fetch (texel centres at (i+.5)/W) --
     the one third-party arithmetic on the Stage-I path (network/field.py:612);
  3. numpy-2 compat (np.math, np.bool);
  4. CPU only: Tensor.cuda / Module.cuda -> identity, torch.randperm drops device='cuda'
     (network/renderer.py:537);
  5. chdir to the reference root so assets/bsdf_256_256.bin resolves (network/field.py:510).

Round 5: the reference root may also be oracle/_ref/ (oracle/make_ref.py: the same files, byte for byte, as an importable zip
+ the FG asset), which is what oracle/run_ref.py uses to TIME the unmodified reference on the GPU box (bench.py's `cpu_baseline`
kind "reference" and `reference_gpu_baseline`).  install(force_cpu=True) applies the CPU-only patches of item 4 even when a GPU
is visible (the host-core timing on the GPU box).
"""
import importlib.abc
import importlib.machinery
import math
import os
import sys
import types

import numpy as np
import torch

_PACKED = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_ref')
REF_ROOT = os.environ.get('NERO_REFERENCE_ROOT') or ('/root/reference' if os.path.isdir('/root/reference/network') else _PACKED)

_STUBS = ('mcubes', 'cv2', 'open3d', 'trimesh', 'raytracing', 'h5py', 'plyfile', 'skimage',
          'transforms3d', 'tensorboardX', 'nvdiffrast', 'xatlas')


class _Permissive(types.ModuleType):
    __path__ = []

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        full = f'{self.__name__}.{name}'
        mod = _Permissive(full)
        sys.modules[full] = mod
        setattr(self, name, mod)
        return mod

    def __call__(self, *a, **k):
        raise RuntimeError(f'stubbed third-party call {self.__name__}')


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split('.')[0] in _STUBS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _Permissive(spec.name)

    def exec_module(self, module):
        pass


def texture_bilinear_clamp(tex, uv, filter_mode='linear', boundary_mode='clamp'):
    """tex [1,H,W,C], uv [1,h,w,2] -> [1,h,w,C]; u indexes W, v indexes H; texel centres at (i+.5)/W."""
    assert filter_mode == 'linear' and boundary_mode == 'clamp'
    _, H, W, C = tex.shape
    u = uv[..., 0] * W - 0.5
    v = uv[..., 1] * H - 0.5
    u = torch.clamp(u, 0.0, W - 1.0)
    v = torch.clamp(v, 0.0, H - 1.0)
    u0 = torch.clamp(torch.floor(u), max=W - 2.0)
    v0 = torch.clamp(torch.floor(v), max=H - 2.0)
    fu = (u - u0).unsqueeze(-1)
    fv = (v - v0).unsqueeze(-1)
    u0 = u0.long()
    v0 = v0.long()
    t = tex[0]
    t00 = t[v0, u0]
    t01 = t[v0, u0 + 1]
    t10 = t[v0 + 1, u0]
    t11 = t[v0 + 1, u0 + 1]
    return (t00 * (1 - fu) + t01 * fu) * (1 - fv) + (t10 * (1 - fu) + t11 * fu) * fv


_installed = False


def install(force_cpu=False):
    global _installed
    if _installed:
        return
    _installed = True
    global _STUBS
    if not os.path.isdir(os.path.join(REF_ROOT, 'colmap')):      # the packaged reference (oracle/_ref) carries network/, utils/, dataset/ only:
        _STUBS = _STUBS + ('colmap',)                            # dataset/database.py:11-12 imports COLMAP model IO, never used on the render path
    sys.meta_path.insert(0, _Finder())
    import nvdiffrast.torch as dr  # stub
    dr.texture = texture_bilinear_clamp
    if not hasattr(np, 'math'):
        np.math = math
    if not hasattr(np, 'bool'):
        np.bool = bool
    if force_cpu or not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.nn.Module.cuda = lambda self, *a, **k: self
        _randperm = torch.randperm

        def randperm(n, *a, **k):
            k.pop('device', None)
            return _randperm(n, *a, **k)
        torch.randperm = randperm
    packed = os.path.join(REF_ROOT, 'nero_ref.zip')
    src = packed if os.path.exists(packed) else REF_ROOT
    if src not in sys.path:
        sys.path.insert(0, src)
    os.chdir(REF_ROOT)


def load_reference(force_cpu=False):
    """Returns (network.renderer, network.field) modules of the unmodified reference."""
    install(force_cpu)
    import warnings
    warnings.filterwarnings('ignore')
    import network.field as field
    import network.renderer as renderer
    return renderer, field
