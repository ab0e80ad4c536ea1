"""The split-sum FG table is input data of the reference (assets/bsdf_256_256.bin, network/field.py:510-511): the product must
load THAT table by default, the way the reference does, and may fall back to its own computed table only loudly."""
import os
import warnings

import numpy as np
import pytest
import torch

from tests.helpers import load_golden, ref_fg_lut, reference_fg_asset


def _fresh_net(cfg=None):
    from nero_amd.renderer import NeROShapeRenderer
    torch.manual_seed(0)
    return NeROShapeRenderer(cfg or {}, training=False)


def test_default_is_the_reference_asset_resolved_like_the_reference(tmp_path, monkeypatch):
    """constructed with cwd = a tree holding assets/bsdf_256_256.bin (no env, no cfg key): FG_LUT is that file, bit for bit"""
    monkeypatch.delenv('NERO_FG_LUT', raising=False)
    monkeypatch.chdir(reference_fg_asset(str(tmp_path)))
    with warnings.catch_warnings():
        warnings.simplefilter('error', RuntimeWarning)
        net = _fresh_net()
    assert torch.equal(net.color_network.FG_LUT, ref_fg_lut())


@pytest.mark.skipif(not os.path.exists('/root/reference/assets/bsdf_256_256.bin'), reason='reference tree not present (GPU box)')
def test_constructed_in_the_reference_tree(monkeypatch):
    """the verdict's acceptance test: a from-scratch NeROShapeRenderer(cfg) constructed in the reference tree carries the
    reference's table (and therefore reproduces tests/golden/bell_s25000.npz through the unchanged golden tests)"""
    monkeypatch.delenv('NERO_FG_LUT', raising=False)
    monkeypatch.chdir('/root/reference')
    _, meta = load_golden('bell_s25000')
    net = _fresh_net(meta['cfg'])
    ref = np.fromfile('/root/reference/assets/bsdf_256_256.bin', dtype=np.float32).reshape(1, 256, 256, 2)
    assert np.array_equal(net.color_network.FG_LUT.numpy(), ref)
    assert np.array_equal(ref, ref_fg_lut().numpy())         # and the test fixture is that same table


def test_cfg_key_and_missing_file(tmp_path, monkeypatch):
    monkeypatch.delenv('NERO_FG_LUT', raising=False)
    monkeypatch.chdir(tmp_path)
    asset = os.path.join(reference_fg_asset(str(tmp_path / 'ref')), 'assets', 'bsdf_256_256.bin')
    net = _fresh_net({'shader_config': {'fg_lut_path': asset}})
    assert torch.equal(net.color_network.FG_LUT, ref_fg_lut())
    with pytest.raises(FileNotFoundError):
        _fresh_net({'shader_config': {'fg_lut_path': str(tmp_path / 'nope.bin')}})


def test_fallback_is_loud_and_its_distance_is_what_the_docstring_says(tmp_path, monkeypatch):
    monkeypatch.delenv('NERO_FG_LUT', raising=False)
    monkeypatch.chdir(tmp_path)                                # no assets/ here
    with pytest.warns(RuntimeWarning, match='COMPUTED split-sum table'):
        net = _fresh_net()
    d = (net.color_network.FG_LUT - ref_fg_lut()).abs()
    assert 1e-4 < float(d.mean()) < 6e-4 and 1e-2 < float(d.max()) < 2.5e-2, (float(d.mean()), float(d.max()))
