"""CPU tier: the drop-in renderers accept every YAML the reference ships (config surface) and expose the
reference's parameter names.  Skipped when the reference tree is not mounted (GPU box)."""
import glob
import os

import pytest
import torch
import yaml

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'configs')), reason='reference tree not available')


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(REF, 'configs', 'shape', '*', '*.yaml')) +
                                        glob.glob(os.path.join(REF, 'configs', 'custom', '*shape*.yaml'))))
def test_shape_yaml_constructs(path):
    from nero_amd.renderer import name2renderer
    cfg = yaml.load(open(path), Loader=yaml.FullLoader)
    net = name2renderer[cfg['network']](cfg, training=False)
    keys = set(net.state_dict().keys())
    assert 'deviation_network.variance' in keys and 'color_network.FG_LUT' in keys
    assert {'sdf_network.lin0.weight_g', 'sdf_network.lin8.weight_v', 'outer_nerf.pts_linears.5.weight',
            'color_network.inner_weight.6.bias'} <= keys
    assert ('color_network.human_light_predictor.0.weight_g' in keys) == bool(cfg.get('shader_config', {}).get('human_light', False))
    n_params = sum(p.numel() for p in net.parameters())
    assert n_params == (2346073 if cfg.get('shader_config', {}).get('human_light', False) else 2206289)


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(REF, 'configs', 'material', '*', '*.yaml')) +
                                        glob.glob(os.path.join(REF, 'configs', 'custom', '*material*.yaml'))))
def test_material_yaml_constructs(path):
    from nero_amd.renderer import name2renderer
    from nero_amd.synthetic import icosphere
    cfg = yaml.load(open(path), Loader=yaml.FullLoader)
    net = name2renderer[cfg['network']](cfg, is_train=False, mesh=icosphere(1))
    keys = set(net.state_dict().keys())
    assert {'shader_network.feats_network.module0.0.weight_g', 'shader_network.feats_network.module1.6.bias',
            'shader_network.inner_light.6.weight_v', 'shader_network.light_pts'} <= keys
    human = cfg['shader_cfg'].get('human_lights', True)
    assert ('shader_network.human_light.0.weight_g' in keys) == bool(human)
    n_params = sum(p.numel() for p in net.parameters())
    assert n_params == (1561886 if (human and cfg['shader_cfg']['outer_light_version'] == 'sphere_direction') else 1403670), n_params


@pytest.mark.parametrize('act', ['exp', 'linear', 'square'])
def test_std_act_matches_the_reference_variance_network(act):
    """std_act (network/field.py:184-198) -- the reference class under the import shim against the drop-in module, the oracle's
    restatement, and the log-domain scalar the HIP kernels are handed (they form exp(10 v'))"""
    from oracle import ref_shim
    from oracle import nero_oracle as O
    from nero_amd.fields import SingleVarianceNetwork
    _, field = ref_shim.load_reference()
    for v in (0.3, 0.07, 2.0):
        ref = field.SingleVarianceNetwork(v, act)
        want = float(ref(torch.zeros(1, 3))[0, 0])
        mine = SingleVarianceNetwork(v, act)
        assert abs(float(mine.inv_s()) - want) <= 1e-6 * abs(want)
        assert abs(float(O.deviation_inv_s({'deviation_network.variance': torch.tensor(v)}, act)) - want) <= 1e-6 * abs(want)
        kv = mine.kernel_variance()
        assert abs(float(torch.exp(kv * 10.0)) - want) <= 3e-6 * abs(want)
        kv.backward()                                              # d v' / d variance chains the activation's slope onto the kernels' d inv_s
        slope = {'exp': 10.0 * want, 'linear': 10.0, 'square': 200.0 * v}[act]
        assert abs(float(mine.variance.grad) * 10.0 * want - slope) <= 1e-4 * slope
    with pytest.raises(NotImplementedError):
        SingleVarianceNetwork(0.3, 'tanh')
