"""Round-5 golden vectors, dumped from the UNMODIFIED reference under oracle/ref_shim.py (build container only):

  python oracle/gen_golden_r5.py

  tests/golden/bell_shape_keys.npz (+ _grads.npz)   a Stage-I render case at NON-YAML network-shape keys -- sdf_n_layers 6 (skip into layer 3),
                               sdf_freq 4 (27 input columns), shader_config.light_pos_freq 6 (39 position columns in front of the indirect-light
                               and occlusion MLPs), sdf_activation 'sigmoid' (accepted and never read by the reference: network/field.py:72) --
                               for network/renderer.py:73-76,118-124 and network/field.py:515 (VERDICT r4 'missing' 4)
  tests/golden/bell_deep_sdf.npz                    sdf_n_layers 9 (the deepest a chain descriptor holds), sdf_freq 5, light_pos_freq 10

TEST INFRASTRUCTURE ONLY: /root/reference does not exist on the GPU box; the committed fixtures are what travels.
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle.gen_golden import run_case  # noqa: E402

if __name__ == '__main__':
    small = dict(n_samples=16, n_importance=16, n_bg_samples=8, up_sample_steps=4)
    run_case('bell_shape_keys', dict(small, sdf_n_layers=6, sdf_freq=4, sdf_activation='sigmoid', shader_config={'light_pos_freq': 6}),
             R=48, step=25000, variance=0.4)
    run_case('bell_deep_sdf', dict(small, sdf_n_layers=9, sdf_freq=5, shader_config={'light_pos_freq': 10}), R=48, step=25000, variance=0.4)
    from oracle import gen_golden_grads as G
    G.dump('bell_shape_keys', G.shape_case('bell_shape_keys'))
