#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 1200 python -m pytest tests/test_determinism.py tests/test_shape_render.py tests/test_engines_extra.py tests/test_stage1_driver.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "^E|passed|failed|FAILED|Error" | cut -c1-300 | head -20
python scripts/determinism.py 4096 12 2>&1 | tail -2
python scripts/step_times.py 4096 20 | tail -1
python scripts/step_times.py 512 30 | tail -1
python scripts/bench_material_step.py 4096 128 128 7 bell fused | tail -1
NERO_REPLAY_ENGINE=f16x3 python scripts/replay_fwd_chain.py 2>&1 | grep "launches differing" | cut -c1-120
