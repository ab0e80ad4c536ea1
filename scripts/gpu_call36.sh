#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 1200 python -m pytest tests/test_units_gpu.py tests/test_shape_render.py tests/test_material_render.py tests/test_stage1_driver.py tests/test_stage2_driver.py tests/test_edge_cases.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -E "^E|passed|failed|FAILED|Error" | cut -c1-400 | head -20
bash scripts/gpu_call34.sh 2>&1 | grep "^enc"
python scripts/step_times.py 4096 20 | tail -1
python scripts/bench_material_step.py 4096 128 128 7 bell fused | tail -1
