#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_engines_extra.py tests/test_stage1_driver.py tests/test_stage2_driver.py tests/test_mlp_engine.py tests/test_material_train.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -E "^E|passed|failed|FAILED|Error" | cut -c1-400 | head -20
python scripts/step_times.py 512 30 | tail -1
python scripts/step_times.py 1024 30 bear | tail -1
python scripts/step_times.py 4096 20 | tail -1
python scripts/bench_material_step.py 4096 128 128 7 bell fused | tail -1
python scripts/bench_material_step.py 2048 256 256 7 bear fused | tail -1
