#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 1200 python -m pytest tests/test_units_gpu.py tests/test_shape_render.py tests/test_material_render.py tests/test_stage1_driver.py tests/test_stage2_driver.py tests/test_material_train.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -E "^E|passed|failed|FAILED|Error" | cut -c1-400 | head -20
python scripts/step_times.py 4096 20 | tail -1
python scripts/step_times.py 4096 20 bear | tail -1
python scripts/bench_material_step.py 4096 128 128 7 bell fused | tail -1
python scripts/bench_material_step.py 2048 256 256 7 bear fused | tail -1
mkdir -p gpurun_out/prof; rm -rf gpurun_out/prof/enc
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/enc -o t --output-format csv -- python scripts/step_times.py 4096 10 > gpurun_out/prof/enc.log 2>&1
s=$(find gpurun_out/prof/enc -name "*kernel_stats.csv" | head -1)
grep -E "encode|pe_vjp|pe_jvp" $s | cut -d, -f1-4 | sed 's/(anonymous namespace):://' | cut -c1-40,170-
rm -rf gpurun_out/prof/enc
