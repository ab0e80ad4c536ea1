#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.txt 2>&1; echo "pytest rc $?"
grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_gpu.txt | tail -15
for drv in py c; do NERO_STEP_DRIVER=$drv timeout 300 python scripts/bench_material_step.py 4096 128 128 7 bell fused 2>&1 | grep fused= | sed "s/^/driver=$drv /" | tee -a gpurun_out/stage2_times.txt; done
NERO_STEP_DRIVER=c timeout 300 python scripts/bench_material_step.py 2048 256 256 7 bear fused 2>&1 | grep fused= | tee -a gpurun_out/stage2_times.txt
