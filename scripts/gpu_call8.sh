#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_gpu.txt 2>&1; echo "pytest rc $?"
grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_gpu.txt | tail -8
python scripts/prof_launches.py 4096 > gpurun_out/prof_launches_new.txt 2>&1; grep -E "bwd|tan|^\{" gpurun_out/prof_launches_new.txt
NERO_BWD_INJ_EPILOGUE=1 python scripts/prof_launches.py 4096 > gpurun_out/prof_launches_injepi.txt 2>&1; grep -E "bwd|^\{" gpurun_out/prof_launches_injepi.txt | tail -4
timeout 200 python scripts/step_times.py 4096 30 2>&1 | grep driver=
