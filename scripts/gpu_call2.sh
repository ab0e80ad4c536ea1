#!/bin/bash
# round-3 GPU call 2: the C-level step driver -- whole GPU test tier, then same-box timing C driver vs Python driver at 4096 / 512 rays
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --durations=12 -p no:cacheprovider > gpurun_out/pytest_gpu.txt 2>&1; echo "pytest rc $?"
tail -25 gpurun_out/pytest_gpu.txt
for drv in py c; do
  for R in 4096 512; do
    NERO_STEP_DRIVER=$drv timeout 200 python scripts/step_times.py $R 30 2>&1 | grep driver= | tee -a gpurun_out/step_times.txt
  done
done
NERO_STEP_DRIVER=c timeout 200 python scripts/step_times.py 1024 30 bear 2>&1 | grep driver= | tee -a gpurun_out/step_times.txt
