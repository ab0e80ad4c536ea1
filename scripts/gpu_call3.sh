#!/bin/bash
# kernel-trace statistics of the fused training step at 4096 and 512 rays (C driver)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/prof
for rays in 4096 512; do
  rm -rf gpurun_out/prof/step$rays
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/step$rays -o step --output-format csv -- python scripts/step_times.py $rays 30 > gpurun_out/prof/step$rays.log 2>&1
  grep driver= gpurun_out/prof/step$rays.log
  find gpurun_out/prof/step$rays -name "*kernel_stats.csv" | head -2
done
rm -rf gpurun_out/prof/*/*.db gpurun_out/prof/*/*/*.db gpurun_out/prof/*/*kernel_trace.csv gpurun_out/prof/*/*/*kernel_trace.csv
du -sh gpurun_out/prof
