#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
rm -rf gpurun_out/prof/trace; mkdir -p gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof/trace -o t --output-format csv -- python scripts/step_times.py 4096 12 > gpurun_out/prof/trace.log 2>&1
f=$(find gpurun_out/prof/trace -name "*kernel_trace.csv" | head -1)
python scripts/gap_analysis.py $f | tee gpurun_out/gap_analysis.txt | head -40
rm -rf gpurun_out/prof/trace; mkdir -p gpurun_out/prof
