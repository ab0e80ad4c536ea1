#!/bin/bash
# r512: where does the 6.9 ms step go?  per-kernel stats + idle gaps
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/prof; rm -rf gpurun_out/prof/r512
python scripts/step_times.py 512 30 | tail -1
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/r512 -o t --output-format csv -- python scripts/step_times.py 512 30 > gpurun_out/prof/r512.log 2>&1
f=$(find gpurun_out/prof/r512 -name "*kernel_trace.csv" | head -1)
python scripts/gap_analysis.py $f > gpurun_out/gap_analysis_r512.txt; head -12 gpurun_out/gap_analysis_r512.txt
s=$(find gpurun_out/prof/r512 -name "*kernel_stats.csv" | head -1)
cp $s gpurun_out/r512_kernel_stats.csv
rm -rf gpurun_out/prof/r512
head -25 gpurun_out/r512_kernel_stats.csv | cut -c1-150
