#!/bin/bash
# short kernel trace (3 timed steps) for a quick look at the small kernels (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/short -o short --output-format csv -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/prof/short.log 2>&1
grep -E "shade_encode|human_encode|head_dw|upsample|merge_sorted|dw_reduce" gpurun_out/prof/short/short_kernel_stats.csv | cut -d, -f1-4 | cut -c1-110
