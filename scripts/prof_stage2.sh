#!/bin/bash
# kernel trace of the Stage-II material step (BASELINE configs[3]: P = 4096 x 128+128 directions); summary -> gpurun_out/prof/stage2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/stage2 -o stage2 --output-format csv -- python scripts/bench_material.py 4096 128 128 7 > gpurun_out/prof/stage2.log 2>&1
tail -3 gpurun_out/prof/stage2.log
