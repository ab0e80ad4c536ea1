#!/bin/bash
# gap analysis of the Stage-II step (C4: 4096 points x 128+128 directions) -> gpurun_out/r04/stage2_gaps.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04 gpurun_out/prof
rm -rf gpurun_out/prof/s2g
NERO_STREAMS=${NERO_STREAMS:-1} timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/s2g -o s2 --output-format csv -- python scripts/bench_material_step.py 4096 128 128 7 bell fused > gpurun_out/r04/stage2_gaps.log 2>&1
find gpurun_out/prof/s2g -name "*kernel_trace.csv" -exec python scripts/gap_analysis.py {} \; > gpurun_out/r04/stage2_gaps.txt 2>&1
rm -rf gpurun_out/prof/s2g
tail -2 gpurun_out/r04/stage2_gaps.log; head -40 gpurun_out/r04/stage2_gaps.txt
