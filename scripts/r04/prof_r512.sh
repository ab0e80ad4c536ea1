#!/bin/bash
# kernel statistics + gap analysis of the 512-ray step (one stream) -> gpurun_out/r04/r512_<tag>_*
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04 gpurun_out/prof
tag=${1:-cur}
export NERO_STREAMS=1
rm -rf gpurun_out/prof/r512
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/r512 -o r512 --output-format csv -- python scripts/step_times.py ${2:-512} 30 > gpurun_out/r04/r512_$tag.log 2>&1
find gpurun_out/prof/r512 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04/r512_${tag}_kernel_stats.csv \;
find gpurun_out/prof/r512 -name "*kernel_trace.csv" -exec python scripts/gap_analysis.py {} \; > gpurun_out/r04/r512_${tag}_gaps.txt 2>&1
rm -rf gpurun_out/prof/r512
tail -1 gpurun_out/r04/r512_$tag.log; head -12 gpurun_out/r04/r512_${tag}_gaps.txt
