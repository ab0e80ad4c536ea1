#!/bin/bash
# per-kernel statistics of the full step and of the glue-free lower bound (scripts/r04/glue_floor.py), one stream
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export NERO_STREAMS=1
mkdir -p gpurun_out/r04 gpurun_out/prof
for m in full bare; do
  rm -rf gpurun_out/prof/gf_$m
  GLUE_MODE=$m timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/gf_$m -o $m --output-format csv -- python scripts/r04/glue_floor.py ${1:-4096} 20 > gpurun_out/r04/gf_$m.log 2>&1
  find gpurun_out/prof/gf_$m -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04/gf_${m}_kernel_stats.csv \;
  tail -1 gpurun_out/r04/gf_$m.log
done
rm -rf gpurun_out/prof/gf_*
