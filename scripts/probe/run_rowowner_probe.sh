#!/bin/bash
# build + run scripts/probe/rowowner_probe.hip on the GPU box; output -> gpurun_out/rowowner_probe.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p $R/gpurun_out $R/build/probe
hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o $R/build/probe/rowowner_probe $R/scripts/probe/rowowner_probe.hip 2>/dev/null || exit 1
{ timeout 120 $R/build/probe/rowowner_probe 1024; timeout 120 $R/build/probe/rowowner_probe 32; timeout 120 $R/build/probe/rowowner_probe 1024; } > $R/gpurun_out/rowowner_probe.txt 2>&1
cat $R/gpurun_out/rowowner_probe.txt
