#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p $R/gpurun_out $R/build
hipcc --offload-arch=gfx950 -O3 -o $R/build/pkfma_probe $R/scripts/probe/pkfma_probe.hip 2>/dev/null && timeout 150 $R/build/pkfma_probe > $R/gpurun_out/pkfma_probe.txt 2>&1
cat $R/gpurun_out/pkfma_probe.txt
