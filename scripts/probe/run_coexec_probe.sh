#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p $R/gpurun_out $R/build
hipcc --offload-arch=gfx950 -O3 -o $R/build/coexec_probe $R/scripts/probe/coexec_probe.hip && timeout 120 $R/build/coexec_probe > $R/gpurun_out/coexec_probe.txt 2>&1
cat $R/gpurun_out/coexec_probe.txt
