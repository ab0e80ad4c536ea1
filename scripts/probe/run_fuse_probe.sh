#!/bin/bash
# build + run scripts/probe/fuse_probe.hip on the GPU box; output -> gpurun_out/fuse_probe.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p $R/gpurun_out $R/build
hipcc --offload-arch=gfx950 -O3 -o $R/build/fuse_probe $R/scripts/probe/fuse_probe.hip && timeout 120 $R/build/fuse_probe > $R/gpurun_out/fuse_probe.txt 2>&1
tail -12 $R/gpurun_out/fuse_probe.txt
