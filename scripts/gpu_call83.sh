#!/bin/bash
R=${GRAFT_REPO_ROOT}; cd $R && mkdir -p gpurun_out
timeout 300 python scripts/trace_bench.py gpurun_out/trace_bench.json 2>&1 | grep "^{" | cut -c1-600
bash scripts/prof_trace.sh 2>&1 | tail -30
