#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R && mkdir -p gpurun_out
timeout 300 python scripts/trace_bench.py gpurun_out/trace_bench.json 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print(d['case'], {k: v for k, v in d.items() if k.startswith('ms_')}, {k: (v['depth_and_position_bit_identical'], v['normals_bit_identical'], v['speedup']) for k, v in d.items() if '_vs_' in k})
"
