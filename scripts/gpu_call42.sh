#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
timeout 1200 python -m pytest tests/test_units_gpu.py tests/test_shape_render.py tests/test_stage1_driver.py tests/test_edge_cases.py -m gpu -q -x -p no:cacheprovider -k "not stage2" 2>&1 | grep -E "^E|passed|failed|FAILED|Error" | cut -c1-400 | head -20
mkdir -p gpurun_out/prof; rm -rf gpurun_out/prof/enc
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/enc -o t --output-format csv -- python scripts/step_times.py 4096 10 > gpurun_out/prof/enc.log 2>&1
python - <<'P'
import csv, glob
f = glob.glob('gpurun_out/prof/enc/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('gather', 'encode', 'pe_', 'ray_points', 'upsample')):
        print(r['Name'].replace('(anonymous namespace)::', '')[:34], r['Calls'], round(float(r['AverageNs']) / 1e3, 1), 'us')
P
rm -rf gpurun_out/prof/enc
python scripts/step_times.py 4096 20 | tail -1
