#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/prof; rm -rf gpurun_out/prof/enc gpurun_out/prof/enc2
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/enc -o t --output-format csv -- python scripts/step_times.py 4096 10 > gpurun_out/prof/enc.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/enc2 -o t --output-format csv -- python scripts/bench_material_step.py 4096 128 128 7 bell fused > gpurun_out/prof/enc2.log 2>&1
python - <<'P'
import csv, glob
for d in ('enc', 'enc2'):
    f = glob.glob(f'gpurun_out/prof/{d}/**/*kernel_stats.csv', recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if any(k in r['Name'] for k in ('encode', 'pe_vjp', 'pe_jvp', 'mc_dir_bwd', 'trace_kernel', 'mc_combine')):
            print(d, r['Name'].replace('(anonymous namespace)::', '')[:34], r['Calls'], round(float(r['AverageNs']) / 1e3, 1), 'us')
P
rm -rf gpurun_out/prof/enc gpurun_out/prof/enc2
