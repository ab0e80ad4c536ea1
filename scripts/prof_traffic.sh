#!/bin/bash
# HBM traffic counters of the Stage-I and Stage-II training steps (separate --pmc passes, as the microarch guide prescribes);
# raw counter CSVs -> gpurun_out/prof/{fetch,write}[_stage2]/, summarised afterwards by scripts/summarize_traffic.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
for c in FETCH_SIZE:fetch WRITE_SIZE:write; do
  cn=${c%%:*}; d=${c##*:}
  rm -rf gpurun_out/prof/$d gpurun_out/prof/${d}_stage2
  timeout 400 rocprofv3 --pmc $cn --kernel-trace -d gpurun_out/prof/$d -o $d --output-format csv -- python scripts/step_times.py 4096 6 > gpurun_out/prof/$d.log 2>&1
  timeout 400 rocprofv3 --pmc $cn --kernel-trace -d gpurun_out/prof/${d}_stage2 -o $d --output-format csv -- python scripts/bench_material_step.py 4096 128 128 7 bell fused > gpurun_out/prof/${d}_stage2.log 2>&1
done
rm -rf gpurun_out/prof/*/*.db gpurun_out/prof/*/*kernel_trace.csv gpurun_out/prof/*/*/*.db gpurun_out/prof/*/*/*kernel_trace.csv
du -sh gpurun_out/prof/* | tail -8
