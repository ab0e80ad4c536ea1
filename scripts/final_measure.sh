#!/bin/bash
# the round's measurement pass (run on the GPU box): bench line, kernel stats of the Stage-I and Stage-II steps, HBM counters, smoke
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/final; mkdir -p $O gpurun_out/prof
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -1 $O/bench.json | cut -c1-400
rm -rf gpurun_out/prof/step gpurun_out/prof/stage2
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/step -o step --output-format csv -- python bench.py --quick > $O/step.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/stage2 -o stage2 --output-format csv -- python scripts/bench_material_step.py 4096 128 128 7 bell fused > $O/stage2.log 2>&1
bash scripts/prof_traffic.sh > $O/traffic.log 2>&1
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
rm -rf gpurun_out/prof/*/*.db gpurun_out/prof/*/*kernel_trace.csv
ls -la $O | head; tail -3 $O/bench.err
