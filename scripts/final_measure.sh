#!/bin/bash
# the round's measurement pass (run on the GPU box; HEAD_SHA=<commit> from the caller): HBM counters first (their summary is what
# bench.py's roofline block reads), kernel stats of the Stage-I and Stage-II steps, the bench line, tracer / head-gradient
# micro-benchmarks, smoke.  Everything judged is copied under gpurun_out/final/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/final; mkdir -p $O gpurun_out/prof
SHA=$(sha256sum nero_amd/libnero_hip.so | cut -c1-12)
bash scripts/prof_traffic.sh > $O/traffic.log 2>&1
python scripts/summarize_traffic.py r03 stage1 ${HEAD_SHA:-unknown} $SHA > $O/traffic_stage1.txt 2>&1
python scripts/summarize_traffic.py r03 stage2 ${HEAD_SHA:-unknown} $SHA > $O/traffic_stage2.txt 2>&1
cp profiles/r03_hbm_traffic_per_kernel.csv profiles/r03_stage2_hbm_traffic_per_kernel.csv $O/
rm -rf gpurun_out/prof/step gpurun_out/prof/stage2
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/step -o step --output-format csv -- python bench.py --quick > $O/step.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/stage2 -o stage2 --output-format csv -- python scripts/bench_material_step.py 4096 128 128 7 bell fused > $O/stage2.log 2>&1
find gpurun_out/prof/step -name "*kernel_stats.csv" -exec cp {} $O/step_kernel_stats.csv \;
find gpurun_out/prof/stage2 -name "*kernel_stats.csv" -exec cp {} $O/stage2_kernel_stats.csv \;
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -1 $O/bench.json | cut -c1-400
timeout 200 python scripts/trace_bench.py $O/trace_bench.json > $O/trace_bench.log 2>&1
timeout 100 python scripts/bench_head_dw.py > $O/head_dw.txt 2>&1; cat $O/head_dw.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
rm -rf gpurun_out/prof/*/*.db gpurun_out/prof/*/*kernel_trace.csv gpurun_out/prof/*/*/*.db gpurun_out/prof/*/*/*kernel_trace.csv gpurun_out/prof/fetch* gpurun_out/prof/write*
ls -la $O | head -20; tail -3 $O/bench.err
