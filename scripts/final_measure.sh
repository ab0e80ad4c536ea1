#!/bin/bash
# the round's measurement pass (run on the GPU box): bench line, kernel stats, HBM counters, Stage-II stats, phase profile, PMC, smoke
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/final; mkdir -p $O gpurun_out/prof
rm -rf gpurun_out/prof/step gpurun_out/prof/fetch gpurun_out/prof/write gpurun_out/prof/stage2
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-600
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/step -o step --output-format csv -- python bench.py --quick > $O/step.log 2>&1
bash scripts/prof_traffic.sh > $O/traffic.log 2>&1
bash scripts/prof_stage2.sh > $O/stage2.log 2>&1
NERO_HIP_LIB=$R/build/variants/libv_base.so timeout 200 python scripts/phase_timing.py 524288 f16x3 2>&1 | grep -v amdgpu > $O/phase_f16x3.txt
NERO_HIP_LIB=$R/build/variants/libv_base.so timeout 200 python scripts/phase_timing.py 524288 f16x3p 2>&1 | grep -v amdgpu > $O/phase_f16x3p.txt
OUT=$R/gpurun_out/pmc bash scripts/prof_chain_pmc.sh > $O/pmc.log 2>&1
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
rm -rf gpurun_out/prof/*/*.db gpurun_out/pmc/pass*/*.db gpurun_out/prof/fetch/*kernel_trace.csv gpurun_out/prof/write/*kernel_trace.csv
ls -la $O gpurun_out/prof/step | head -30
