#!/bin/bash
# round 6 measurement pass (GPU box).  HEAD_SHA=<commit> from the caller.  Profiles run with NERO_STREAMS=1: with the two branch streams
# per-kernel durations overlap (a 5 us kernel "takes" 1 ms waiting for a CU) and PMC passes serialise kernels anyway; the bench line
# runs the default (two streams).  Everything judged is copied under gpurun_out/final/ and from there into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/final; rm -rf $O; mkdir -p $O gpurun_out/prof
SHA=$(sha256sum nero_amd/libnero_hip.so | cut -c1-12)
HEAD=${HEAD_SHA:-unknown}
export NERO_STREAMS=1
bash scripts/prof_traffic.sh > $O/traffic.log 2>&1
python scripts/summarize_traffic.py r06 stage1 $HEAD $SHA > $O/traffic_stage1.txt 2>&1
python scripts/summarize_traffic.py r06 stage2 $HEAD $SHA > $O/traffic_stage2.txt 2>&1
cp profiles/r06_hbm_traffic_per_kernel.csv profiles/r06_stage2_hbm_traffic_per_kernel.csv $O/
# MFMA-busy
M=gpurun_out/prof/mfma; rm -rf $M; mkdir -p $M
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA --kernel-trace -d $M -o p --output-format csv -- python scripts/step_times.py 4096 6 > $M.log 2>&1
python scripts/r06/summarize_mfma.py $HEAD $SHA > $O/mfma.txt 2>&1; cp profiles/r06_mfma_busy_per_kernel.csv $O/
rm -rf $M
# kernel stats: Stage I at 4096 and 512 rays, Stage II; gap analysis at 512 rays (default two streams AND one stream)
for tag in step:"bench.py --quick" r512:"scripts/step_times.py 512 30" stage2:"scripts/bench_material_step.py 4096 128 128 7 bell fused"; do
  n=${tag%%:*}; cmd=${tag#*:}
  rm -rf gpurun_out/prof/$n
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/$n -o $n --output-format csv -- python $cmd > $O/$n.log 2>&1
  find gpurun_out/prof/$n -name "*kernel_stats.csv" -exec cp {} $O/${n}_kernel_stats.csv \;
  if [ $n = r512 ]; then find gpurun_out/prof/$n -name "*kernel_trace.csv" -exec python scripts/gap_analysis.py {} \; > $O/gap_analysis_r512_one_stream.txt 2>&1; fi
  if [ $n = step ]; then find gpurun_out/prof/$n -name "*kernel_trace.csv" -exec python scripts/gap_analysis.py {} \; > $O/gap_analysis.txt 2>&1; fi
done
unset NERO_STREAMS
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -1 $O/bench.json | cut -c1-300
timeout 300 python bench.py --stage 2 > $O/bench_stage2.json 2>> $O/bench.err
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
rm -rf gpurun_out/prof/*/*.db gpurun_out/prof/*/*kernel_trace.csv gpurun_out/prof/*/*/*.db gpurun_out/prof/*/*/*kernel_trace.csv gpurun_out/prof/fetch* gpurun_out/prof/write* gpurun_out/prof/step gpurun_out/prof/r512 gpurun_out/prof/stage2
# round-6 extras
bash scripts/probe/run_rowowner_probe.sh > /dev/null 2>&1; cp gpurun_out/rowowner_probe.txt $O/rowowner_probe.txt
python scripts/r06/bench_rowowner.py 2>&1 | grep -v Warn | tail -6 > $O/three_organisations.txt
python scripts/r06/dropin_ab.py 2>&1 | grep -v Warn | tail -24 > $O/dropin_ab.txt
python scripts/r06/dropin_profile.py 512 2>&1 | grep "rays:" >> $O/dropin_ab.txt
python scripts/r06/dropin_ab2.py 512 2>&1 | grep -v Warn | tail -2 >> $O/dropin_ab.txt
TAG=_final bash scripts/r06/dropin_trace.sh > /dev/null 2>&1; cp gpurun_out/r06/dropin_trace_final.txt $O/dropin_trace.txt
{ for lib in "" 2acc; do echo "=== lib ${lib:-in-tree (one accumulator set, round 5/6 default)}"; if [ -n "$lib" ]; then export NERO_HIP_LIB=$PWD/build/variants/lib_$lib.so; else unset NERO_HIP_LIB; fi; python scripts/r06/smoke_diff.py 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -14; done; unset NERO_HIP_LIB; } > $O/smoke_diff.txt 2>&1
cp gpurun_out/parity_at_size.json $O/ 2>/dev/null
ls -la $O | head -40; tail -3 $O/bench.err
