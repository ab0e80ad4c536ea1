#!/bin/bash
# PMC passes over the tracer microbenchmark (scripts/trace_bench.py, the 4096 x 256 secondary-ray case, both traversals), on the GPU box:
#   bash scripts/prof_trace.sh  ->  gpurun_out/trace_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/prof
export TRACE_CASES=${TRACE_CASES:-0} TRACE_ITERS=4
rm -f gpurun_out/trace_pmc.txt
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD" \
           "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TA_TCP_STATE_READ_sum"; do
    i=$((i+1)); O=gpurun_out/prof/tr$i; rm -rf $O; mkdir -p $O
    timeout 200 rocprofv3 --pmc $grp --kernel-trace -d $O -o p --output-format csv -- python ${TRACE_SCRIPT:-scripts/trace_bench.py} > $O.log 2>&1
    python - "$O" "$grp" >> gpurun_out/trace_pmc.txt <<'P'
import collections, csv, glob, sys
fs = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
if not fs:
    print('pass failed:', sys.argv[2]); print(open(sys.argv[1] + '.log').read()[-600:]); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(fs[0])):
    k = r['Kernel_Name']
    if 'trace_' not in k: continue
    k = 'trace_overlap_kernel' if 'overlap' in k else 'trace_kernel'
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    n[(k, r['Counter_Name'])] += 1
for k in acc:
    print(k, ' '.join(f"{c}={v / n[(k, c)]:.4g}" for c, v in acc[k].items()), f"(per dispatch, {max(n[(k, c)] for c in acc[k])} dispatches)")
P
    rm -rf $O
done
cat gpurun_out/trace_pmc.txt
