#!/bin/bash
# PMC passes over ONE variant of scripts/trace_step_rays.py at a time (STEP_RAYS_ONLY), both traversal kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/prof
for VAR in "step rays" "step rays, inward directions mirrored outward" "synthetic rays"; do
  export STEP_RAYS_ONLY="$VAR"
  echo "=== $VAR"
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
             "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    i=$((i+1)); O=gpurun_out/prof/ts$i; rm -rf $O; mkdir -p $O
    timeout 200 rocprofv3 --pmc $grp --kernel-trace -d $O -o p --output-format csv -- python scripts/trace_step_rays.py > $O.log 2>&1
    python - "$O" <<'P'
import collections, csv, glob, sys
fs = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
if not fs:
    print('pass failed'); sys.exit(0)
rows = list(csv.DictReader(open(fs[0])))
rows = [r for r in rows if 'trace_' in r['Kernel_Name']]
# the step itself launches the tracer 3 times before the timed loops: drop each kernel's first dispatches of the default mode by keeping the LAST 8 per kernel
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = 'overlap' if 'overlap' in r['Kernel_Name'] else 'one-at-a-time'
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in acc:
    print('  ', k, ' '.join(f"{c}={sum(v[-8:]) / len(v[-8:]):.4g}" for c, v in acc[k].items()))
P
    rm -rf $O
  done
done
