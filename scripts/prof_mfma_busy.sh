#!/bin/bash
# MFMA-busy of every kernel of the Stage-I training step (one rocprofv3 --pmc pass, kernel trace only), run on the GPU box:
#   bash scripts/prof_mfma_busy.sh  ->  gpurun_out/mfma_busy.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/prof/mfma; rm -rf $O; mkdir -p $O
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA --kernel-trace -d $O -o p --output-format csv -- python scripts/step_times.py 4096 6 > $O.log 2>&1
tail -1 $O.log
python - <<'P'
import collections, csv, glob
f = glob.glob('gpurun_out/prof/mfma/**/*counter_collection.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Dispatch_Id']))
first = [i for i, r in enumerate(rows) if 'wn_forward_kernel' in r['Kernel_Name']]
start_id = int(rows[first[len(first) // 2]]['Dispatch_Id'])          # steady-state steps only
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in rows:
    if int(r['Dispatch_Id']) < start_id:
        continue
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].replace(',', ';')
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
        n[k] += 1
with open('gpurun_out/mfma_busy.csv', 'w') as fo:
    fo.write('# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA over the Stage-I training step (4096 rays), steady-state steps.\n')
    fo.write('# GRBM_GUI_ACTIVE sums over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs: mfma_busy_frac = MFMA_BUSY / (1024 x GUI_ACTIVE / 8).\n')
    fo.write('kernel,dispatches,gui_active_cycles_per_dispatch,sq_busy_cycles_per_dispatch,mfma_busy_cycles_per_dispatch,mfma_insts_per_dispatch,mfma_busy_frac\n')
    for k in sorted(acc, key=lambda k: -acc[k]['GRBM_GUI_ACTIVE'])[:14]:
        d = acc[k]
        m = max(n[k], 1)
        frac = d['SQ_VALU_MFMA_BUSY_CYCLES'] / (128 * d['GRBM_GUI_ACTIVE']) if d['GRBM_GUI_ACTIVE'] else 0.0
        fo.write(f"{k},{n[k]},{d['GRBM_GUI_ACTIVE']/m:.0f},{d['SQ_BUSY_CYCLES']/m:.0f},{d['SQ_VALU_MFMA_BUSY_CYCLES']/m:.0f},{d['SQ_INSTS_MFMA']/m:.0f},{frac:.4f}\n")
print(open('gpurun_out/mfma_busy.csv').read())
P
rm -rf $O
