#!/bin/bash
# effective shader clock and MFMA-busy of the three chain organisations (512-thread lock step, two workgroups per CU, row owner) on the same
# chains: rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA + --kernel-trace over scripts/r06/bench_rowowner.py
#   bash scripts/r06/clock_by_organisation.sh  ->  gpurun_out/r06/clock_by_organisation.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/prof/clk; rm -rf $O; mkdir -p $O gpurun_out/r06
timeout 500 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES --kernel-trace -d $O -o p --output-format csv -- python scripts/r06/bench_rowowner.py > $O.log 2>&1
tail -5 $O.log
python - <<'P'
import collections, csv, glob
cc = glob.glob('gpurun_out/prof/clk/**/*counter_collection.csv', recursive=True)[0]
kt = glob.glob('gpurun_out/prof/clk/**/*kernel_trace.csv', recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[int(r['Dispatch_Id'])] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); t = collections.defaultdict(float); seen = set()
for r in csv.DictReader(open(cc)):
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].replace(',', ';')
    if 'fwd_' not in k: continue
    key = (k, int(r['Grid_Size']) if 'Grid_Size' in r else 0)
    acc[key][r['Counter_Name']] += float(r['Counter_Value'])
    d = int(r['Dispatch_Id'])
    if (key, d) not in seen:
        seen.add((key, d)); n[key] += 1; t[key] += dur.get(d, 0.0)
with open('gpurun_out/r06/clock_by_organisation.csv', 'w') as fo:
    fo.write('# rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES --kernel-trace over scripts/r06/bench_rowowner.py (profiled passes clock ~3 % lower than unprofiled ones)\n')
    fo.write('# effective_GHz = (GRBM_GUI_ACTIVE / 8 XCDs) / kernel duration;  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles);  f16 MFMA rate = insts x 32768 FLOP / duration\n')
    fo.write('kernel,grid,dispatches,ms_per_dispatch,cycles_per_dispatch,effective_GHz,mfma_insts_per_dispatch,mfma_busy_frac,PFLOPs_f16\n')
    for key in sorted(acc, key=lambda k: (k[0], k[1])):
        d, m = acc[key], max(n[key], 1)
        cyc = d['GRBM_GUI_ACTIVE'] / 8 / m; ms = t[key] / m * 1e3; mi = d['SQ_INSTS_MFMA'] / m
        fo.write(f"{key[0]},{key[1]},{n[key]},{ms:.4f},{cyc:.0f},{cyc / (ms * 1e6):.3f},{mi:.0f},{d['SQ_VALU_MFMA_BUSY_CYCLES'] / m / (1024 * cyc):.4f},{mi * 32768 / (ms * 1e-3) / 1e15:.3f}\n")
print(open('gpurun_out/r06/clock_by_organisation.csv').read())
P
