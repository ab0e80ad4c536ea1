#!/bin/bash
# HBM traffic counters of the step's kernels (separate --pmc passes, as the microarch guide prescribes); summary -> gpurun_out/prof/hbm_traffic.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/prof/fetch -o fetch --output-format csv -- python bench.py --steps 2 --warmup 2 --quick > gpurun_out/prof/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/prof/write -o write --output-format csv -- python bench.py --steps 2 --warmup 2 --quick > gpurun_out/prof/write.log 2>&1
python - <<'PY'
import csv, glob, collections
tot = {}
for tag, cn in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
    f = glob.glob(f'gpurun_out/prof/{tag}/**/*counter_collection.csv', recursive=True)
    a = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if r['Counter_Name'] != cn: continue
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].split('<')[0]
        a[k] += float(r['Counter_Value']); n[k] += 1
    tot[tag] = (a, n)
ks = sorted(tot['fetch'][0], key=lambda k: -(tot['fetch'][0][k] + tot['write'][0].get(k, 0)))[:14]
with open('gpurun_out/prof/hbm_traffic.csv', 'w') as fo:
    fo.write('# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py --steps 2 --warmup 2`; KB per launch (average).\n')
    fo.write('# FETCH_SIZE on gfx950 reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM section): corrected = 2 x raw.  Includes the f32-engine comparison steps of bench.py.\n')
    fo.write('kernel,launches,fetch_kb_raw,fetch_kb_corrected,write_kb\n')
    for k in ks:
        nf = tot['fetch'][1][k]; fr = tot['fetch'][0][k] / nf
        nw = tot['write'][1].get(k, 0); wr = tot['write'][0].get(k, 0) / nw if nw else 0
        fo.write(f'{k},{nf},{fr:.0f},{2*fr:.0f},{wr:.0f}\n')
print(open('gpurun_out/prof/hbm_traffic.csv').read())
PY
