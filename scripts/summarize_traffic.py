"""gpurun_out/prof/{fetch,write}/*counter_collection.csv (scripts/prof_traffic.sh) -> profiles/rNN_hbm_traffic_per_kernel.csv:
average HBM KB per launch and GB per training step of every kernel, the 64-ray start-up pass excluded.
usage: python scripts/summarize_traffic.py [round tag, default r02]"""
import collections, csv, glob, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
def base(k):
    return k.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].split('<')[0].strip()
res = {}
passes = 0
for kind, cn in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
    f = glob.glob(os.path.join(ROOT, f'gpurun_out/prof/{kind}/**/*counter_collection.csv'), recursive=True)
    rows = [r for r in csv.DictReader(open(f[0])) if r['Counter_Name'] == cn]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    wn = [i for i, r in enumerate(rows) if 'wn_forward_kernel' in r['Kernel_Name']]
    rows = rows[wn[1]:]                                   # everything from the second pass on (the first is the 64-ray start-up)
    passes = sum(1 for r in rows if 'composite_bwd_kernel' in r['Kernel_Name'])
    a, n = collections.defaultdict(float), collections.Counter()
    for r in rows:
        a[base(r['Kernel_Name'])] += float(r['Counter_Value']); n[base(r['Kernel_Name'])] += 1
    res[kind] = (a, n)
ks = sorted(res['fetch'][0], key=lambda k: -(2 * res['fetch'][0][k] + res['write'][0].get(k, 0)))[:20]
out = os.path.join(ROOT, 'profiles', f'{tag}_hbm_traffic_per_kernel.csv')
tot = 0.0
with open(out, 'w') as fo:
    fo.write(f'# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, scripts/prof_traffic.sh) over `python bench.py --steps 2 --warmup 2 --quick`: {passes} full forward+backward passes on the default engines; the 64-ray start-up pass is excluded.\n')
    fo.write('# FETCH_SIZE on gfx950 reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM section): fetch_kb = 2 x raw.  KB per launch (average) and GB per training step.\n')
    fo.write('kernel,launches_per_step,fetch_kb_raw,fetch_kb,write_kb,gb_per_step\n')
    for k in ks:
        nf = res['fetch'][1][k]; fr = res['fetch'][0][k] / nf
        nw = res['write'][1].get(k, 0); wr = res['write'][0].get(k, 0) / nw if nw else 0.0
        gb = (2 * fr + wr) * nf / passes / 1e6 * 1.024
        tot += gb
        fo.write(f'{k},{nf / passes:.0f},{fr:.0f},{2 * fr:.0f},{wr:.0f},{gb:.2f}\n')
    fo.write(f'# sum of the listed kernels: {tot:.1f} GB per step\n')
print(open(out).read())
