"""gpurun_out/prof/{fetch,write}[_stage2]/*counter_collection.csv (scripts/prof_traffic.sh) -> profiles/rNN_[stage2_]hbm_traffic_per_kernel.csv:
average HBM KB per launch and GB per training step of every kernel, start-up passes excluded.
usage: python scripts/summarize_traffic.py <round tag> <stage1|stage2> <head commit> <library sha12>"""
import collections, csv, glob, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
which = sys.argv[2] if len(sys.argv) > 2 else 'stage1'
head = sys.argv[3] if len(sys.argv) > 3 else 'unknown'
sha = sys.argv[4] if len(sys.argv) > 4 else 'unknown'
suffix = '' if which == 'stage1' else '_stage2'
marker = 'composite_bwd' if which == 'stage1' else 'mc_combine_bwd'       # one launch per training step
first = 'wn_forward_kernel'
def base(k):
    return k.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].split('<')[0].strip()
res = {}
passes = 0
for kind, cn in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
    f = glob.glob(os.path.join(ROOT, f'gpurun_out/prof/{kind}{suffix}/**/*counter_collection.csv'), recursive=True)
    rows = [r for r in csv.DictReader(open(f[0])) if r['Counter_Name'] == cn]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    wn = [i for i, r in enumerate(rows) if first in r['Kernel_Name']]
    rows = rows[wn[2] if len(wn) > 2 else 0:]                 # skip construction / the first (start-up) steps
    passes = sum(1 for r in rows if marker in r['Kernel_Name'])
    a, n = collections.defaultdict(float), collections.Counter()
    for r in rows:
        a[base(r['Kernel_Name'])] += float(r['Counter_Value']); n[base(r['Kernel_Name'])] += 1
    res[kind] = (a, n)
ks = sorted(res['fetch'][0], key=lambda k: -(2 * res['fetch'][0][k] + res['write'][0].get(k, 0)))[:24]
out = os.path.join(ROOT, 'profiles', f'{tag}{suffix}_hbm_traffic_per_kernel.csv')
tot = 0.0
with open(out, 'w') as fo:
    fo.write(f'# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, scripts/prof_traffic.sh) over the {which} training step: {passes} steps on the default engines, start-up steps excluded.\n')
    fo.write(f'# taken on: head {head}, libnero_hip.so sha256[:12] = {sha}\n')
    fo.write('# FETCH_SIZE on gfx950 reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM section): fetch_kb = 2 x raw.  KB per launch (average) and GB per training step.\n')
    fo.write('kernel,launches_per_step,fetch_kb_raw,fetch_kb,write_kb,gb_per_step\n')
    for k in ks:
        nf = res['fetch'][1][k]; fr = res['fetch'][0][k] / nf
        nw = res['write'][1].get(k, 0); wr = res['write'][0].get(k, 0) / nw if nw else 0.0
        gb = (2 * fr + wr) * nf / max(passes, 1) / 1e6 * 1.024
        tot += gb
        fo.write(f'{k},{nf / max(passes, 1):.0f},{fr:.0f},{2 * fr:.0f},{wr:.0f},{gb:.2f}\n')
    fo.write(f'# sum of the listed kernels: {tot:.1f} GB per step\n')
print(open(out).read())
