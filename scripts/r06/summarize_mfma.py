"""gpurun_out/prof/mfma/**/*counter_collection.csv (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA over
scripts/step_times.py) -> profiles/r06_mfma_busy_per_kernel.csv, stamped with the commit and library hash it was taken on.
usage: python scripts/r06/summarize_mfma.py <head commit> <library sha12>"""
import collections, csv, glob, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
head, sha = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ('unknown', 'unknown')
f = glob.glob(os.path.join(ROOT, 'gpurun_out/prof/mfma/**/*counter_collection.csv'), recursive=True)[0]
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
out = os.path.join(ROOT, 'profiles', 'r06_mfma_busy_per_kernel.csv')
with open(out, 'w') as fo:
    fo.write('# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA --kernel-trace over the Stage-I training step (4096 rays, scripts/step_times.py, one stream), steady-state steps, per dispatch.\n')
    fo.write(f'# taken on: head {head}, libnero_hip.so sha256[:12] = {sha}\n')
    fo.write('# GRBM_GUI_ACTIVE is summed over the 8 XCDs (cycles_per_dispatch = that / 8), SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs:\n')
    fo.write('# mfma_busy_frac = MFMA_BUSY / (1024 x cycles_per_dispatch) = the share of all SIMD-cycles of the launch in which the matrix pipe is occupied.\n')
    fo.write('kernel,dispatches,cycles_per_dispatch,mfma_busy_cycles_per_dispatch,mfma_insts_per_dispatch,busy_cycles_per_mfma,mfma_busy_frac\n')
    for k in sorted(acc, key=lambda k: -acc[k]['GRBM_GUI_ACTIVE'])[:16]:
        d, m = acc[k], max(n[k], 1)
        cyc = d['GRBM_GUI_ACTIVE'] / 8 / m
        mb, mi = d['SQ_VALU_MFMA_BUSY_CYCLES'] / m, d['SQ_INSTS_MFMA'] / m
        fo.write(f"{k},{n[k]},{cyc:.0f},{mb:.0f},{mi:.0f},{(mb / mi if mi else 0):.1f},{(mb / (1024 * cyc) if cyc else 0):.4f}\n")
print(open(out).read())
