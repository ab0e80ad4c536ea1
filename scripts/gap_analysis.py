"""idle gaps of the GPU inside the training step from a rocprofv3 --kernel-trace CSV: per step (delimited by wn_forward_kernel), the sum
of the gaps between consecutive kernels and the largest gaps with the kernels around them.
usage: python scripts/gap_analysis.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
name = lambda r: r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:44]
starts = [i for i, r in enumerate(rows) if 'wn_forward_kernel' in r['Kernel_Name']]
steps = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)]
steps = steps[len(steps) // 2:][:6]                       # steady-state steps
for a, b in steps:
    seg = rows[a:b]
    span = (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e6
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg) / 1e6
    gaps = []
    for x, y in zip(seg[:-1], seg[1:]):
        g = (int(y['Start_Timestamp']) - int(x['End_Timestamp'])) / 1e3
        gaps.append((g, name(x), name(y)))
    big = sorted(gaps, reverse=True)[:8]
    print(f'step: {len(seg)} launches, span {span:.2f} ms, kernel time {busy:.2f} ms, idle {span - busy:.2f} ms; gaps > 10 us: {sum(1 for g in gaps if g[0] > 10)} '
          f'(sum {sum(g[0] for g in gaps if g[0] > 10) / 1e3:.2f} ms)')
    for g, x, y in big:
        print(f'     {g:8.1f} us  after {x:44s} before {y}')
