"""kernels of one drop-in training step (forward({'step': s}) + torch Adam) from a rocprofv3 --kernel-trace CSV: per steady-state step
(delimited by the weight-norm forward kernel) launches, kernel time, idle time, and the kernels grouped by name -- ours (namespace-less
HIP kernels of libnero_hip.so) against torch's (at::native::*, rocprim, hipMemset / copy shaders).
usage: python scripts/r06/dropin_trace_analysis.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def name(r):
    return r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:70]
starts = [i for i, r in enumerate(rows) if 'wn_forward' in r['Kernel_Name']]
steps = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)]
steps = steps[len(steps) // 2:][:8]
agg = collections.OrderedDict()
tot = dict(n=0, span=0.0, busy=0.0)
for a, b in steps:
    seg = rows[a:b]
    span = (int(rows[b]['Start_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e6
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg) / 1e6
    tot['n'] += len(seg); tot['span'] += span; tot['busy'] += busy
    for r in seg:
        k = name(r)
        c = agg.setdefault(k, [0, 0.0])
        c[0] += 1; c[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
n = len(steps)
print(f'{n} steps: {tot["n"] / n:.0f} launches per step, step period {tot["span"] / n:.3f} ms, kernel time {tot["busy"] / n:.3f} ms, idle {(tot["span"] - tot["busy"]) / n:.3f} ms')
is_torch = lambda k: k.startswith('at::') or 'rocprim' in k or 'rocclr' in k or k.startswith('at_')
for label, pred in (('torch / runtime kernels', is_torch), ('libnero_hip kernels', lambda k: not is_torch(k))):
    sel = [(k, v) for k, v in agg.items() if pred(k)]
    print(f'{label}: {sum(v[0] for _, v in sel) / n:.0f} launches, {sum(v[1] for _, v in sel) / n / 1e3:.3f} ms per step')
    for k, v in sorted(sel, key=lambda kv: -kv[1][1])[:40]:
        print(f'   {v[0] / n:6.1f} x  {v[1] / n:8.1f} us  {k}')
