"""instruction mix of one kernel in a saved gfx950 .s file (scripts/isa_stats.py writes them under build/isa/<tag>/): whole kernel and
every backward-branch loop.  usage: python scripts/isa_mix.py build/isa/<tag>/<file>.s <mangled-name-substring> [min-mfma-in-loop]"""
import re, sys
from collections import Counter

s = open(sys.argv[1]).read()
want = sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 1
names = [n for n in re.findall(r'^(_Z\S+):', s, re.M) if want in n]
for nm in names:
    i = s.index('\n' + nm + ':')
    body = s[i:s.index('s_endpgm', i)].split('\n')

    def mix(seg):
        ops = [x.split()[0] for x in seg if x.startswith('\t') and len(x.split()) and not x.strip().startswith(('.', ';'))]
        c = Counter(ops)
        cls = lambda f: sum(v for k, v in c.items() if f(k))
        return c, dict(mfma=cls(lambda k: 'mfma' in k), valu=cls(lambda k: k.startswith('v_') and 'mfma' not in k),
                       salu=cls(lambda k: k.startswith('s_')), ds=cls(lambda k: k.startswith('ds_')),
                       vmem=cls(lambda k: k.startswith(('buffer_', 'global_', 'scratch_', 'flat_'))))
    print(nm)
    print('  whole kernel', mix(body)[1])
    labels = {}
    for k, l in enumerate(body):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = k
    for k, l in enumerate(body):
        t = l.split()
        if len(t) > 1 and t[0] in ('s_cbranch_scc1', 's_cbranch_scc0', 's_cbranch_vccnz', 's_cbranch_vccz', 's_cbranch_execnz', 's_branch') \
                and t[1] in labels and labels[t[1]] < k:
            c, m = mix(body[labels[t[1]]:k])
            if m['mfma'] >= min_mfma:
                print('  loop', t[1], 'lines', labels[t[1]], '-', k, m)
                print('    valu:', sorted(((v, kk) for kk, v in c.items() if kk.startswith('v_') and 'mfma' not in kk), reverse=True)[:18])
