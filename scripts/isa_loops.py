"""Per-loop instruction mix of a gfx950 .s file (hipcc -save-temps): python scripts/isa_loops.py file.s [name-substring]"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\n\.Lfunc_end', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat not in name: continue
    lines = [l.strip() for l in body.split('\n')]
    lines = [l for l in lines if l and not l.startswith(';') and not (l.startswith('.') and not l.startswith('.LBB'))]
    print(name, len(lines), "lines; scratch ops total", sum(l.startswith('scratch') for l in lines))
    labels = {}
    for idx, l in enumerate(lines):
        mm = re.match(r'(\.LBB\d+_\d+):', l)
        if mm: labels[mm.group(1)] = idx
    for idx, l in enumerate(lines):
        mm = re.match(r's_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)', l)
        if not mm: continue
        t = mm.group(1) or mm.group(2)
        if t in labels and labels[t] < idx:
            c = Counter()
            for b in lines[labels[t]:idx + 1]:
                op = b.split()[0]
                if op.startswith('scratch'): c['scratch'] += 1
                elif op.startswith('ds_'): c['ds'] += 1
                elif op.startswith('global') or op.startswith('buffer'): c['vmem'] += 1
                elif 'dpp' in b: c['dpp'] += 1
                elif op.startswith('v_readlane') or op.startswith('v_writelane'): c['lane'] += 1
                elif op.startswith('v_mov') or op.startswith('v_accvgpr'): c['v_mov'] += 1
                elif op.startswith('v_'): c['valu'] += 1
                elif op.startswith('s_waitcnt'): c['wait'] += 1
                elif op.startswith('s_'): c['salu'] += 1
                elif op.startswith('.LBB'): c['label'] += 1
                else: c[op] += 1
            print('  loop %s lines %d..%d (%d)' % (t, labels[t], idx, idx - labels[t] + 1), dict(c))
