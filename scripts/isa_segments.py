"""Where does a kernel spill?  Instruction mix of a gfx950 .s file (hipcc -S --cuda-device-only) per barrier-separated
segment of one function: python scripts/isa_segments.py file.s <function-name-substring>"""
import sys
lines = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and pat in l.split(':')[0] and l.rstrip().split(';')[0].strip().endswith(':'))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = [l.strip() for l in lines[start + 1:end]]
body = [l for l in body if l and not l.startswith(';')]
seg, cnt, order = 0, {}, []
for l in body:
    op = l.split()[0]
    if op == 's_barrier':
        seg += 1
        continue
    if l.startswith('.LBB') or op.endswith(':'):
        continue
    d = cnt.setdefault(seg, dict(n=0, sc_ld=0, sc_st=0, ds=0, vmem=0, dpp=0, lane=0, valu=0, salu=0, call=0))
    d['n'] += 1
    if op.startswith('scratch_load'): d['sc_ld'] += 1
    elif op.startswith('scratch_store'): d['sc_st'] += 1
    elif op.startswith('ds_'): d['ds'] += 1
    elif op.startswith('global') or op.startswith('buffer') or op.startswith('flat'): d['vmem'] += 1
    elif 'dpp' in l: d['dpp'] += 1
    elif op.startswith('v_readlane') or op.startswith('v_writelane'): d['lane'] += 1
    elif op.startswith('s_swappc') or op.startswith('s_setpc'): d['call'] += 1
    elif op.startswith('v_'): d['valu'] += 1
    elif op.startswith('s_'): d['salu'] += 1
print(len(body), "instructions,", seg, "barriers")
for k, v in cnt.items():
    print("segment %2d" % k, v)
