"""Instruction statistics of the loops of every kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only):
per backward-branch region -- instructions, branches, s_waitcnt, s_nop, transcendentals.  python tools/loop_stats.py file.s"""
import re, sys
s = open(sys.argv[1]).read()
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\n\s+s_endpgm', s, re.S | re.M):
    lines = [l.strip() for l in m.group(2).split('\n')]
    labels = {l.split(':')[0]: i for i, l in enumerate(lines) if re.match(r'^\.LBB\d+_\d+:', l)}
    loops = []
    for i, l in enumerate(lines):
        mm = re.match(r's_c?branch\w* (\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            loops.append((labels[mm.group(1)], i))
    cnt = lambda a, b, f: sum(1 for l in lines[a:b] if f(l))
    isin = lambda l: l and not l.startswith(('.', ';'))
    print(m.group(1))
    for a, b in loops:
        print('   lines %d-%d: instr %d  branches %d  waitcnt %d  nop %d  transcendental %d  dpp %d' % (
            a, b, cnt(a, b, isin), cnt(a, b, lambda l: l.startswith(('s_cbranch', 's_branch'))), cnt(a, b, lambda l: 's_waitcnt' in l),
            cnt(a, b, lambda l: l.startswith('s_nop')), cnt(a, b, lambda l: l.startswith(('v_sin', 'v_cos', 'v_rcp', 'v_rsq', 'v_sqrt', 'v_exp'))),
            cnt(a, b, lambda l: 'quad_perm' in l or 'row_' in l)))
