#!/usr/bin/env python3
"""Instruction mix per basic-block group of ONE kernel of a gfx950 ISA listing.
Usage: tools/isa_kernel_loops.py listing.s <substring of the mangled kernel name> [min_instructions]"""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 100
start = [i for i, l in enumerate(lines) if l.startswith('_ZN') and key in l and l.rstrip().split(':')[0].endswith(l.split(':')[0])][0]
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
cur = 'entry'
loops = {}
KEYS = (('dpp', lambda o: 'dpp' in o), ('ds', lambda o: o.startswith('ds_')), ('f64', lambda o: 'f64' in o),
        ('wait', lambda o: o.startswith('s_waitcnt')), ('nop', lambda o: o.startswith('s_nop')),
        ('acc', lambda o: 'accvgpr' in o), ('lane', lambda o: o.startswith(('v_readlane', 'v_writelane'))),
        ('mov', lambda o: o.startswith('v_mov')), ('vmem', lambda o: o.startswith(('global_', 'buffer_', 'flat_', 'scratch_'))),
        ('branch', lambda o: o.startswith(('s_cbranch', 's_branch'))))
for l in lines[start:end]:
    m = re.match(r'^(\.LBB\d+_\d+):\s*;?(.*)', l)
    if m:
        h = re.search(r'in Loop: Header=(BB\d+_\d+) Depth=(\d+)', m.group(2))
        cur = ('.L' + h.group(1) + ' depth ' + h.group(2)) if h else m.group(1)
        continue
    m = re.match(r'^; %bb\.\d+:\s*;?(.*)', l)
    if m:
        h = re.search(r'in Loop: Header=(BB\d+_\d+) Depth=(\d+)', m.group(1))
        if h:
            cur = '.L' + h.group(1) + ' depth ' + h.group(2)
        continue
    if not l.startswith('\t') or l.startswith('\t.') or l.startswith('\t;') or not l.strip():
        continue
    op = l.split()[0]
    d = loops.setdefault(cur, {'n': 0})
    d['n'] += 1
    for k, pred in KEYS:
        if pred(op):
            d[k] = d.get(k, 0) + 1
for k, v in loops.items():
    if v['n'] >= min_n:
        print(k, v)
