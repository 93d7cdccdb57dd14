#!/usr/bin/env python3
"""Per-loop instruction mix of a gfx950 ISA listing (hipcc -S --cuda-device-only): for every loop
the compiler annotated, instruction count and the share of scratch (spill) traffic, AGPR copies,
SGPR-spill lane moves, fp64 arithmetic, memory operations.  Blocks are attributed to the innermost
loop named in their annotation.  Usage: tools/isa_loops.py kernel.s [min_instructions]"""
import re
import sys
from collections import OrderedDict

path = sys.argv[1]
min_n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
loops = OrderedDict()
cur = None
pending_label = None
for line in open(path):
    m = re.match(r'^(\.LBB\d+_\d+):\s*;?(.*)', line)
    if m:
        pending_label = m.group(1)
        ann = m.group(2)
        cur = None
        h = re.search(r'in Loop: Header=(BB\d+_\d+) Depth=(\d+)', ann)
        if h:
            cur = '.L' + h.group(1)
        elif 'Loop Header' in ann:
            cur = pending_label
        continue
    if pending_label and re.match(r'^\s*;\s*(=>)?\s*(This )?(Inner )?Loop Header', line):
        cur = pending_label
        continue
    if cur is None or not line.startswith('\t') or line.startswith('\t.') or line.startswith('\t;'):
        continue
    op = line.split()[0]
    s = loops.setdefault(cur, dict(n=0, scratch=0, agpr=0, lane=0, f64=0, vmem=0, lds=0, salu=0, wait=0))
    s['n'] += 1
    if op.startswith('scratch_'): s['scratch'] += 1
    elif op.startswith('v_accvgpr'): s['agpr'] += 1
    elif op.startswith(('v_readlane', 'v_writelane')): s['lane'] += 1
    elif 'f64' in op: s['f64'] += 1
    elif op.startswith(('global_', 'buffer_', 'flat_')): s['vmem'] += 1
    elif op.startswith('ds_'): s['lds'] += 1
    elif op.startswith(('s_waitcnt', 's_nop')): s['wait'] += 1
    elif op.startswith('s_'): s['salu'] += 1
for k, s in loops.items():
    if s['n'] >= min_n:
        print(k, ' '.join(f"{a}={b}" for a, b in s.items()))
