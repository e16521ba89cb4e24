#!/usr/bin/env python3
"""Count the instructions of an ISA listing (hipcc -S) between two line numbers, by class.
usage: isa_count.py file.s first last"""
import re, sys, collections
f, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
c = collections.Counter()
ops = collections.Counter()
for i, l in enumerate(open(f), 1):
    if i < a or i > b: continue
    l = l.strip()
    if not l or l.startswith((';', '.', '//')) or l.endswith(':'): continue
    op = l.split()[0]
    if op.startswith('v_'): c['valu'] += 1
    elif op.startswith('ds_'): c['lds'] += 1
    elif op.startswith('s_waitcnt') or op.startswith('s_nop'): c['wait/nop'] += 1
    elif op.startswith('s_'): c['salu'] += 1
    elif op.startswith(('global_', 'scratch_', 'buffer_', 'flat_')): c['vmem'] += 1
    else: c['other'] += 1
    ops[re.sub(r'_e32|_e64|_sdwa|_dpp', '', op)] += 1
print(dict(c))
print(ops.most_common(40))
