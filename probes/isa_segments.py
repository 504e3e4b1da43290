#!/usr/bin/env python3
"""Instruction counts of a kernel between its s_barrier instructions, from the assembly probes/kernel_regs.sh leaves in
/tmp/sa_hip_last.s (no GPU needed): probes/isa_segments.py <substring of the mangled kernel name> [min instructions]"""
import sys
from collections import Counter

def cls(i):
    op = i.split()[0]
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_barrier'): return 'barrier'
    if op.startswith('s_load') or op.startswith('s_buffer_load'): return 'smem'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return 'branch'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    return 'other'

pat = sys.argv[1]
minlen = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lines = open('/tmp/sa_hip_last.s').read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and pat in l.split(':')[0] and l.rstrip().split(';')[0].rstrip().endswith(':'))
seg, cur, label = [], [], 'entry'
for l in lines[start + 1:]:
    t = l.split(';')[0].strip()
    if not t or t.startswith('.') and not t.endswith(':'):
        continue
    if t.endswith(':'):
        if cur: seg.append((label, cur))
        cur, label = [], t[:-1]
        continue
    cur.append(t)
    if t.startswith('s_endpgm'):
        break
seg.append((label, cur))
tot = Counter()
for label, sg in seg:
    c = Counter(cls(i) for i in sg)
    tot.update(c)
    if len(sg) >= minlen:
        print('%-12s %4d  %s' % (label, len(sg), ' '.join('%s=%d' % kv for kv in sorted(c.items()))))
print('total', sum(tot.values()), dict(tot))
