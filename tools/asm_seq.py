#!/usr/bin/env python3
"""Instruction-class string of a line range of a hipcc -S dump (M mfma, V valu, L lds, G global, X scratch, W waitcnt, S salu, A accvgpr)."""
import sys
lines = open(sys.argv[1]).read().split('\n')
a, b = int(sys.argv[2]), int(sys.argv[3])
def cls(op):
    if op.startswith('v_mfma'): return 'M'
    if op.startswith('ds_'): return 'L'
    if op.startswith(('global_', 'buffer_', 'flat_')): return 'G'
    if op.startswith('scratch_'): return 'X'
    if op.startswith('s_waitcnt'): return 'W'
    if op.startswith('s_'): return 's'
    if op.startswith('v_accvgpr'): return 'A'
    if op.startswith('v_'): return 'v'
    return '?'
out = []
for l in lines[a - 1:b]:
    t = l.strip()
    if not t or t.startswith(('.', ';')) or t.endswith(':'): continue
    out.append(cls(t.split()[0]))
s = ''.join(out)
for i in range(0, len(s), 160): print(s[i:i + 160])
