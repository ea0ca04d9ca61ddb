#!/usr/bin/env python3
"""Compressed instruction-class sequence of one kernel in a hipcc --save-temps .s dump, starting at its first MFMA:
M mfma, r/w ds_read/ds_write, G/S global load/store, F flat, B barrier, |..| s_waitcnt, v VALU, s SALU.
usage: asm_seq2.py file.s <mangled-name-prefix> [chars]"""
import sys
lines = open(sys.argv[1]).read().split('\n')
pre = sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
start = next(i for i, l in enumerate(lines) if l.startswith(pre) and ':' in l)
end = next(j for j in range(start + 1, len(lines)) if lines[j].startswith('.Lfunc_end'))
seq = []
for l in lines[start:end]:
    t = l.strip()
    if not t or t.startswith(('.', ';')) or t.endswith(':'):
        continue
    op = t.split()[0]
    if op.startswith('v_mfma'): k = 'M'
    elif op.startswith('ds_read'): k = 'r'
    elif op.startswith('ds_write') or op.startswith('ds_add'): k = 'w'
    elif op.startswith('global_load'): k = 'G'
    elif op.startswith('global_store'): k = 'S'
    elif op.startswith('flat_'): k = 'F'
    elif op.startswith('s_waitcnt'): k = '|' + t.split(None, 1)[1].replace(' ', '') + '|'
    elif op.startswith('s_barrier'): k = 'B'
    elif op.startswith('v_'): k = 'v'
    elif op.startswith('s_'): k = 's'
    else: k = '?'
    seq.append(k)
out, prev, cnt = [], None, 0
for k in seq:
    if k == prev: cnt += 1
    else:
        if prev is not None: out.append(prev + (str(cnt) if cnt > 1 else ''))
        prev, cnt = k, 1
out.append(prev + (str(cnt) if cnt > 1 else ''))
s = ' '.join(out)
i = s.find('M')
print(s[max(0, i - 300):i + n])
