#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S dump:  asm_mix.py gpe.s <mangled-name-prefix> [top]"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
prefix = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and ':' in l)
end = next(j for j in range(start + 1, len(lines)) if lines[j].startswith('.Lfunc_end'))
ops = collections.Counter()
for line in lines[start + 1:end]:
    t = line.strip()
    if not t or t.startswith(('.', ';')) or t.endswith(':'):
        continue
    ops[t.split()[0]] += 1
print(lines[start][:60], 'total instr', sum(ops.values()))
for k, v in ops.most_common(top):
    print('    %-30s %d' % (k, v))
