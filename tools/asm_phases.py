#!/usr/bin/env python3
"""Where the matrix pipe idles: walk one kernel of a hipcc -S dump in program order and print stretches of
non-MFMA instructions (>= min_len) between MFMA groups, with their instruction-class mix.
usage: asm_phases.py file.s <mangled-name-substring> [min_len]"""
import collections, re, sys
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
min_len = int(sys.argv[3]) if len(sys.argv) > 3 else 24
start = next(i for i, l in enumerate(lines) if key in l and l.startswith("_Z") and ":" in l)
end = next(j for j in range(start + 1, len(lines)) if lines[j].startswith('.Lfunc_end'))
def cls(op):
    if op.startswith('v_mfma'): return 'M'
    if op.startswith('ds_'): return 'L'
    if op.startswith(('global_', 'buffer_', 'flat_')): return 'G'
    if op.startswith('scratch_'): return 'X'
    if op.startswith('s_waitcnt'): return 'W'
    if op.startswith('s_'): return 'S'
    if op.startswith('v_accvgpr'): return 'A'
    if op.startswith('v_'): return 'V'
    return '?'
seq = []
for ln, line in enumerate(lines[start + 1:end], start + 2):
    t = line.strip()
    if not t or t.startswith(('.', ';')):
        continue
    if t.endswith(':'):
        seq.append(('B', t, ln)); continue
    seq.append((cls(t.split()[0]), t.split()[0], ln))
tot = collections.Counter(c for c, _, _ in seq)
print('totals', dict(tot))
run = []; mf = 0
def flush():
    global run, mf
    if len(run) >= min_len:
        c = collections.Counter(x[0] for x in run)
        print('  after %4d MFMA: %4d non-MFMA  lines %d-%d  %s' % (mf, len(run), run[0][2], run[-1][2], dict(c)))
    run = []
for item in seq:
    if item[0] == 'M':
        flush(); mf += 1
    else:
        run.append(item)
flush()
