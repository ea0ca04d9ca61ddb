#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (VGPR / scratch / occupancy per kernel)."""
import re
import sys

txt = open(sys.argv[1]).read()
blocks = re.split(r'remark: [^\n]*Function Name: ', txt)[1:]
K = {'v': r'VGPRs', 'a': r'AGPRs', 's': r'ScratchSize \[bytes/lane\]', 'o': r'Occupancy \[waves/SIMD\]',
     'l': r'LDS Size \[bytes/block\]', 'sg': r'SGPRs'}
for b in blocks:
    name = b.split('\n')[0].strip()
    vals = {}
    for k, pat in K.items():
        m = re.search(pat + r': (\d+)', b)
        vals[k] = m.group(1) if m else '?'
    print("%-70s VGPR %4s AGPR %4s SGPR %4s scratch %5s occ %2s LDS %6s" % (name[:70], vals['v'], vals['a'], vals['sg'],
                                                                         vals['s'], vals['o'], vals['l']))
