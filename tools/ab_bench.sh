#!/bin/bash
# usage: tools/ab_bench.sh <workload> <lib1> <lib2> ...   -- bench.py (no profiler, 30 steps) with alternative library builds, same box
wl=$1; shift
for lib in "$@"; do
  GPE_HIP_LIB=$PWD/$lib python3 bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-alt-mode 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; f=d.get('roofline_forward',{})
print('%-34s %8.3f ms/step  bwd %7.3f ms (%.3f)  fwd %7.3f ms (%.3f)' % ('$lib'.split('/')[-1], d['ms_per_step'], r['avg_launch_ms'], r['frac'], f.get('avg_launch_ms',0), f.get('frac',0)))"
done
