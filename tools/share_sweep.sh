#!/bin/bash
# usage (GPU box): [VAR=GPE_FWD_SHARE] tools/share_sweep.sh <workload> <share> ...  -- bench with GPE_PIPE_SHARE (or $VAR) = share: uneven split of a
# CU's tiles between its two workgroups, /1024
wl=$1; shift
mkdir -p gpurun_out/share
for v in "$@"; do
  env ${VAR:-GPE_PIPE_SHARE}=$v timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --no-alt-mode --blocks 20 > gpurun_out/share/${wl}_$v.json 2> gpurun_out/share/${wl}_$v.err
  python -c "
import json
j=json.loads(open('gpurun_out/share/${wl}_$v.json').read().strip().splitlines()[-1])
print('${VAR:-GPE_PIPE_SHARE} $v', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['avg_launch_ms'], j.get('parity_check',{}).get('ok'))"
done
