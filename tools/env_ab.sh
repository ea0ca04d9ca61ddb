#!/bin/bash
# usage (GPU box): tools/env_ab.sh <workload> VAR=val[,VAR=val...] ...   -- bench once per environment setting ("-" = default), same box
wl=$1; shift
mkdir -p gpurun_out/envab
for spec in "$@"; do
  tag=$(echo "$spec" | tr '=,' '__')
  if [ "$spec" = "-" ]; then envs=""; else envs=$(echo "$spec" | tr ',' ' '); fi
  env $envs timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --no-alt-mode --blocks 20 > gpurun_out/envab/${wl}_$tag.json 2> gpurun_out/envab/${wl}_$tag.err
  python -c "
import json
j=json.loads(open('gpurun_out/envab/${wl}_$tag.json').read().strip().splitlines()[-1])
print('$spec', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['avg_launch_ms'], j['roofline_forward']['avg_launch_ms'], j['roofline_forward']['kernel'], j.get('parity_check',{}).get('ok'))"
done
