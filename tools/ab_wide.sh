#!/bin/bash
# usage (on the GPU box): tools/ab_wide.sh <variant> ...   -- bench cfg3 / cfg4 with libgpe_hip.so ("base") and build/variants/libgpe_<variant>.so
mkdir -p gpurun_out/ab
for v in base "$@"; do
  for w in cfg3_2d_5x128 cfg4_2d_6x128_rot; do
    if [ $v = base ]; then unset GPE_HIP_LIB; else export GPE_HIP_LIB=build/variants/libgpe_$v.so; fi
    timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline --blocks 10 > gpurun_out/ab/${v}_$w.json 2> gpurun_out/ab/${v}_$w.err
    python -c "
import json
j=json.loads(open('gpurun_out/ab/${v}_$w.json').read().strip().splitlines()[-1])
print('$v','$w',j['value'],j['ms_per_step'],j['roofline']['frac'],j.get('parity_check',{}).get('ok'))"
  done
done
