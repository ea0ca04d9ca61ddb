#!/bin/bash
# usage: tools/bench_variants.sh <tag> [variant.so ...]   -- bench the default build and each variant, print a table
tag=$1; shift
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_${tag}_default.json 2> gpurun_out/bench_${tag}.err
for v in "$@"; do
  n=$(basename $v .so)
  GPE_HIP_LIB=$PWD/$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_${tag}_${n}.json 2>> gpurun_out/bench_${tag}.err
done
python - "$tag" <<'PY'
import json,glob,sys
tag=sys.argv[1]
for f in sorted(glob.glob(f"gpurun_out/bench_{tag}_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print("%-40s %.4g pts/s  %.3f ms/step  bwd %.3f ms (%.3f)  fwd %.3f ms (%.3f)"%(f.split('/')[-1], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline_forward"]["avg_launch_ms"], d["roofline_forward"]["frac"]))
    except Exception as ex: print(f,"ERR",ex)
PY
