#!/bin/bash
# usage: tools/bench_env.sh <tag> "ENV1=a ENV2=b" "ENV1=c" ...   -- bench.py under each environment setting; print a table
tag=$1; shift
i=0
for envs in "$@"; do
  i=$((i+1))
  env $envs python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_${tag}_${i}.json 2>> gpurun_out/bench_${tag}.err
  python - "gpurun_out/bench_${tag}_${i}.json" "$envs" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-44s %.4g pts/s  %.3f ms/step  bwd %.3f ms (%.3f)  fwd %.3f ms (%.3f)"%(sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline_forward"]["avg_launch_ms"], d["roofline_forward"]["frac"]))
except Exception as ex: print(sys.argv[2],"ERR",ex)
PY
done
