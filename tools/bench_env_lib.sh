#!/bin/bash
# usage: tools/bench_env_lib.sh <variant-name> "ENV=a ENV2=b" ...   -- NS bench line + small-batch step time per environment
lib=$PWD/build/variants/libgpe_$1.so; shift
for envs in "$@"; do
  env GPE_HIP_LIB=$lib $envs python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/be.json 2> gpurun_out/be.err
  python - "$envs" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/be.json").read().strip().splitlines()[-1])
    print("%-40s %.4g pts/s %.3f ms/step bwd %.3f ms (%.3f) fwd %.3f ms" % (sys.argv[1], d["value"], d["ms_per_step"],
          d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline_forward"]["avg_launch_ms"]))
except Exception as ex:
    print(sys.argv[1], "ERR", ex, open("gpurun_out/be.err").read()[-300:])
PY
done
