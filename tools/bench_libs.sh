#!/bin/bash
# usage: tools/bench_libs.sh name [name ...]   -- NS bench line of build/variants/libgpe_<name>.so, one summary line each
for v in "$@"; do
  GPE_HIP_LIB=$PWD/build/variants/libgpe_$v.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/v_$v.json 2> gpurun_out/v_$v.err
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/v_%s.json" % v).read().strip().splitlines()[-1])
    print("%-12s %.4g pts/s %.3f ms/step bwd %.3f ms (%.3f) fwd %.3f ms loss %.6f mu %.6f" % (v, d["value"], d["ms_per_step"],
          d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline_forward"]["avg_launch_ms"], d["final_loss"], d["final_mu"]))
except Exception as ex:
    print(v, "ERR", ex, open("gpurun_out/v_%s.err" % v).read()[-400:])
PY
done
