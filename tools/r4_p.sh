#!/bin/bash
out=gpurun_out/r4p; mkdir -p $out
line() { python - "$@" <<'PY'
import json,sys
tag,f=sys.argv[1:3]
try:
    a=json.loads(open(f).read().strip().splitlines()[-1])
    pc=a.get("parity_check",{})
    print("%-40s %.4f ms/step  fwd %.4f (%.3f)  bwd %.4f (%.3f)  parity %s  %.4g pts/s  %s"%(tag,a["ms_per_step"],a["roofline_forward"]["avg_launch_ms"],a["roofline_forward"]["frac"],a["roofline"]["avg_launch_ms"],a["roofline"]["frac"],pc.get("ok"),a["value"],a["roofline_forward"]["kernel"]))
except Exception as ex: print(tag,"ERR",ex)
PY
}
B="--steps 20 --warmup 5 --no-cpu-baseline --no-alt-mode --blocks 15"
for rep in 1 2; do
  for w in cfg3_2d_5x128 cfg4_2d_6x128_rot; do
    unset GPE_HIP_LIB
    python bench.py --workload $w $B --parity-points 8192 > $out/base_$w.json 2> $out/base_$w.err; line "default $w" $out/base_$w.json
    GPE_HIP_LIB=$PWD/build/variants/libgpe_fswp.so python bench.py --workload $w $B --parity-points 8192 > $out/fswp_$w.json 2> $out/fswp_$w.err; line "FCOOP_SWP $w" $out/fswp_$w.json
  done
done
