#!/bin/bash
# multi-tile forward kernel for H = 128 (w_forward_mt): parity first, then A/B against f_forward_coop<128> and w_forward
out=gpurun_out/r4m; mkdir -p $out
export GPE_HIP_LIB=$PWD/build/variants/libgpe_widemt.so
GPE_WIDE=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "128 or cfg4 or complex" > $out/pytest_mt.log 2>&1; rc=$?; echo "parity with w_forward_mt (GPE_WIDE=1) rc $rc $(tail -1 $out/pytest_mt.log)"
if [ $rc -ne 0 ]; then tail -30 $out/pytest_mt.log; exit 1; fi
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
    python bench.py --workload $w $B --parity-points 8192 > $out/coop_$w.json 2> $out/coop_$w.err; line "f_forward_coop $w" $out/coop_$w.json
    GPE_WIDE=1 python bench.py --workload $w $B --parity-points 8192 > $out/mt_$w.json 2> $out/mt_$w.err; line "w_forward_mt $w" $out/mt_$w.json
    GPE_WIDE=1 GPE_WIDE_FWD_MT=0 python bench.py --workload $w $B --parity-points 8192 > $out/w1_$w.json 2> $out/w1_$w.err; line "w_forward (1 tile) $w" $out/w1_$w.json
  done
done
