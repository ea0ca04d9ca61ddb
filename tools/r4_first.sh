#!/bin/bash
# round 4, first GPU call: new parity tests, full-batch in-run oracle check, data-parallel step at world 1 beside the plain step
out=gpurun_out/r4a
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "large_batch_launch or native_rccl or stale_gradient or uneven_tile or kernel_variants_agree" > $out/pytest_sel.log 2>&1
echo "pytest sel rc $?"; tail -3 $out/pytest_sel.log
timeout -k 10 600 python -m pytest tests/test_bench_contract.py tests/test_gpu_dp.py -x -q -m gpu > $out/pytest_bench.log 2>&1
echo "pytest bench rc $?"; tail -3 $out/pytest_bench.log
for wl in ns_2d_4x64 cfg3_2d_5x128 cfg2_1d_4x64; do
  python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-alt-mode > $out/bench_$wl.json 2> $out/bench_$wl.err
  RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-alt-mode --no-parity-check > $out/dp_world1_$wl.json 2> $out/dp_world1_$wl.err
  python - $out $wl <<'PY'
import json,sys
o,w=sys.argv[1:]
a=json.loads(open(f"{o}/bench_{w}.json").read().strip().splitlines()[-1]); b=json.loads(open(f"{o}/dp_world1_{w}.json").read().strip().splitlines()[-1])
pc=a.get("parity_check",{})
print(w,"plain %.4f ms  dp(world 1) %.4f ms  (+%.2f %%)  fwd %.3f/%.3f bwd %.3f/%.3f | parity ok=%s pts=%s same=%s split=%s oracle %.0fs"%(a["ms_per_step"],b["ms_per_step"],100*(b["ms_per_step"]/a["ms_per_step"]-1),
  a["roofline_forward"]["avg_launch_ms"],b["roofline_forward"]["avg_launch_ms"],a["roofline"]["avg_launch_ms"],b["roofline"]["avg_launch_ms"],pc.get("ok"),pc.get("points"),pc.get("same_kernels_as_timed"),pc.get("uneven_split"),pc.get("oracle_seconds",0)))
PY
done
