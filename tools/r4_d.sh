#!/bin/bash
# round 4, fourth GPU call: rewritten driver tests (default / split-bf16), diet in the forward kernels, graph default, accuracy schedules
out=gpurun_out/r4d
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_surface.py tests/test_gpu_parity.py -q -m gpu -k "vary_beta or driver or update_kernel_forms or side_stream_and_graph or history_and_run or kernel_variants_agree or class_surface or gravity or box" > $out/pytest_sel.log 2>&1
echo "pytest sel rc $?"; tail -6 $out/pytest_sel.log
GPE_FWD_B6=1 GPE_BWD_B6=1 GPE_COOP_FWD_MAX_TILES=0 timeout -k 10 600 python -m pytest tests/test_gpu_surface.py -q -m gpu -k "notebook_driver_reproduces or refine_driver_against or vary_beta_driver" > $out/pytest_drivers_b6.log 2>&1
echo "pytest drivers (split-bf16 forced) rc $?"; tail -4 $out/pytest_drivers_b6.log
line() { python - "$@" <<'PY'
import json,sys
tag,f=sys.argv[1:3]
try:
    a=json.loads(open(f).read().strip().splitlines()[-1])
    pc=a.get("parity_check",{})
    print("%-44s %.4f ms/step  fwd %.4f (%.3f)  bwd %.4f (%.3f)  parity %s  %.4g pts/s"%(tag,a["ms_per_step"],a["roofline_forward"]["avg_launch_ms"],a["roofline_forward"]["frac"],a["roofline"]["avg_launch_ms"],a["roofline"]["frac"],pc.get("ok"),a["value"]))
except Exception as ex: print(tag,"ERR",ex)
PY
}
B="--steps 20 --warmup 5 --no-cpu-baseline --no-alt-mode --blocks 15"
for w in cfg3_2d_5x128 cfg4_2d_6x128_rot cfg5_3d_6x256 ns_2d_4x64 cfg2_1d_4x64 cfg1_1d_4x32; do
  python bench.py --workload $w $B --parity-budget 30 > $out/bench_$w.json 2> $out/bench_$w.err; line "$w" $out/bench_$w.json
done
for n in 2048 4000 16384; do python3 tools/small_n_step.py $n 3200 >> $out/small_batch.txt 2>&1; GPE_GRAPH=0 python3 tools/small_n_step.py $n 3200 >> $out/small_batch.txt 2>&1; done; cat $out/small_batch.txt
# shorter in-suite accuracy schedules
for sch in "--epochs 1500 --final 25000 --stages 12" "--epochs 2500 --final 40000 --stages 12"; do
  t0=$(date +%s); python tools/accuracy_nd.py --case cfg3_2d $sch --out $out/acc_cfg3_short.json > $out/acc_cfg3_short.log 2>&1; echo "cfg3 [$sch] $(( $(date +%s) - t0 )) s: $(tail -1 $out/acc_cfg3_short.log)"
done
