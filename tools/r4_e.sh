#!/bin/bash
# round 4, fifth GPU call: split update (small P), per-lane small gradients in w_bwd_map, then the WHOLE GPU suite (timed)
out=gpurun_out/r4e
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_surface.py -q -m gpu -x -k "update_kernel_forms or side_stream_and_graph or history_and_run or kernel_variants_agree or golden_refine_trace or golden_notebook_trace or golden_trace or vary_beta_driver or step_matches_oracle" > $out/pytest_sel.log 2>&1
echo "pytest sel rc $?"; tail -5 $out/pytest_sel.log
for n in 2048 4000 16384 65536; do
  for env in "" "GPE_SPLIT_UPDATE=0" "GPE_GRAPH=0"; do echo -n "[$env] " >> $out/small_batch.txt; env $env python3 tools/small_n_step.py $n 3200 >> $out/small_batch.txt 2>&1; done
done
cat $out/small_batch.txt
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
for w in cfg1_1d_4x32 cfg2_1d_4x64; do
  python bench.py --workload $w $B > $out/bench_$w.json 2> $out/bench_$w.err; line "$w" $out/bench_$w.json
  GPE_SPLIT_UPDATE=0 python bench.py --workload $w $B --no-parity-check > $out/bench_${w}_nosplit.json 2>/dev/null; line "$w SPLIT_UPDATE=0" $out/bench_${w}_nosplit.json
done
GPE_FUSE_SEED_MAX=60000 python bench.py --workload cfg2_1d_4x64 $B --no-parity-check > $out/bench_cfg2_noseedf.json 2>/dev/null; line "cfg2 seeds by k_seed_pde" $out/bench_cfg2_noseedf.json
for rep in 1 2; do
for v in widebuf base; do
  for w in cfg3_2d_5x128 cfg4_2d_6x128_rot; do
    if [ $v = base ]; then unset GPE_HIP_LIB; else export GPE_HIP_LIB=$PWD/build/variants/libgpe_$v.so; fi
    python bench.py --workload $w $B --parity-points 8192 > $out/ab_${v}_$w.json 2> $out/ab_${v}_$w.err; line "$v $w" $out/ab_${v}_$w.json
  done
done
done
unset GPE_HIP_LIB
t0=$(date +%s)
timeout -k 10 1100 python -m pytest tests -q -m gpu --durations=25 > $out/pytest_gpu_full.log 2>&1
echo "full GPU suite rc $? in $(( $(date +%s) - t0 )) s"; tail -40 $out/pytest_gpu_full.log
