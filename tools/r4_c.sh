#!/bin/bash
# round 4, third GPU call: fused reduce+update, multi-step graphs, wide-kernel A/B (buffer addressing, LDS software pipeline), driver divergence calibration
out=gpurun_out/r4c
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_surface.py -q -m gpu -k "vary_beta or update_kernel_forms or side_stream_and_graph or history_and_run or kernel_variants_agree or golden_refine_trace or golden_notebook_trace or cosine_loss" > $out/pytest_sel.log 2>&1
echo "pytest sel rc $?"; tail -8 $out/pytest_sel.log
for n in 2048 4000 16384; do
  for env in "" "GPE_FUSE_UPDATE=0" "GPE_GRAPH=1" "GPE_GRAPH=1 GPE_GRAPH_STEPS=32" "GPE_GRAPH=1 GPE_GRAPH_STEPS=1"; do
    echo -n "[$env] " >> $out/small_batch.txt; env $env python3 tools/small_n_step.py $n 3200 >> $out/small_batch.txt 2>&1
  done
done
cat $out/small_batch.txt
line() { python - "$@" <<'PY'
import json,sys
tag,f=sys.argv[1:3]
try:
    a=json.loads(open(f).read().strip().splitlines()[-1])
    pc=a.get("parity_check",{})
    print("%-44s %.4f ms/step  fwd %.4f (%.3f)  bwd %.4f (%.3f)  parity %s"%(tag,a["ms_per_step"],a["roofline_forward"]["avg_launch_ms"],a["roofline_forward"]["frac"],a["roofline"]["avg_launch_ms"],a["roofline"]["frac"],pc.get("ok")))
except Exception as ex: print(tag,"ERR",ex)
PY
}
B="--steps 20 --warmup 5 --no-cpu-baseline --no-alt-mode --blocks 15"
for rep in 1 2; do
for v in wideold base wideswp; do
  for w in cfg3_2d_5x128 cfg4_2d_6x128_rot; do
    if [ $v = base ]; then unset GPE_HIP_LIB; else export GPE_HIP_LIB=$PWD/build/variants/libgpe_$v.so; fi
    python bench.py --workload $w $B --parity-points 8192 > $out/ab_${v}_$w.json 2> $out/ab_${v}_$w.err; line "$v $w" $out/ab_${v}_$w.json
  done
done
done
unset GPE_HIP_LIB
for v in wideold base wideswp; do
    if [ $v = base ]; then unset GPE_HIP_LIB; else export GPE_HIP_LIB=$PWD/build/variants/libgpe_$v.so; fi
    python bench.py --workload cfg5_3d_6x256 $B --no-parity-check > $out/ab_${v}_cfg5.json 2> $out/ab_${v}_cfg5.err; line "$v cfg5" $out/ab_${v}_cfg5.json
done
unset GPE_HIP_LIB
B2="--steps 20 --warmup 5 --no-cpu-baseline --no-alt-mode --no-parity-check"
python bench.py --workload cfg2_1d_4x64 $B2 > $out/cfg2_default.json 2>/dev/null; line "cfg2 default (fused update)" $out/cfg2_default.json
GPE_FUSE_UPDATE=0 python bench.py --workload cfg2_1d_4x64 $B2 > $out/cfg2_nofu.json 2>/dev/null; line "cfg2 FUSE_UPDATE=0" $out/cfg2_nofu.json
GPE_COOP_FWD_MAX_TILES=8192 GPE_FUSE_HEAD_MAX=70000 python bench.py --workload cfg2_1d_4x64 $B2 > $out/cfg2_coophead.json 2>/dev/null; line "cfg2 coop fwd 8192 + head 70000" $out/cfg2_coophead.json
python bench.py --workload cfg1_1d_4x32 $B2 > $out/cfg1_default.json 2>/dev/null; line "cfg1 default" $out/cfg1_default.json
GPE_GRAPH=1 python bench.py --workload cfg1_1d_4x32 $B2 > $out/cfg1_graph.json 2>/dev/null; line "cfg1 GRAPH=1 (8 steps)" $out/cfg1_graph.json
echo "--- driver divergence, default kernels"; timeout -k 10 600 python tools/driver_divergence.py > $out/driver_divergence.txt 2>&1; cat $out/driver_divergence.txt
echo "--- driver divergence, split-bf16 kernels forced"; GPE_FWD_B6=1 GPE_BWD_B6=1 GPE_COOP_FWD_MAX_TILES=0 timeout -k 10 600 python tools/driver_divergence.py > $out/driver_divergence_b6.txt 2>&1; cat $out/driver_divergence_b6.txt
