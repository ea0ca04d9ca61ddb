#!/bin/bash
# round 4, second GPU call: beta-sweep tests, data-parallel step with inline collectives, cfg2 threshold sweeps, small-batch baseline
out=gpurun_out/r4b
mkdir -p $out
export TMPDIR=/tmp
R=$PWD
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_surface.py -x -q -m gpu -k "vary_beta or native_rccl or stale_gradient" > $out/pytest_sel.log 2>&1
echo "pytest sel rc $?"; tail -5 $out/pytest_sel.log
timeout -k 10 600 python -m pytest tests/test_bench_contract.py tests/test_gpu_dp.py -x -q -m gpu > $out/pytest_bench.log 2>&1
echo "pytest bench rc $?"; tail -3 $out/pytest_bench.log
line() { python - "$@" <<'PY'
import json,sys
tag,f=sys.argv[1:3]
try:
    a=json.loads(open(f).read().strip().splitlines()[-1])
    print("%-40s %.4f ms/step  fwd %.4f (%.3f)  bwd %.4f (%.3f)  %s"%(tag,a["ms_per_step"],a["roofline_forward"]["avg_launch_ms"],a["roofline_forward"]["frac"],a["roofline"]["avg_launch_ms"],a["roofline"]["frac"],a.get("config",{}).get("exchange")))
except Exception as ex: print(tag,"ERR",ex)
PY
}
B="--steps 20 --warmup 5 --no-cpu-baseline --no-alt-mode --no-parity-check"
DP="RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611"
for wl in ns_2d_4x64 cfg3_2d_5x128 cfg2_1d_4x64 cfg5_3d_6x256; do
  python bench.py --workload $wl $B > $out/plain_$wl.json 2> $out/plain_$wl.err; line "plain $wl" $out/plain_$wl.json
  env $DP python bench.py --workload $wl $B > $out/dp_world1_$wl.json 2> $out/dp_$wl.err; line "dp inline $wl" $out/dp_world1_$wl.json
  env $DP GPE_DP_INLINE=0 python bench.py --workload $wl $B > $out/dp_world1_twostream_$wl.json 2>> $out/dp_$wl.err; line "dp two-stream $wl" $out/dp_world1_twostream_$wl.json
done
# timeline of one data-parallel step of cfg2 at world 1 (kernel trace only)
d=$(mktemp -d /tmp/kt.XXXX)
(cd /tmp && env $DP rocprofv3 --kernel-trace --output-format csv -d $d -- python3 $R/bench.py --workload cfg2_1d_4x64 --steps 20 --warmup 5 --blocks 1 --no-cpu-baseline --no-alt-mode --no-parity-check > $R/$out/dp_trace.json 2> $R/$out/dp_trace.err)
cp $d/*/*kernel_trace.csv $out/dp_world1_cfg2_kernel_trace.csv 2>/dev/null; rm -rf $d
python tools/step_timeline.py $out/dp_world1_cfg2_kernel_trace.csv > $out/dp_world1_cfg2_timeline.txt 2>&1; cat $out/dp_world1_cfg2_timeline.txt
# cfg2: thresholds
for v in 2048 4096 8192; do GPE_COOP_FWD_MAX_TILES=$v python bench.py --workload cfg2_1d_4x64 $B > $out/cfg2_coopfwd_$v.json 2>/dev/null; line "cfg2 COOP_FWD_MAX_TILES=$v" $out/cfg2_coopfwd_$v.json; done
GPE_COOP_FWD_MAX_TILES=4096 GPE_FUSE_HEAD_MAX=65536 python bench.py --workload cfg2_1d_4x64 $B > $out/cfg2_coopfwd_head.json 2>/dev/null; line "cfg2 coop fwd + head in it" $out/cfg2_coopfwd_head.json
GPE_FUSE_SEED=0 python bench.py --workload cfg2_1d_4x64 $B > $out/cfg2_noseedf.json 2>/dev/null; line "cfg2 FUSE_SEED=0" $out/cfg2_noseedf.json
GPE_SHARE_MIN_TILES=4 python bench.py --workload cfg2_1d_4x64 $B > $out/cfg2_share4.json 2>/dev/null; line "cfg2 SHARE_MIN_TILES=4" $out/cfg2_share4.json
GPE_WLDS=0 python bench.py --workload cfg2_1d_4x64 $B > $out/cfg2_wlds0.json 2>/dev/null; line "cfg2 WLDS=0" $out/cfg2_wlds0.json
# small batches
for n in 2048 4000 16384; do python3 tools/small_n_step.py $n 3000 >> $out/small_batch.txt; done; cat $out/small_batch.txt
