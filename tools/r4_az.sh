#!/bin/bash
# final records of round 4 on the build with slab_column_sum: GPU suite, bench lines (NS with cpu_baseline, cfg1 / cfg2 / cfg3), kernel stats of NS and cfg2, small batches
o=gpurun_out/r4az; mkdir -p $o; R=$PWD; export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests -m gpu -x -q --durations=8 > $o/pytest_gpu.log 2>&1; rc=$?; tail -2 $o/pytest_gpu.log
[ $rc = 0 ] || exit $rc
for wl in cfg1_1d_4x32 cfg2_1d_4x64 cfg3_2d_5x128; do
  timeout -k 10 200 python3 bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline > $o/bench_$wl.json 2> $o/bench_$wl.err && echo "bench $wl done" || exit 1
done
timeout -k 10 250 python3 bench.py --steps 20 --warmup 5 > $o/bench_ns_2d_4x64.json 2> $o/bench_ns.err && cut -c1-260 $o/bench_ns_2d_4x64.json || exit 1
for wl in ns_2d_4x64 cfg2_1d_4x64; do
  d=$(mktemp -d /tmp/ks.XXXX)
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/bench.py --workload $wl --steps 20 --warmup 5 --blocks 5 --no-parity-check --no-cpu-baseline --no-alt-mode > $R/$o/bench_${wl}_under_rocprof.json 2> $R/$o/stats_$wl.err) || exit 1
  cp $d/*/*kernel_stats.csv $o/kernel_stats_$wl.csv; rm -rf $d; echo "stats $wl done"; head -6 $o/kernel_stats_$wl.csv | cut -c1-60,200-330
done
for n in 2048 4000 16384 131072; do python3 tools/small_n_step.py $n 3000 2>/dev/null >> $o/small_batch.txt; done
for n in 2048 4000; do echo -n "[GPE_GRAPH=0] " >> $o/small_batch.txt; GPE_GRAPH=0 python3 tools/small_n_step.py $n 3000 2>/dev/null >> $o/small_batch.txt; done
cat $o/small_batch.txt
