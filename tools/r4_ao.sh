#!/bin/bash
# rocprofv3 --kernel-trace --stats of the final build, all five workloads, a longer run than tools/profile_round4.sh's (105 steps instead of 26: the first
# dispatches of a process are slower -- cold caches, clock ramp -- and weighed on the short run's average)
out=gpurun_out/r4ao; mkdir -p $out; R=$PWD
export TMPDIR=/tmp
for wl in ns_2d_4x64 cfg3_2d_5x128 cfg4_2d_6x128_rot cfg5_3d_6x256 cfg2_1d_4x64; do
  st=30; [ $wl = cfg5_3d_6x256 ] && st=8
  d=$(mktemp -d /tmp/ks.XXXX)
  (cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/bench.py --workload $wl --steps $st --warmup 5 --blocks 3 --no-parity-check --no-cpu-baseline --no-alt-mode > $R/$out/bench_${wl}_under_rocprof.json 2> $R/$out/stats_$wl.err) || { echo "$wl failed"; tail -3 $out/stats_$wl.err; exit 1; }
  cp $d/*/*kernel_stats.csv $out/kernel_stats_$wl.csv; rm -rf $d
  echo "== $wl"; python3 tools/kstats.py $out/kernel_stats_$wl.csv | head -8
done
