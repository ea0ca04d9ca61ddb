#!/bin/bash
# usage (GPU box, repo root): tools/profile_round3.sh <outdir under gpurun_out> [quick]
# Everything profiles/r03/ holds besides the accuracy records: bench lines of every workload (median of repeated K-step blocks, in-run
# parity_check), rocprofv3 kernel stats (NS, cfg5), SQ-counter + HBM-traffic passes (NS, cfg5), the small-batch step times.
out=$1
mkdir -p $out
R=$PWD
export TMPDIR=/tmp
for wl in ns_2d_4x64 cfg5_3d_6x256; do
  d=$(mktemp -d /tmp/ks.XXXX)
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/bench.py --workload $wl --steps 10 --warmup 3 --blocks 2 --no-parity-check --no-cpu-baseline --no-alt-mode > $R/$out/bench_${wl}_under_rocprof.json 2> $R/$out/stats_$wl.err)
  cp $d/*/*kernel_stats.csv $out/kernel_stats_$wl.csv; rm -rf $d
  echo "stats $wl done"
done
tools/pmc_sq.sh ns_2d_4x64 $out/pmc 1048576 > $out/pmc_ns.log 2>&1; echo "pmc ns done"
tools/pmc_sq.sh cfg5_3d_6x256 $out/pmc 524288 > $out/pmc_cfg5.log 2>&1; echo "pmc cfg5 done"
cp $out/pmc/*.json $out/ 2>/dev/null
for wl in cfg1_1d_4x32 cfg2_1d_4x64 cfg3_2d_5x128 cfg4_2d_6x128_rot cfg5_3d_6x256; do
  python3 bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_$wl.json 2> $out/bench_$wl.err
  echo "bench $wl done"
done
python3 tools/small_n_step.py 2048 3000 > $out/small_batch.txt
python3 tools/small_n_step.py 4000 3000 >> $out/small_batch.txt
python3 tools/small_n_step.py 16384 3000 >> $out/small_batch.txt
python3 tools/small_n_step.py 131072 2000 >> $out/small_batch.txt
cat $out/small_batch.txt
for n in 2048 4000 16384; do GPE_FUSE_HEAD=0 python3 tools/small_n_step.py $n 3000 >> $out/small_batch_nohead.txt; done
python3 bench.py --steps 20 --warmup 5 > $out/bench_ns_2d_4x64.json 2> $out/bench_ns.err
cut -c1-300 $out/bench_ns_2d_4x64.json
rm -rf $out/pmc
