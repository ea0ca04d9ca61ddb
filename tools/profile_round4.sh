#!/bin/bash
# usage (GPU box, repo root): tools/profile_round4.sh <outdir under gpurun_out> [part]
# Everything profiles/r04/ holds besides the accuracy records.  part 1: bench lines of every workload (plain and data-parallel at world 1),
# small batches; part 2: rocprofv3 kernel stats + PMC passes (SQ counters, FETCH / WRITE) of NS, cfg3 and cfg5.
out=$1; part=${2:-1}
mkdir -p $out
R=$PWD
export TMPDIR=/tmp
DP="RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611"
if [ $part = 1 ]; then
  for wl in cfg1_1d_4x32 cfg2_1d_4x64 cfg3_2d_5x128 cfg4_2d_6x128_rot cfg5_3d_6x256; do
    python3 bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_$wl.json 2> $out/bench_$wl.err; echo "bench $wl done"
  done
  for wl in ns_2d_4x64 cfg3_2d_5x128 cfg5_3d_6x256 cfg2_1d_4x64; do
    env $DP python3 bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-alt-mode --no-parity-check > $out/dp_world1_$wl.json 2> $out/dp_world1_$wl.err; echo "dp $wl done"
  done
  for n in 2048 4000 16384 131072; do python3 tools/small_n_step.py $n 3000 >> $out/small_batch.txt; done
  for n in 2048 4000 16384; do echo -n "[GPE_GRAPH=0] " >> $out/small_batch.txt; GPE_GRAPH=0 python3 tools/small_n_step.py $n 3000 >> $out/small_batch.txt; done
  for n in 2048 4000 16384; do echo -n "[GPE_FUSE_UPDATE=1] " >> $out/small_batch.txt; GPE_FUSE_UPDATE=1 python3 tools/small_n_step.py $n 3000 >> $out/small_batch.txt; done
  cat $out/small_batch.txt
  python3 bench.py --steps 20 --warmup 5 > $out/bench_ns_2d_4x64.json 2> $out/bench_ns.err
  cut -c1-300 $out/bench_ns_2d_4x64.json
else
  for wl in ns_2d_4x64 cfg3_2d_5x128 cfg5_3d_6x256; do
    d=$(mktemp -d /tmp/ks.XXXX)
    (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/bench.py --workload $wl --steps 10 --warmup 3 --blocks 2 --no-parity-check --no-cpu-baseline --no-alt-mode > $R/$out/bench_${wl}_under_rocprof.json 2> $R/$out/stats_$wl.err)
    cp $d/*/*kernel_stats.csv $out/kernel_stats_$wl.csv; rm -rf $d
    echo "stats $wl done"
  done
  tools/pmc_sq.sh ns_2d_4x64 $out/pmc 1048576 > $out/pmc_ns.log 2>&1; echo "pmc ns done"
  tools/pmc_sq.sh cfg3_2d_5x128 $out/pmc 131072 > $out/pmc_cfg3.log 2>&1; echo "pmc cfg3 done"
  tools/pmc_sq.sh cfg5_3d_6x256 $out/pmc 524288 > $out/pmc_cfg5.log 2>&1; echo "pmc cfg5 done"
  cp $out/pmc/*.json $out/ 2>/dev/null
  rm -rf $out/pmc
fi
