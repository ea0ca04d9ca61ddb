#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>
# Collects what profiles/<round>/ holds: rocprofv3 kernel stats of the bench command, separate FETCH_SIZE / WRITE_SIZE counter
# passes (HBM traffic), the default bench line (with cpu_baseline), the two other workloads, the small-batch step time.
tag=$1
R=$PWD
out=$R/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/stats.err
echo "stats pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/pmc_fetch.err
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/pmc_write.err
echo "write pass done"
cd $R
python3 tools/pmc_traffic.py $out/pmc_fetch/*/*counter_collection.csv $out/pmc_write/*/*counter_collection.csv 1048576 $out/traffic_ns.json \
  "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
cp $out/stats/*/*kernel_stats.csv $out/bench_ns_kernel_stats.csv
python3 bench.py --workload cfg2_1d_4x64 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_cfg2_1d_4x64.json
python3 bench.py --workload cfg3_2d_5x128 --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_cfg3_2d_5x128.json
python3 tools/small_n_step.py 4000 3000 > $out/small_n.txt
python3 tools/small_n_step.py 16384 3000 >> $out/small_n.txt
python3 tools/small_n_step.py 131072 2000 >> $out/small_n.txt
cat $out/small_n.txt
echo "default bench (with cpu baseline)"
python3 bench.py > $out/bench_ns_default.json
cut -c1-400 $out/bench_ns_default.json
rm -rf $out/stats $out/pmc_fetch $out/pmc_write
