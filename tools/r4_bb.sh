#!/bin/bash
# final build of round 4: bench lines first (NS with cpu_baseline, cfg2, cfg1, cfg3), small batches, then the whole GPU suite
o=gpurun_out/r4bb; mkdir -p $o
timeout -k 10 250 python3 bench.py --steps 20 --warmup 5 > $o/bench_ns_2d_4x64.json 2> $o/bench_ns.err && cut -c1-230 $o/bench_ns_2d_4x64.json || exit 1
for wl in cfg2_1d_4x64 cfg1_1d_4x32 cfg3_2d_5x128; do
  timeout -k 10 200 python3 bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline > $o/bench_$wl.json 2> $o/bench_$wl.err && echo "bench $wl done" || exit 1
done
for n in 2048 4000 16384 131072; do python3 tools/small_n_step.py $n 3000 2>/dev/null >> $o/small_batch.txt; done
for n in 2048 4000; do echo -n "[GPE_GRAPH=0] " >> $o/small_batch.txt; GPE_GRAPH=0 python3 tools/small_n_step.py $n 3000 2>/dev/null >> $o/small_batch.txt; done
cat $o/small_batch.txt
timeout -k 10 600 python -m pytest tests -m gpu -x -q --durations=6 > $o/pytest_gpu.log 2>&1; rc=$?; tail -2 $o/pytest_gpu.log; exit $rc
