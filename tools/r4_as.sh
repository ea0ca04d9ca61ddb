#!/bin/bash
out=gpurun_out/r4as; mkdir -p $out
t0=$(date +%s)
timeout -k 10 900 python -m pytest tests -q -m gpu --durations=8 > $out/pytest_gpu_full.log 2>&1
echo "full GPU suite rc $? in $(( $(date +%s) - t0 )) s"; tail -12 $out/pytest_gpu_full.log
for wl in cfg3_2d_5x128 cfg4_2d_6x128_rot; do
  python3 bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_$wl.json 2> $out/bench_$wl.err; echo "bench $wl rc $?"
done
python3 bench.py --steps 20 --warmup 5 > $out/bench_ns_2d_4x64.json 2> $out/bench_ns.err; echo "bench ns rc $?"; cut -c1-200 $out/bench_ns_2d_4x64.json
