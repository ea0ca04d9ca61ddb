#!/bin/bash
# effective clock of the dominant kernels (one counter, GRBM_GUI_ACTIVE, with the kernel trace of the same dispatches)
R=$PWD; out=$PWD/gpurun_out/r4s; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
for wl in ns_2d_4x64 cfg3_2d_5x128 cfg5_3d_6x256; do
  st=30; [ $wl = cfg5_3d_6x256 ] && st=8
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/clk_$wl -- python3 $R/bench.py --workload $wl --steps $st --warmup 5 --blocks 1 --no-parity-check --no-cpu-baseline --no-alt-mode > $out/clk_$wl.json 2> $out/clk_$wl.err || { echo "$wl failed"; tail -3 $out/clk_$wl.err; exit 1; }
  echo "== $wl"
  python3 $R/tools/clock_summary.py $out/clock_$wl.json $out/clk_$wl/*/*counter_collection.csv $(ls $out/clk_$wl/*/*kernel_trace.csv 2>/dev/null | head -1)
  head -2 $(ls $out/clk_$wl/*/*counter_collection.csv | head -1) > $out/clk_${wl}_csv_head.txt
  rm -rf $out/clk_$wl
done
