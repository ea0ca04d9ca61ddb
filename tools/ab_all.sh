#!/bin/bash
# usage: tools/ab_all.sh <lib_a> <lib_b>   -- bench.py of every workload with two library builds, A B A B, same box
for wl in ns_2d_4x64 cfg2_1d_4x64 cfg3_2d_5x128 cfg4_2d_6x128_rot cfg5_3d_6x256 cfg1_1d_4x32; do
  echo "== $wl"
  bash tools/ab_bench.sh $wl "$1" "$2" "$1" "$2"
done
