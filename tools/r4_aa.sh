#!/bin/bash
out=gpurun_out/r4aa; mkdir -p $out
for v in 1 0; do
  echo "== GPE_MERGE_BC=$v"
  GPE_MERGE_BC=$v timeout -k 10 200 python tools/pinn2d_reference_size.py 2>&1 | grep -v amdgpu | grep "100, 100\|128, 128\|64, 64" | cut -c1-110
done | tee $out/merge_ab.txt
