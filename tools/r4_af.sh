#!/bin/bash
out=gpurun_out/r4af; mkdir -p $out
for v in "" "GPE_WIDE=0" "GPE_WIDE=1"; do
  echo "== ${v:-default}"
  env $v timeout -k 10 200 python tools/pinn2d_reference_size.py 2>&1 | grep -v amdgpu | grep "100, 100\|128, 128" | cut -c1-100
done | tee $out/wide_small_ab.txt
# cfg3-like sizes: where does the per-map reverse pass start to win?
for n in 4096 16384 65536; do
  for v in "" "GPE_WIDE=0"; do
    echo -n "${v:-default}: "; env $v python tools/step_time_nd.py 2,128,128,128,128,128,1 $n 300 2>&1 | grep -v amdgpu | tail -1 | cut -c1-160
  done
done | tee -a $out/wide_small_ab.txt
