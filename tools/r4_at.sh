#!/bin/bash
out=gpurun_out/r4at; mkdir -p $out
for r in 1 2; do
for lib in fastbase wreg4; do
  for n in 4096 1048576; do
    st=200; [ $n = 1048576 ] && st=20
    echo -n "$lib  "; GPE_HIP_LIB=$PWD/build/variants/libgpe_$lib.so python tools/step_time_nd.py 2,64,64,64,64,64,64,1 $n $st 2>&1 | grep -v amdgpu | tail -1 | cut -c1-120
  done
done; done | tee $out/wreg_ab.txt
