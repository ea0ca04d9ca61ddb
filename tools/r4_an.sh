#!/bin/bash
out=gpurun_out/r4an; mkdir -p $out
for L in 2,64,64,64,64,1 2,64,64,64,64,64,1 2,64,64,64,64,64,64,1 2,128,128,128,128,128,1 2,128,128,128,128,128,128,1; do
  for n in 4096 65536 1048576; do
    st=200; [ $n = 1048576 ] && st=20
    python tools/step_time_nd.py $L $n $st 2>&1 | grep -v amdgpu | tail -1 | cut -c1-170
  done
done | tee $out/deep_nets.txt
