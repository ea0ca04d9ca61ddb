#!/bin/bash
out=gpurun_out/r4ap; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "four_maps or five_maps or deep_h64 or variants or 2d_64x4_g500 or cfg1" > $out/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $out/pytest.log
for L in 2,64,64,64,64,64,1 2,64,64,64,64,64,64,1 3,64,64,64,64,64,64,1 1,32,32,32,32,32,32,1; do
  for n in 4096 65536 1048576; do
    st=200; [ $n = 1048576 ] && st=20
    python tools/step_time_nd.py $L $n $st 2>&1 | grep -v amdgpu | tail -1 | cut -c1-170
  done
done | tee $out/deep_nets_after.txt
