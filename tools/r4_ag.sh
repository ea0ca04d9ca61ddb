#!/bin/bash
out=gpurun_out/r4ag; mkdir -p $out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "128 or variants or pads or cfg3 or cfg4 or per_map or wide" > $out/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $out/pytest.log
for n in 4096 16384 32768 49152 65536; do
  for v in "" "GPE_WIDE_MIN_TILES=0"; do
    echo -n "${v:-default}: "; env $v python tools/step_time_nd.py 2,128,128,128,128,128,1 $n 300 2>&1 | grep -v amdgpu | tail -1 | cut -c1-150
  done
done | tee $out/wide_min_tiles.txt
python tools/pinn2d_reference_size.py 2>&1 | grep -v amdgpu | grep "100, 100" | cut -c1-100 | tee -a $out/wide_min_tiles.txt
