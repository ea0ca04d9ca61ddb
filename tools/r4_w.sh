#!/bin/bash
out=gpurun_out/r4w; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_surface.py -x -q -m gpu -k "odd_width or reference_2d_arch or ragged or pads_to or padded_hidden or golden_2d or pinn2d or native_rccl or shard_additivity" > $out/pytest_pad.log 2>&1
echo "pytest rc $?"; tail -15 $out/pytest_pad.log
timeout -k 10 300 python tools/pinn2d_reference_size.py > $out/pinn2d_reference_size.txt 2>&1
echo "rc $?"; grep -v amdgpu $out/pinn2d_reference_size.txt | tail
