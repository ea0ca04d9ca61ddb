#!/bin/bash
out=gpurun_out/r4v; mkdir -p $out
timeout -k 10 400 python tools/pinn2d_reference_size.py > $out/pinn2d_reference_size.txt 2>&1
echo "rc $?"; grep -v amdgpu $out/pinn2d_reference_size.txt | tail
