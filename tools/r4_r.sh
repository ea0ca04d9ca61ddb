#!/bin/bash
mkdir -p gpurun_out/r4r
timeout -k 10 300 python tools/pinn2d_step_time.py > gpurun_out/r4r/pinn2d_step_time.txt 2>&1
echo "rc $?"; grep -v amdgpu gpurun_out/r4r/pinn2d_step_time.txt | tail -5
