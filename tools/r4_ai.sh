#!/bin/bash
mkdir -p gpurun_out/r4ai
timeout -k 10 200 python tools/residual_step_time.py 2>&1 | grep -v amdgpu | tee gpurun_out/r4ai/residual_step_time.txt
