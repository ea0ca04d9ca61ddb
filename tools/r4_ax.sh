#!/bin/bash
mkdir -p gpurun_out/r4ax
timeout -k 10 600 python tools/fuzz_parity.py 500 31 7 1 dp 2>&1 | grep -v amdgpu | grep "BAD\|EXC\|cases" | cut -c1-330 | tee gpurun_out/r4ax/fuzz_traj.txt
