#!/bin/bash
# counters for the two workloads that had none: cfg4 (traffic: null in its bench line), cfg2
mkdir -p gpurun_out/r4y
timeout -k 10 500 bash tools/pmc_sq.sh cfg4_2d_6x128_rot gpurun_out/r4y 262144 2>&1 | grep -v amdgpu | tail -12
timeout -k 10 300 bash tools/pmc_sq.sh cfg2_1d_4x64 gpurun_out/r4y 65536 2>&1 | grep -v amdgpu | tail -8
ls gpurun_out/r4y
