#!/bin/bash
# round 4: the 2D classes' loss (energy-functional lambda + regularisers) on the GPU -- new parity cases, goldens, surface
mkdir -p gpurun_out/r4q
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_surface.py -x -q -m gpu -k "energy or class or reg_f or pinn2d or 2d_riesz or 2d_64x4_g500 or 2d_N1 or golden_2d" > gpurun_out/r4q/pytest_new.log 2>&1
echo "pytest new rc $?"; tail -15 gpurun_out/r4q/pytest_new.log
