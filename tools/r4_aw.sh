#!/bin/bash
mkdir -p gpurun_out/r4aw
for sd in 3 4 5 6; do python tools/fuzz_parity.py 300 $sd 7 2>&1 | grep -v amdgpu | grep "BAD\|EXC\|cases" | cut -c1-300; done | tee gpurun_out/r4aw/fuzz_seeds.txt
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "randomised" 2>&1 | tail -2
