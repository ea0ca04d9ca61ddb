#!/bin/bash
out=gpurun_out/r4t; mkdir -p $out
t0=$(date +%s)
timeout -k 10 1000 python -m pytest tests -q -m gpu --durations=12 > $out/pytest_gpu_full.log 2>&1
echo "full GPU suite rc $? in $(( $(date +%s) - t0 )) s"; tail -22 $out/pytest_gpu_full.log
