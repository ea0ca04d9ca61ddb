#!/bin/bash
out=gpurun_out/r4i; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "step_matches_oracle or update_kernel_forms" > $out/pytest_step.log 2>&1; echo "step tests rc $? $(tail -1 $out/pytest_step.log)"
t0=$(date +%s)
timeout -k 10 1000 python -m pytest tests -q -m gpu --durations=12 > $out/pytest_gpu_full.log 2>&1
echo "full GPU suite rc $? in $(( $(date +%s) - t0 )) s"; tail -25 $out/pytest_gpu_full.log
