#!/bin/bash
out=gpurun_out/r4u; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "native_rccl_step or shard_additivity or large_batch" > $out/pytest_new2.log 2>&1
echo "pytest rc $?"; tail -12 $out/pytest_new2.log
