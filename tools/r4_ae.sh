#!/bin/bash
mkdir -p gpurun_out/r4ae
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "error_behaviour or capi or abi" > gpurun_out/r4ae/pytest.log 2>&1; echo "rc $?"; tail -3 gpurun_out/r4ae/pytest.log
