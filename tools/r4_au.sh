#!/bin/bash
mkdir -p gpurun_out/r4au
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "variants" > gpurun_out/r4au/pytest.log 2>&1; echo "rc $?"; tail -5 gpurun_out/r4au/pytest.log
