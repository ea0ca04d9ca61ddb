#!/bin/bash
out=gpurun_out/r4am; mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_surface.py -x -q -m gpu -k "padded_hidden or error_behaviour or pads_to or refine or vary_beta or box" > $out/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $out/pytest.log
