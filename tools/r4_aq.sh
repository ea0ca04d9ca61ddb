#!/bin/bash
out=gpurun_out/r4aq; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_surface.py -x -q -m gpu -k "residual or box2gauss or box_to_gaussian or b2g or four_maps or variants" > $out/pytest.log 2>&1; echo "pytest rc $?"; tail -6 $out/pytest.log
python tools/residual_step_time.py 2>&1 | grep -v amdgpu | tee $out/residual_step_time.txt
