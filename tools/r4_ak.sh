#!/bin/bash
out=gpurun_out/r4ak; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_surface.py -x -q -m gpu -k "generic or residual or box2gauss or box_to_gaussian or variants" > $out/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $out/pytest.log
python tools/residual_step_time.py 2>&1 | grep -v amdgpu | tee $out/residual_step_time.txt
for c in 32 64 128 256; do echo -n "GPE_GEN_MIN_CHUNK=$c: "; GPE_GEN_MIN_CHUNK=$c python tools/residual_step_time.py 2>&1 | grep residual | cut -c1-60; done | tee -a $out/residual_step_time.txt
