#!/bin/bash
out=gpurun_out/r4al; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "padded_hidden or error_behaviour or pads_to" > $out/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $out/pytest.log
python tools/pinn2d_reference_size.py 2>&1 | grep -v amdgpu | grep "400, 400" | cut -c1-200 | tee $out/ref400.txt
for c in 16 32; do echo -n "GPE_GEN_MIN_CHUNK=$c: "; GPE_GEN_MIN_CHUNK=$c python tools/residual_step_time.py 2>&1 | grep residual | cut -c1-60; done | tee $out/chunk16.txt
