#!/bin/bash
out=gpurun_out/r4ac; mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_gpu_surface.py -x -q -m gpu -k "driver or pinn2d or notebook or refine" > $out/pytest_drivers.log 2>&1; echo "pytest rc $?"; tail -4 $out/pytest_drivers.log
timeout -k 10 200 python tools/accuracy_refine.py --tol 1e-7 --out $out/accuracy_refine_tol1e-7.json > $out/accuracy_refine.log 2>&1; echo "refine rc $?"; tail -3 $out/accuracy_refine.log
